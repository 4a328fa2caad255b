#!/usr/bin/env python3
"""A/B: FeedForward of a transformer block as one fused kernel (csrc/mlp_fused.hip) against the two launches it replaces, on the
64x64-level shapes (C = 320).  Interleaved rounds in one process, random data, cold-ish weights (a 512 MB fill between launches
is optional: LCM_AB_COLD=1).  Prints the median / min per arm and checks the outputs are bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa: E402,F401
from sdlcm_amd import ops  # noqa: E402
from sdlcm_amd.packing import pack_ff2_cols, pack_geglu  # noqa: E402

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.float16)


def case(M, img, rounds=12):
    C, Fh = 320, 1280
    x = rnd(M, C, seed=1).to(DEV)
    W1 = pack_geglu(rnd(2 * Fh, C, seed=4, scale=C ** -0.5).float(), None)[0].half().to(DEV)
    c = rnd(2 * Fh, seed=5).float().to(DEV)
    g = torch.zeros(2 * Fh, dtype=torch.float32, device=DEV)
    ops.ln_fold_refresh(W1, g)
    W2, b2 = pack_ff2_cols(rnd(C, Fh, seed=6, scale=Fh ** -0.5)).to(DEV), rnd(C, seed=7).to(DEV)
    ff = torch.empty(M, Fh, dtype=torch.float16, device=DEV)
    ha, hb = x.clone(), x.clone()
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV) if os.environ.get("LCM_AB_COLD") == "1" else None

    def two():
        ops.gemm_ln(ha, W1, g, c, ff, epilogue=1, img_rows=img)
        ops.gemm(ff, W2, ha, bias=b2, res=ha, img_rows=img)

    def one():
        ops.mlp_geglu(hb, W1, g, c, W2, b2, hb, img_rows=img)

    two(); one()
    torch.cuda.synchronize()
    same = torch.equal(ha, hb)
    t = {"two": [], "one": []}
    for r in range(rounds):
        for name, fn in (("two", two), ("one", one)) if r % 2 == 0 else (("one", one), ("two", two)):
            ha.copy_(x); hb.copy_(x)
            if flush is not None:
                flush.fill_(1)
            ops.debug_spin(100)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            t[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * 2560 * C + 2.0 * M * C * Fh
    for k, v in t.items():
        v.sort()
        print(f"M {M:6d} img {img:5d} {k}: median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us  ({fl / v[len(v) // 2] / 1e6:6.0f} TFLOP/s)", flush=True)
    print(f"   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    for M, img in ((32768, 4096), (73728, 9216), (24576, 4096), (16384, 4096)):
        case(M, img)
