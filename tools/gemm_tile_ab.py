#!/usr/bin/env python3
"""Tile / variant sweep of the plain and LayerNorm-folded GEMM on the large-M UNet shapes of a batch-8 pass (GPU box), one process,
interleaved rounds; checks that every candidate gives the same bits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)


def case(M, N, K, ln=False, geglu=False, res=False, cands=((128, 128), (128, 160), (128, 64), (64, 160), (64, 64)), variants=(1, 2, 3)):
    a, w = rnd(M, K), rnd(N, K) * K ** -0.5
    No = N // 2 if geglu else N
    bias, r = rnd(N), (rnd(M, No) if res else None)
    g, c = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    ref = None
    out = []
    for bm, bn in cands:
        if N % bn or (bn == 160 and geglu):
            continue
        for v in variants:
            ops.plan_clear()
            ops.plan_set(0, M, N, K, 1, bm, bn, 1, v)
            o = torch.empty(M, No, device=DEV, dtype=torch.float16)
            if ln:
                fn = lambda: ops.gemm_ln(a, w, g, c, o, epilogue=1 if geglu else 0, img_rows=M // 8)
            else:
                fn = lambda: ops.gemm(a, w, o, bias=bias, res=r, epilogue=1 if geglu else 0, img_rows=M // 8)
            fn(); fn()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            if ref is None:
                ref = o.clone()
            same = torch.equal(ref, o)
            out.append((best, f"{bm}x{bn}/v{v} {best:6.1f}us{'' if same else ' DIFF!'}"))
    ops.plan_reset()
    fl = 2.0 * M * N * K
    out.sort()
    print(f"gemm{'_ln' if ln else ''}{'+geglu' if geglu else ''}{'+res' if res else ''} M{M} N{N} K{K}: best {fl / out[0][0] / 1e6:5.0f} TF | " + "  ".join(t for _, t in out[:7]), flush=True)


if __name__ == "__main__":
    case(32768, 2560, 320, ln=True, geglu=True)
    case(32768, 960, 320, ln=True)
    case(32768, 320, 320, ln=True)
    case(32768, 320, 320, res=True)
    case(32768, 320, 1280, res=True)
    case(8192, 5120, 640, ln=True, geglu=True)
    case(8192, 1920, 640, ln=True)
    case(8192, 640, 640, res=True)
    case(8192, 640, 2560, res=True)
    case(2048, 10240, 1280, ln=True, geglu=True)
    case(616, 24960, 768)
