#!/usr/bin/env python3
"""Ring depth of the plain GEMM on the batch-8 transformer shapes: 2 / 3 / 4 stages per tile, timed COLD (caches flushed before
every launch, as the autotuner does: in situ a layer's weights and operands never come from a previous run of the same layer)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import autotune, ops
DEV = "cuda:0"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)


def case(M, N, K, res=True, tiles=((128, 160), (64, 160), (128, 128), (128, 64))):
    a, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N)
    r = rnd(M, N) if res else None
    o = torch.empty(M, N, device=DEV, dtype=torch.float16)
    out = []
    for bm, bn in tiles:
        if N % bn:
            continue
        for v in (2, 3, 4):
            ops.plan_clear()
            ops.plan_set(0, M, N, K, 1, bm, bn, 1, v)
            fn = lambda: ops.gemm(a, w, o, bias=b, res=r, img_rows=M // 8)
            out.append((autotune._time_cold(fn, 8) * 1e3, f"{bm}x{bn}/S{v}"))
    ops.plan_reset()
    out.sort()
    print(f"M{M} N{N} K{K}: " + "  ".join(f"{n} {t:5.1f}" for t, n in out), flush=True)


if __name__ == "__main__":
    case(8192, 640, 640); case(8192, 640, 2560); case(2048, 1280, 1280); case(2048, 1280, 5120); case(32768, 320, 320); case(32768, 320, 1280)
