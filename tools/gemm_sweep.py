#!/usr/bin/env python3
"""Every (tile, ring depth) of the plain GEMM kernel on one shape, timed cold; the shipped / first-use plan's choice is marked.
usage: gemm_sweep.py M N K [img_rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import autotune, ops
DEV = "cuda:0"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
ws = torch.empty(256 << 20, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)


def case(M, N, K, img_rows):
    a, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N)
    o = torch.empty(M, N, device=DEV, dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, o, bias=b, img_rows=img_rows)
    ops.plan_reset()
    t_default = autotune._time_cold(fn, 8) * 1e3
    sp_canon = ops.canonical_splits(0, img_rows, N, K)
    out = []
    for bm, bn in ((128, 160), (64, 160), (128, 128), (128, 64), (64, 128), (64, 64)):
        if N % bn:
            continue
        for v in (1, 2, 3, 4):
            ops.plan_clear()
            try:
                ops.plan_set(0, M, N, K, 1, bm, bn, sp_canon, v)
                out.append((autotune._time_cold(fn, 8) * 1e3, f"{bm}x{bn}/S{v}"))
            except Exception as e:
                out.append((1e9, f"{bm}x{bn}/S{v} {str(e)[:40]}"))
    ops.plan_reset()
    out.sort()
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K} (canonical splits {sp_canon}): default plan {t_default:6.1f} us ({fl / t_default / 1e6:4.0f} TF/s) | " +
          "  ".join(f"{n} {t:5.1f}" for t, n in out[:8]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 3:
        M, N, K = (int(v) for v in sys.argv[1:4])
        case(M, N, K, int(sys.argv[4]) if len(sys.argv) > 4 else M)
    else:
        for M, N, K in ((1024, 10240, 1280), (1024, 1280, 5120), (1024, 3840, 1280), (1024, 1280, 1280), (4096, 5120, 640), (4096, 640, 2560),
                        (4096, 1920, 640), (4096, 640, 640), (2048, 10240, 1280), (2048, 1280, 5120)):
            case(M, N, K, M if M <= 4096 else 1024)
