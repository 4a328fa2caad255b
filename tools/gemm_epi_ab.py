#!/usr/bin/env python3
"""A/B of the staged epilogue on the plain GEMMs with a residual (batch-8 transformer shapes); interleaved rounds, one process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
DEV = "cuda:0"


def case(M, N, K, img):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).half().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(DEV)
    b = torch.randn(N, generator=g).half().to(DEV)
    h = torch.randn(M, N, generator=g).half().to(DEV)
    t = {0: [], 1: []}
    for r in range(10):
        for st in ((0, 1) if r % 2 == 0 else (1, 0)):
            ops.set_staged_epilogue(3 if st else 0)
            ops.gemm(a, w, h, bias=b, res=h, img_rows=img)
            ops.debug_spin(100)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm(a, w, h, bias=b, res=h, img_rows=img)
            e1.record(); e1.synchronize()
            t[st].append(e0.elapsed_time(e1) / 8 * 1e3)
    ops.set_staged_epilogue(1)
    for st in (0, 1):
        t[st].sort()
    print(f"gemm+res M{M} N{N} K{K}: per-lane {t[0][len(t[0]) // 2]:6.1f} us (min {t[0][0]:6.1f})   staged {t[1][len(t[1]) // 2]:6.1f} us (min {t[1][0]:6.1f})", flush=True)


if __name__ == "__main__":
    case(32768, 320, 320, 4096)
    case(8192, 640, 640, 1024)
    case(8192, 640, 2560, 1024)
    case(2048, 1280, 1280, 256)
    case(2048, 1280, 5120, 256)
    case(4096, 320, 320, 4096)
    case(1024, 640, 640, 1024)
