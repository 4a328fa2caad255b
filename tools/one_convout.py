#!/usr/bin/env python3
"""Time conv_out (<= 4 output channels) plain and with the fused GroupNorm on the VAE / UNet shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
DEV = "cuda"
for (B, H, C, Cout, mode) in ((8, 512, 128, 3, 1), (1, 512, 128, 3, 1), (8, 64, 320, 4, 0), (1, 64, 320, 4, 0), (1, 1024, 128, 3, 1)):
    M = B * H * H
    x = torch.randn(M, C, device=DEV, dtype=torch.float16)
    w = torch.randn(Cout, 9 * C, device=DEV, dtype=torch.float16) * 0.02
    b = torch.randn(Cout, device=DEV, dtype=torch.float16)
    sc, sh = torch.rand(B, C, device=DEV) + 0.5, torch.randn(B, C, device=DEV) * 0.1
    o = torch.empty(M, Cout, device=DEV, dtype=torch.uint8 if mode else torch.float32)
    for name, kw in (("plain", {}), ("gn", dict(gn_scale=sc, gn_shift=sh, silu=True))):
        fn = lambda: ops.conv3x3_smalln(x, w, o, B, H, H, C, Cout, bias=b, mode=mode, **kw)
        fn(); fn()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        print(f"conv_out B{B} {H}x{H} {C}->{Cout} {name:5s}: {best:7.1f} us  ({x.numel() * 2 / best / 1e6:.2f} TB/s of input)", flush=True)
