#!/usr/bin/env python3
"""Time conv_in from the fp32 latent (lcm_conv3x3_c4_f32in) on the UNet / VAE shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
DEV = "cuda"
for (B, H, Cout, pre) in ((8, 64, 320, False), (8, 64, 512, True), (1, 64, 320, False), (1, 64, 512, True), (1, 128, 320, False), (8, 96, 320, False)):
    lat = torch.randn(B, 4, H, H, device=DEV)
    w = torch.randn(Cout, 36, device=DEV, dtype=torch.float16) * 0.1
    b = torch.randn(Cout, device=DEV, dtype=torch.float16)
    pw, pb = torch.randn(4, 4, device=DEV), torch.randn(4, device=DEV)
    o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
    fn = lambda: ops.conv3x3_c4(lat, w, o, B, H, H, Cout, bias=b, pre_w=pw if pre else None, pre_b=pb if pre else None, in_scale=0.5)
    fn(); fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"conv_c4 B{B} {H}x{H} Cout{Cout} pre={pre}: {best:6.1f} us ({o.numel() * 2 / best / 1e6:.2f} TB/s of output)", flush=True)
