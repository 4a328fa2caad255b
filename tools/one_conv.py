#!/usr/bin/env python3
"""Run one conv3x3 shape a few times (target for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
B, H, Cin, Cout = (int(v) for v in sys.argv[1:5])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ops.set_kernel_variant(variant)
x = torch.randn(B * H * H, Cin, device="cuda", dtype=torch.float16)
w = torch.randn(Cout, 9 * Cin, device="cuda", dtype=torch.float16)
o = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.float16)
gn = len(sys.argv) > 6 and "gn" in sys.argv[6:]      # GroupNorm-fused form (scale / shift tables applied while staging the halo)
res = torch.randn(B * H * H, Cout, device="cuda", dtype=torch.float16) if "res" in sys.argv[6:] else None      # + residual (conv2 of a resnet)
if gn:
    sc, sh = torch.rand(B, Cin, device="cuda") + 0.5, torch.randn(B, Cin, device="cuda") * 0.1
for _ in range(5):
    if gn:
        ops.conv3x3_gn(x, w, o, B, H, H, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True, res=res)
    else:
        ops.conv3x3(x, w, o, B, H, H, Cin, Cout, res=res)
torch.cuda.synchronize()
