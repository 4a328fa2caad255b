#!/usr/bin/env python3
"""One boundary convolution a few times (target for rocprofv3 --pmc passes):  one_boundary.py out|in B H [C]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
DEV = "cuda"
kind, B, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
C = int(sys.argv[4]) if len(sys.argv) > 4 else (128 if kind == "out" else 320)
if kind == "out":
    M = B * H * H
    x = torch.randn(M, C, device=DEV, dtype=torch.float16)
    w = torch.randn(3, 9 * C, device=DEV, dtype=torch.float16) * 0.02
    b = torch.randn(3, device=DEV, dtype=torch.float16)
    sc, sh = torch.rand(B, C, device=DEV) + 0.5, torch.randn(B, C, device=DEV) * 0.1
    o = torch.empty(M, 3, device=DEV, dtype=torch.uint8)
    fn = lambda: ops.conv3x3_smalln(x, w, o, B, H, H, C, 3, bias=b, mode=1, gn_scale=sc, gn_shift=sh, silu=True)
else:
    lat = torch.randn(B, 4, H, H, device=DEV)
    w = torch.randn(C, 36, device=DEV, dtype=torch.float16) * 0.1
    b = torch.randn(C, device=DEV, dtype=torch.float16)
    o = torch.empty(B * H * H, C, device=DEV, dtype=torch.float16)
    fn = lambda: ops.conv3x3_c4(lat, w, o, B, H, H, C, bias=b)
for _ in range(6):
    fn()
torch.cuda.synchronize()
