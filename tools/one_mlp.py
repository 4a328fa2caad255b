#!/usr/bin/env python3
"""Run the fused FeedForward kernel a few times (rocprofv3 --pmc target).  one_mlp.py M [img_rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
from sdlcm_amd.packing import pack_ff2_cols, pack_geglu
M = int(sys.argv[1])
img = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
C, Fh = 320, 1280
g0 = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g0).half().cuda()
W1 = pack_geglu(torch.randn(2 * Fh, C, generator=g0) * C ** -0.5, None)[0].half().cuda()
c = torch.randn(2 * Fh, generator=g0).cuda()
g = torch.zeros(2 * Fh, dtype=torch.float32, device="cuda")
ops.ln_fold_refresh(W1, g)
W2, b2 = pack_ff2_cols((torch.randn(C, Fh, generator=g0) * Fh ** -0.5).half()).cuda(), torch.randn(C, generator=g0).half().cuda()
for _ in range(5):
    ops.mlp_geglu(x, W1, g, c, W2, b2, x, img_rows=img)
torch.cuda.synchronize()
