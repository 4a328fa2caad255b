#!/usr/bin/env python3
"""Run one attention shape a few times (target for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
B, S, Sk, d = (int(v) for v in sys.argv[1:5])
if len(sys.argv) > 5:
    ops.set_attention_impl(int(sys.argv[5]))
if len(sys.argv) > 6:
    ops.set_attention_waves(int(sys.argv[6]))
C = 8 * d
q, k, v = (torch.randn(B * n, C, device="cuda", dtype=torch.float16) for n in (S, Sk, Sk))
o = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
for _ in range(6):
    ops.attention(q, k, v, o, B, 8, S, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C)
torch.cuda.synchronize()
