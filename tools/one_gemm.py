#!/usr/bin/env python3
"""Run one GEMM shape a few times (rocprofv3 --pmc calibration target: every A byte is read exactly once when N == 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
a = torch.randn(M, K, device="cuda", dtype=torch.float16)
w = torch.randn(N, K, device="cuda", dtype=torch.float16)
o = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(3):
    ops.gemm(a, w, o)
torch.cuda.synchronize()
