#!/usr/bin/env python3
"""Run one GEMM shape a few times (rocprofv3 --pmc target).  one_gemm.py M N K [ln] [geglu] [res]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
flags = set(sys.argv[4:])
a = torch.randn(M, K, device="cuda", dtype=torch.float16)
w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5
No = N // 2 if "geglu" in flags else N
o = torch.empty(M, No, device="cuda", dtype=torch.float16)
g, c = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
r = torch.randn(M, No, device="cuda", dtype=torch.float16) if "res" in flags else None
b = torch.randn(N, device="cuda", dtype=torch.float16)
img = M // 8 if M % 8 == 0 and M >= 8192 else 0
for _ in range(5):
    if "ln" in flags:
        ops.gemm_ln(a, w, g, c, o, epilogue=1 if "geglu" in flags else 0, img_rows=img)
    else:
        ops.gemm(a, w, o, bias=b, res=r, epilogue=1 if "geglu" in flags else 0, img_rows=img)
torch.cuda.synchronize()
