#!/usr/bin/env python3
"""64- against 128-row workgroups of the register-staged attention kernel on the small-grid d = 160 / cross-attention launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
DEV = "cuda"
for (B, heads, Sq, Sk, d) in ((1, 8, 256, 256, 160), (1, 8, 256, 77, 160), (1, 8, 64, 64, 160), (1, 8, 64, 77, 160), (8, 8, 256, 256, 160),
                              (1, 8, 4096, 77, 40), (1, 8, 1024, 77, 80), (8, 8, 4096, 77, 40), (1, 20, 1024, 77, 64)):
    C = heads * d
    q = torch.randn(B * Sq, C, device=DEV, dtype=torch.float16)
    k = torch.randn(B * Sk, C, device=DEV, dtype=torch.float16)
    v = torch.randn(B * Sk, C, device=DEV, dtype=torch.float16)
    res, outs = {}, {}
    for r in range(5):
        for w in (4, 2):
            ops.set_attention_waves(w)
            o = torch.empty(B * Sq, C, dtype=torch.float16, device=DEV)
            fn = lambda: ops.attention(q, k, v, o, B, heads, Sq, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C)
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[w] = o
    ops.set_attention_waves(0)
    print(f"attn B{B} h{heads} S{Sq}x{Sk} d{d}: 128-row {min(res[4]):6.1f}us  64-row {min(res[2]):6.1f}us  equal {torch.equal(outs[4], outs[2])}", flush=True)
