#!/usr/bin/env python3
"""64-row (auto, small grids) against 128-row workgroups of the streaming self-attention kernel on the batch-1 shapes that
leave CUs idle; interleaved rounds, bit-equality."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
for (B, heads, S, d) in ((1, 8, 1024, 80), (1, 20, 1024, 64), (2, 8, 1024, 80), (1, 10, 4096, 64), (1, 8, 4096, 40), (1, 8, 256, 160)):
    C = heads * d
    qkv = torch.randn(B * S, 3 * C, device=DEV, dtype=torch.float16)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    res, outs = {}, {}
    for r in range(5):
        for w in (4, 0):
            ops.set_attention_waves(w)
            o = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
            fn = lambda: ops.attention(q, k, v, o, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(w, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[w] = o
    ops.set_attention_waves(0)
    print(f"attn B{B} h{heads} S{S} d{d}: waves4 {min(res[4]):7.1f}us  auto {min(res[0]):7.1f}us  equal {torch.equal(outs[4], outs[0])}", flush=True)
