#!/usr/bin/env python3
"""Key-split (two wave groups per query block, >= 1024 keys) against the unsplit streaming attention: accuracy of both against
torch SDPA (fp32), interleaved timing rounds."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
SHAPES = ((1, 8, 9216, 40), (8, 8, 9216, 40), (1, 8, 2304, 80), (8, 8, 2304, 80), (1, 10, 16384, 64)) if len(sys.argv) > 1 and sys.argv[1] == 'big' else None
for (B, heads, S, d) in SHAPES or ((1, 8, 4096, 40), (8, 8, 4096, 40), (1, 8, 1024, 80), (8, 8, 1024, 80), (1, 20, 1024, 64), (1, 10, 4096, 64), (1, 8, 1088, 40), (2, 8, 1030, 80)):
    C = heads * d
    g = torch.Generator().manual_seed(S + d)
    qkv = torch.randn(B * S, 3 * C, generator=g).half()
    qkv[S - 30, C:2 * C] = qkv[5, :C] * 3
    t = qkv.to(DEV)
    q, k, v = t[:, :C], t[:, C:2 * C], t[:, 2 * C:]
    ref = None
    if B * heads * S <= 40000:
        qh, kh, vh = (qkv[:, i * C:(i + 1) * C].float().reshape(B, S, heads, d).transpose(1, 2) for i in range(3))
        ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * S, C)
    res, outs = {}, {}
    for r in range(5):
        for ks in (0, 1):
            ops.set_attention_ksplit(ks)
            o = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
            fn = lambda: ops.attention(q, k, v, o, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(ks, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[ks] = o.float().cpu()
    ops.set_attention_ksplit(1)
    err = "" if ref is None else f"  max|err| unsplit {(outs[0] - ref).abs().max():.2e} split {(outs[1] - ref).abs().max():.2e}"
    print(f"attn B{B} h{heads} S{S} d{d}: unsplit {min(res[0]):7.1f}us  key-split {min(res[1]):7.1f}us  max|split-unsplit| {(outs[0] - outs[1]).abs().max():.2e}{err}", flush=True)
