#!/usr/bin/env python3
"""A/B of the two self-attention kernels on the UNet shapes (GPU box): correctness against torch SDPA (fp32, CPU) on one
shape each, then interleaved timing rounds in one process (cdna_hip_programming.md rule 24)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"


def run(B, heads, S, d, impl, waves=0, q=None, k=None, v=None):
    C = heads * d
    o = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
    ops.set_attention_impl(impl); ops.set_attention_waves(waves)
    ops.attention(q, k, v, o, B, heads, S, q.shape[0] // B if False else k.shape[0] // B, d, ldq=q.stride(0), ldk=k.stride(0), ldv=v.stride(0), ldo=C)
    ops.set_attention_impl(1); ops.set_attention_waves(0)
    return o


def check(B, heads, Sq, Sk, d):
    C = heads * d
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B * s, C, generator=g).half() for s in (Sq, Sk, Sk))
    if Sk > 400:
        k[300] = q[7] * 6
        k[Sk - 30] = q[100] * 8
    qh, kh, vh = (t.float().reshape(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * Sq, C)
    outs = {}
    for name, impl, w in (("old", 0, 0), ("new4", 1, 4), ("new8", 1, 8)):
        C_ = heads * d
        o = torch.empty(B * Sq, C_, dtype=torch.float16, device=DEV)
        ops.set_attention_impl(impl); ops.set_attention_waves(w)
        ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), o, B, heads, Sq, Sk, d, ldq=C_, ldk=C_, ldv=C_, ldo=C_)
        torch.cuda.synchronize()
        outs[name] = o.float().cpu()
    ops.set_attention_impl(1); ops.set_attention_waves(0)
    e = {n: (o - ref).abs().max().item() for n, o in outs.items()}
    same = torch.equal(outs["new4"], outs["new8"])
    print(f"check B{B} h{heads} S{Sq}x{Sk} d{d}: max|err| {e}  new4==new8: {same}  ref scale {ref.abs().max().item():.3f}", flush=True)


def bench(B, heads, S, d, rounds=5, iters=20):
    C = heads * d
    qkv = torch.randn(B * S, 3 * C, device=DEV, dtype=torch.float16)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
    res = {}
    for r in range(rounds):
        for name, impl, w in (("old", 0, 0), ("new4", 1, 4), ("new8", 1, 8)):
            ops.set_attention_impl(impl); ops.set_attention_waves(w)
            for _ in range(3):
                ops.attention(q, k, v, o, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.attention(q, k, v, o, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / iters * 1e3)
    ops.set_attention_impl(1); ops.set_attention_waves(0)
    fl = 4.0 * B * heads * S * S * d
    print(f"bench B{B} h{heads} S{S} d{d}: " + "  ".join(f"{n} {min(t):7.1f}us ({fl / min(t) / 1e6:5.0f} TF)" for n, t in res.items()), flush=True)


if __name__ == "__main__":
    for shp in ((1, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (1, 10, 1024, 1024, 64), (1, 8, 3185, 3185, 40), (2, 8, 200, 130, 80),
                (1, 2, 333, 129, 64), (3, 8, 256, 256, 40)):
        check(*shp)
    for shp in ((1, 8, 4096, 40), (8, 8, 4096, 40), (1, 8, 1024, 80), (8, 8, 1024, 80), (1, 10, 4096, 64), (1, 20, 1024, 64), (2, 8, 9216, 40)):
        bench(*shp)
