#!/usr/bin/env python3
"""Images/s of batch-1 512x512 4-step passes with 1..4 lanes in flight (one captured graph per lane, replayed concurrently), and
of batch-2 passes on 1..2 lanes: where the low-load serving mode saturates."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops, weights
from sdlcm_amd.pipeline import LcmHipPipeline, draw_noise, guidance_scale_embedding

os.environ.setdefault("LCM_LANES", "4")
pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device="cuda:0")
pipe.max_lanes = 4 if hasattr(pipe, "max_lanes") else None


def prime(B, lane):
    P = pipe.plan(B, 64, 64, 4, lane=lane)
    with torch.cuda.stream(P.lane.stream):
        P.ehs.copy_(torch.randn(B * 77, 768, generator=torch.Generator().manual_seed(1)).half())
        for b in range(B):
            l0, extra = draw_noise(1000 + b, 64, 64, 3)
            P.lat0[b].copy_(l0[0])
            for i, e in enumerate(extra):
                P.noise[i, b].copy_(e[0])
        P.wemb.copy_(torch.from_numpy(guidance_scale_embedding(np.zeros(B, np.float32), P.wemb.shape[1])).half())
        pipe.tune(P)
        pipe._enqueue(P, 1.0)
        P.lane.stream.synchronize()
        g = ops.Graph()
        with g:
            pipe._enqueue(P, 1.0)
        P.graph = g
    return P


def run(Ps, K=10, W=2):
    for _ in range(W):
        for P in Ps:
            with torch.cuda.stream(P.lane.stream):
                P.graph.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for P in Ps:
            with torch.cuda.stream(P.lane.stream):
                P.graph.launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


for B in tuple(int(v) for v in sys.argv[1:]) or (1, 2):
    Ps = []
    for lane in range(4 if B == 1 else (3 if B == 2 else 2)):
        try:
            Ps.append(prime(B, lane))
        except Exception as e:
            print("lane", lane, "unavailable:", e); break
        dt = run(Ps)
        print(f"batch {B} x {len(Ps)} lanes: {B * len(Ps) / dt:6.1f} images/s, {dt * 1e3:6.2f} ms per round (latency of an image)", flush=True)
