#!/usr/bin/env python3
"""End-to-end run_job latency through the worker (prompt -> CLIP -> sampler -> D2H -> PNG), GPU box only."""
import os, sys, time, statistics
from dataclasses import dataclass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MODEL"] = "synthetic"
import sdlcm_amd  # noqa
from sdlcm_amd.backends.worker_factory import create_hip_worker
from sdlcm_amd.backends import hip_worker


@dataclass
class Req:
    prompt: str = "a beautiful mountain landscape at sunset"
    size: str = "512x512"
    num_inference_steps: int = 4
    guidance_scale: float = 1.0
    seed: int = 42


@dataclass
class Job:
    req: Req


w = create_hip_worker(worker_id=0)
for lvl in (6, 1):
    os.environ["LCM_PNG_COMPRESS"] = str(lvl)
    for i in range(3):
        w.run_job(Job(Req(seed=i)))
    ts, tp = [], []
    for i in range(20):
        t0 = time.perf_counter()
        out, seed = w._generate(Job(Req(seed=100 + i)))
        t1 = time.perf_counter()
        png = hip_worker.encode_png(out["rgb"][0])
        t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); tp.append((t2 - t1) * 1e3)
    print(f"png level {lvl}: generate (CLIP + noise draw + H2D + graph + D2H) p50 {statistics.median(ts):.1f} ms, "
          f"PNG encode p50 {statistics.median(tp):.1f} ms ({len(png) / 1e3:.0f} KB), run_job p50 {statistics.median([a + b for a, b in zip(ts, tp)]):.1f} ms")
w.close()
