#!/usr/bin/env python3
"""sha256 of the PNGs a worker returns for fixed requests: run it in several fresh processes (and under different launch-plan
settings) and diff the output -- the reference's same-seed contract across processes (tests/test_sdxl_worker.py:171-198).
   python tools/png_hash.py [label]"""
import hashlib, os, sys
from dataclasses import dataclass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MODEL", "synthetic")
os.environ.setdefault("MODEL_ROOT", "/nonexistent")
import sdlcm_amd  # noqa
from sdlcm_amd.backends.worker_factory import create_hip_worker


@dataclass
class Req:
    prompt: str
    size: str
    num_inference_steps: int = 4
    guidance_scale: float = 1.0
    seed: int = 12345


@dataclass
class Job:
    req: Req


w = create_hip_worker(worker_id=0)
label = sys.argv[1] if len(sys.argv) > 1 else ""
for size in ("128x128", "256x256", "512x512", "640x360"):
    png, seed = w.run_job(Job(Req(prompt="a beautiful mountain landscape at sunset", size=size)))
    print(f"{size} {hashlib.sha256(png).hexdigest()}", flush=True)
jobs = [Job(Req(prompt=f"prompt {i}", size="256x256", seed=i)) for i in range(8)]
res = w.run_jobs(jobs)                               # one batched pass of 8
for i, (png, seed) in enumerate(res):
    print(f"batch8[{i}] {hashlib.sha256(png).hexdigest()}", flush=True)
for i in (0, 5):
    png, _ = w.run_job(jobs[i])                      # the same requests alone
    print(f"solo[{i}] {hashlib.sha256(png).hexdigest()}", flush=True)
w.close()
