#!/usr/bin/env python3
"""Where a run_job call behind the single-consumer pool spends its time (LCM_WORKER_TIMING=1): worker_calls.py [clients] [requests]
-> per call: jobs, gather / drain+prepare / pass / own PNG, and the gap between calls (the pool thread outside run_job)."""
import json, os, sys
os.environ["LCM_WORKER_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
import bench
import numpy as np
from sdlcm_amd.backends import hip_worker
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = int(sys.argv[2]) if len(sys.argv) > 2 else 128
res = bench.worker_leg(n_clients=n, n_requests=r, keep_timing=True)
tm = res.pop("timing")
calls = [t for t in tm if t[0] == "call"][-(r // n):]
ps = np.array([t[:3] for t in tm if t[0] != "call"][-2 * (r // n):], float)
print("passes (batch, conditioning ms, sampler ms):", np.round(ps[:, 0]), np.round(ps[:, 1] * 1e3, 1), np.round(ps[:, 2] * 1e3, 1))
a = np.array([c[1:] for c in calls], float)
print(json.dumps({"images_per_s": res["images_per_s"], "calls": len(calls), "jobs_per_call": a[:, 0].mean(),
                  "gather_ms": a[:, 1].mean() * 1e3, "drain_prepare_ms": a[:, 2].mean() * 1e3, "pass_ms": a[:, 3].mean() * 1e3,
                  "own_png_ms": a[:, 4].mean() * 1e3, "sum_ms": a[:, 1:].sum(1).mean() * 1e3}, indent=1))
