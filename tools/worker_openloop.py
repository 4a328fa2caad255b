#!/usr/bin/env python3
"""Open-loop load through the single-consumer pool-shaped loop: requests arrive at a fixed rate (uniform spacing, no client waits for
its answer before the next one is sent), 512x512 4 steps, PNG included -> achieved images/s and latency p50 / p95 / max per rate.
usage: worker_openloop.py [seconds per rate] [rate ...]"""
import json, os, sys, threading, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MODEL", "synthetic")
os.environ.setdefault("MODEL_ROOT", "/nonexistent")
import sdlcm_amd  # noqa
from tools import minipool
from sdlcm_amd.backends.worker_factory import create_hip_worker

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
rates = [float(v) for v in sys.argv[2:]] or [20, 40, 60, 80, 100, 110, 120]
w = create_hip_worker(worker_id=0)
pool = minipool.MiniPool(lambda worker_id: w, {"m": "synthetic"}, "m", queue_max=512)
w.bind_queue(pool.q)


def req(i):
    return SimpleNamespace(prompt=f"open loop prompt {i}", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=7000 + i,
                           style_lora=None)


eng, key = w._engine, w._job_key(req(0))
for lane in range(eng.n_lanes):                      # every (lane, batch size) plan tuned and captured before anything is timed
    for bsz in eng.batch_sizes:
        eng.run_batch(key, [w._prepare(req(i), key) for i in range(bsz)], lane)
out = []
try:
    for rate in rates:
        n = int(rate * secs)
        lat, lock, done = [], threading.Lock(), threading.Event()
        t0 = time.perf_counter()

        def finished(f, ts):
            f.result()
            with lock:
                lat.append(time.perf_counter() - ts)
                if len(lat) == n:
                    done.set()
        for i in range(n):
            due = t0 + i / rate
            d = due - time.perf_counter()
            if d > 0:
                time.sleep(d)
            ts = time.perf_counter()
            pool.submit_job(minipool.GenerationJob(req=req(i))).add_done_callback(lambda f, ts=ts: finished(f, ts))
        done.wait(120)
        dt = time.perf_counter() - t0
        lat.sort()
        row = {"offered_per_s": rate, "served_per_s": round(len(lat) / dt, 1), "requests": n, "latency_p50_ms": round(lat[len(lat) // 2] * 1e3, 1),
               "latency_p95_ms": round(lat[int(0.95 * (len(lat) - 1))] * 1e3, 1), "latency_max_ms": round(lat[-1] * 1e3, 1)}
        print(json.dumps(row), flush=True)
        out.append(row)
        time.sleep(0.3)
finally:
    w.bind_queue(None)
    pool._worker = None
    pool.shutdown()
    w.close()
