#!/usr/bin/env python3
"""End-to-end run_job latency through the worker (prompt -> CLIP -> sampler -> D2H -> PNG), GPU box only."""
import os, sys, time, statistics
from dataclasses import dataclass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MODEL"] = "synthetic"
os.environ["LCM_WORKER_TIMING"] = "1"
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
        (rgb, _), seed = w._submit(Job(Req(seed=100 + i)))
        t1 = time.perf_counter()
        png = hip_worker.encode_png(rgb)
        t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); tp.append((t2 - t1) * 1e3)
    print(f"png level {lvl}: generate (CLIP + noise draw + H2D + graph + D2H) p50 {statistics.median(ts):.1f} ms, "
          f"PNG encode p50 {statistics.median(tp):.1f} ms ({len(png) / 1e3:.0f} KB), run_job p50 {statistics.median([a + b for a, b in zip(ts, tp)]):.1f} ms")

# ---- loaded case: N pool threads, one worker object each (as backends/worker_pool.py creates them), all
# attached to the one resident engine; jobs queued behind a running pass coalesce into batched passes.
import threading
for nthreads in (1, 2, 3, 4, 8, 16):
    ws = [w] + [create_hip_worker(worker_id=i) for i in range(1, nthreads)]
    per = 24
    lat = [[] for _ in ws]
    def loop(k):
        for i in range(per):
            t0 = time.perf_counter()
            ws[k].run_job(Job(Req(seed=1000 * k + i)))
            lat[k].append((time.perf_counter() - t0) * 1e3)
    for k in range(len(ws)):
        ws[k].run_job(Job(Req(seed=k)))
    wu = [threading.Thread(target=lambda k=k: [ws[k].run_job(Job(Req(seed=50 + k))) for _ in range(3)]) for k in range(len(ws))]
    [t.start() for t in wu]; [t.join() for t in wu]          # concurrent warm-up: every lane builds the plans it will use
    n0 = len(w._engine.batcher.batches) if w._engine.batcher else 0
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(k,)) for k in range(len(ws))]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    allp = sorted(x for l in lat for x in l)
    bs = w._engine.batcher.batches[n0:] if w._engine.batcher else []
    print(f"{nthreads:2d} callers: {nthreads * per / dt:6.1f} img/s  run_job p50 {allp[len(allp) // 2]:.1f} ms p95 {allp[int(len(allp) * 0.95)]:.1f} ms  "
          f"passes {len(bs)} mean batch {sum(bs) / max(1, len(bs)):.2f}", flush=True)
    tm = w._engine.timing
    if tm:
        big = [x for x in tm[-len(bs):] if x[0] == max(b for b in bs)] if bs else []
        if len(big) > 2:
            gaps = [b[3] - a[3] - b[1] - b[2] for a, b in zip(big, big[1:])]
            print(f"    batches of {big[0][0]}: conditioning {1e3 * sum(x[1] for x in big) / len(big):.1f} ms, sampler call {1e3 * sum(x[2] for x in big) / len(big):.1f} ms, "
                  f"dispatcher idle between passes {1e3 * sum(gaps) / len(gaps):.1f} ms", flush=True)
    for x in ws[1:]:
        x.close()

# ---- the sampler alone under concurrency (no PNG encode in the callers' loop): what the lanes buy on the GPU side.
# Two callers, micro-batching off (LCM_MICROBATCH=1 run) or on: each caller blocks in _submit (CLIP + H2D + graph + D2H).
for nthreads in (1, 2):
    ws = [w] + [create_hip_worker(worker_id=i) for i in range(1, nthreads)]
    per = 40
    def loop2(k):
        for i in range(per):
            ws[k]._submit(Job(Req(seed=3000 * k + i)))
    wu = [threading.Thread(target=lambda k=k: [ws[k]._submit(Job(Req(seed=70 + k))) for _ in range(3)]) for k in range(len(ws))]
    [t.start() for t in wu]; [t.join() for t in wu]
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop2, args=(k,)) for k in range(len(ws))]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"sampler only (no PNG), {nthreads} callers, lanes={w._engine.n_lanes}, microbatch={os.environ.get('LCM_MICROBATCH', '8')}: "
          f"{nthreads * per / dt:6.1f} img/s", flush=True)
    for x in ws[1:]:
        x.close()
w.close()
