"""One scripted session of a worker pool with a recording fake worker.  Run against the reference's ``WorkerPool``
(backends/worker_pool.py:135-419) in the build container by tests/golden/make_contract_golden.py -- the trace is committed as
``worker_contract.json["pool"]`` -- and against ``tools/minipool.MiniPool`` by the CPU tests, which must reproduce that trace:
factory call (keyword form, thread, the MODEL_ROOT / MODEL it sees), which thread runs the jobs, what a job's future receives,
the exception path (worker_pool.py:333-336), the same-mode switch no-op (:309-314), ordering around a mode switch, the
teardown order (:258-278) and the queue-full error (:358-366).

The scenario only talks to the pool through ``make_pool`` and the three job constructors, so it is the same code on both sides.
"""
from __future__ import annotations

import os
import queue
import threading
import time


def run_scenario(make_pool, GenerationJob, ModeSwitchJob, CustomJob):
    """make_pool(factory, registry_event, queue_max) -> pool with submit_job / switch_mode / get_current_mode / shutdown.
    -> list of events (JSON-serialisable).  The order of events recorded on ONE thread is deterministic; how the two threads'
    events interleave is not, so readers compare ``by_thread(events)``."""
    events = []
    lock = threading.Lock()
    serial = [0]

    def tname():
        return threading.current_thread().name

    def ev(*e):
        with lock:
            events.append([tname()] + list(e))         # first field: the thread the event was recorded on

    class RecWorker:
        def __init__(self, worker_id):
            serial[0] += 1
            self.serial, self.worker_id = serial[0], worker_id
            ev("worker_init", self.serial, worker_id, tname(), os.environ.get("MODEL_ROOT"), os.environ.get("MODEL"))

        def run_job(self, job):
            r = job.req
            ev("run_job", self.serial, tname(), r["id"])
            if r.get("fail"):
                raise RuntimeError(f"boom {r['id']}")
            return (b"png-%d" % r["id"], r.get("seed", 0))

        def __del__(self):
            ev("worker_del", self.serial, tname())

    def factory(*args, **kwargs):
        ev("factory", list(args), dict(kwargs), tname())
        return RecWorker(*args, **kwargs)

    def outcome(fut, timeout=10.0):
        try:
            r = fut.result(timeout=timeout)
            if isinstance(r, tuple):
                r = [x.decode() if isinstance(x, bytes) else x for x in r]
            return ["result", r]
        except Exception as e:      # noqa
            return ["error", type(e).__name__, str(e)]

    old_env = {k: os.environ.get(k) for k in ("MODEL_ROOT", "MODEL")}
    pool = make_pool(factory, lambda name, mode: ev("registry", name, mode), 4)
    try:
        ev("mode", pool.get_current_mode())
        # 1. a generation job: result tuple passes through untouched
        ev("job1", outcome(pool.submit_job(GenerationJob(req={"id": 1, "seed": 11}))))
        # 2. a failing job: the future carries the exception; the worker keeps serving
        ev("job2", outcome(pool.submit_job(GenerationJob(req={"id": 2, "fail": True}))))
        ev("job3", outcome(pool.submit_job(GenerationJob(req={"id": 3, "seed": 33}))))
        # 3. switching to the current mode: no factory call
        ev("switch_same", outcome(pool.switch_mode("mode-a")))
        # 4. slow custom job, a switch behind it, a generation job behind the switch: strict queue order
        started = threading.Event()

        def slow():
            started.set()
            time.sleep(0.15)
            ev("custom_ran", tname())
            return "slow_done"

        f_slow = pool.submit_job(CustomJob(handler=slow))
        f_switch = pool.switch_mode("mode-b")
        f_after = pool.submit_job(GenerationJob(req={"id": 4, "seed": 44}))
        ev("slow", outcome(f_slow))
        ev("switch_b", outcome(f_switch))
        ev("job4", outcome(f_after))
        ev("mode", pool.get_current_mode())
        del f_slow, f_switch, f_after
        # 5. an already resolved future is left alone (worker_pool.py:330-331)
        j = GenerationJob(req={"id": 5, "seed": 55})
        j.fut.set_result("preset")
        pool.submit_job(j)
        pool.q.join()
        ev("job5", outcome(j.fut))
        del j
        # 6. queue full: the consumer is held inside a job, the queue (max 4) is filled, one more is refused
        gate, inside = threading.Event(), threading.Event()

        def hold():
            inside.set()
            gate.wait(10)
            return "released"

        f_hold = pool.submit_job(CustomJob(handler=hold))
        inside.wait(10)
        fill = [pool.submit_job(GenerationJob(req={"id": 60 + i, "seed": i})) for i in range(4)]
        try:
            pool.submit_job(GenerationJob(req={"id": 99}))
            ev("overflow", ["accepted"])
        except queue.Full as e:
            ev("overflow", ["error", "Full", str(e)])
        gate.set()
        ev("hold", outcome(f_hold))
        ev("fill", [outcome(f) for f in fill])
        del f_hold, fill
        # 7. unknown mode: refused at submission
        try:
            pool.switch_mode("no-such-mode")
            ev("switch_unknown", ["accepted"])
        except Exception as e:      # noqa
            ev("switch_unknown", ["error", type(e).__name__])
    finally:
        pool.shutdown()
        ev("shutdown_done", pool.get_current_mode())
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return events


def by_thread(events):
    out = {}
    for e in events:
        out.setdefault(e[0], []).append(list(e[1:]))
    return out
