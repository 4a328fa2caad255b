"""Batching behind the reference's single-consumer pool (backends/worker_pool.py:294-341) -- CPU tests.

``WorkerPool`` has ONE consumer thread that calls ``job.execute(worker)`` synchronously, so ``HipLcmWorker.run_job`` itself
collects the batch: it drains the compatible ``GenerationJob``s queued behind the running one from ``pool.q`` and finishes
them as the pool's loop would.  The pool here is ``tools/minipool.MiniPool``, which is first shown to behave like the
reference's ``WorkerPool`` (a session recorded from the reference in the build container, replayed)."""
import json
import os
import sys
import threading
import time
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdlcm_amd  # noqa: E402,F401
from tools import minipool  # noqa: E402
from pool_scenario import by_thread, run_scenario  # noqa: E402


def test_minipool_replays_the_reference_pool_recording():
    """tests/golden/worker_contract.json["pool"] is the trace of tests/pool_scenario.py against the reference's WorkerPool with
    a recording fake factory / worker (tests/golden/make_contract_golden.py).  MiniPool must produce the same trace, thread by
    thread: factory keywords and thread, the environment the worker sees, jobs on 'WorkerThread', futures' results and errors,
    same-mode switch no-op, order around a switch, teardown (unregister -> del worker on the switching thread -> factory), the
    queue-full text."""
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "worker_contract.json")))

    def make_pool(factory, registry_event, queue_max):
        return minipool.MiniPool(factory, {"mode-a": "a.safetensors", "mode-b": "b.safetensors"}, "mode-a", model_root="/models",
                                 queue_max=queue_max, on_event=lambda name, mode: registry_event(name, mode))

    got = run_scenario(make_pool, minipool.GenerationJob, minipool.ModeSwitchJob, minipool.CustomJob)
    want = by_thread(doc["pool"])
    have = by_thread(json.loads(json.dumps(got)))
    assert set(have) == set(want) == {"MainThread", "WorkerThread"}
    for t in want:
        assert have[t] == want[t], t


# ----------------------------------------------------------------------------------------------------------------------
class _FakePipe:
    sched = types.SimpleNamespace(init_noise_sigma=1.0)


def _fake_worker(run_pass, lanes=2, max_batch=8):
    """A HipLcmWorker whose engine is the real dispatcher (MicroBatcher) over a recording stand-in for the sampler."""
    from sdlcm_amd.backends import hip_worker
    from sdlcm_amd.backends.batching import MicroBatcher
    eng = types.SimpleNamespace(pipe=_FakePipe(), batch_sizes=(1, 2, 4, 8), passes=[])

    def run_batch(key, items, lane=0):
        eng.passes.append((lane, key, [it[0].prompt for it in items]))
        run_pass(key, items)
        return [(np.full((8, 8, 3), it[1] % 251, np.uint8), np.zeros((1, 4, 8, 8), np.float16)) for it in items]

    eng.batcher = MicroBatcher(run_batch, max_batch=max_batch, lanes=lanes)
    w = object.__new__(hip_worker.HipLcmWorker)
    w.worker_id, w._engine = 0, eng
    return w


def _req(i, size="64x64", steps=2, seed=None):
    return types.SimpleNamespace(prompt=f"p{i}", size=size, num_inference_steps=steps, guidance_scale=1.0, seed=1000 + i if seed is None else seed)


class _Held:
    """Keeps the pool's consumer inside a job so that a known queue can be built behind it."""

    def __init__(self, pool):
        self.gate, inside = threading.Event(), threading.Event()

        def hold():
            inside.set()
            self.gate.wait(10)
        self.fut = pool.submit_job(minipool.CustomJob(handler=hold))
        assert inside.wait(10)

    def release(self):
        self.gate.set()
        self.fut.result(10)


def _pool_with(worker, queue_max=64):
    pool = minipool.MiniPool(lambda worker_id: worker, {"m": "synthetic", "n": "synthetic"}, "m", queue_max=queue_max)
    worker.bind_queue(pool.q)
    return pool


def test_eight_queued_jobs_become_one_pass_and_every_future_gets_its_own_result():
    from sdlcm_amd.backends.hip_worker import encode_png
    w = _fake_worker(lambda key, items: time.sleep(0.01))
    pool = _pool_with(w)
    try:
        held = _Held(pool)
        futs = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(8)]
        held.release()
        res = [f.result(10) for f in futs]
        pool.q.join()                                     # task_done was called for the drained jobs too
        assert [len(p[2]) for p in w._engine.passes] == [8]
        assert w._engine.passes[0][2] == [f"p{i}" for i in range(8)]          # queue order inside the batch
        for i, (png, seed) in enumerate(res):
            assert seed == 1000 + i and png == encode_png(np.full((8, 8, 3), seed % 251, np.uint8))
    finally:
        pool.shutdown()


def test_sixteen_queued_jobs_fill_both_lanes_with_full_batches():
    started, both = [], threading.Event()

    def run_pass(key, items):
        started.append(len(items))
        if len(started) == 2:
            both.set()
        assert both.wait(5), "the two full batches were not in flight together"

    w = _fake_worker(run_pass)
    pool = _pool_with(w)
    try:
        held = _Held(pool)
        futs = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(19)]
        held.release()
        assert [f.result(10)[1] for f in futs] == [1000 + i for i in range(19)]
        sizes = [len(p[2]) for p in w._engine.passes]
        assert sizes[:2] == [8, 8] and {p[0] for p in w._engine.passes[:2]} == {0, 1}
        assert sorted(sizes[2:]) in ([1, 2], [3]) or sum(sizes[2:]) == 3       # the remaining three: the next call's batch
    finally:
        pool.shutdown()


def test_other_keys_keep_their_place_and_a_mode_switch_is_a_barrier():
    order = []
    w = _fake_worker(lambda key, items: order.append(("pass", key[0], [it[0].prompt for it in items])))
    pool = _pool_with(w)
    try:
        held = _Held(pool)
        a = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(3)]                     # 64x64
        b = pool.submit_job(minipool.GenerationJob(req=_req(10, size="128x128")))                       # another key: stays queued
        a2 = pool.submit_job(minipool.GenerationJob(req=_req(3)))                                        # joins the 64x64 batch over it
        bad = pool.submit_job(minipool.GenerationJob(req=_req(11, size="bogus")))                       # raises in its own turn
        sw = pool.submit_job(minipool.ModeSwitchJob(target_mode="m", on_complete=lambda m: order.append(("switch", m))))
        sw.add_done_callback(lambda f: order.append(("switch_done",)))
        after = [pool.submit_job(minipool.GenerationJob(req=_req(20 + i))) for i in range(2)]          # 64x64, but behind the switch
        held.release()
        for f in a + [a2, b] + after:
            f.result(10)
        with pytest.raises(RuntimeError, match="Invalid size 'bogus'"):
            bad.result(10)
        assert sw.result(10) == {"mode": "m", "status": "already_loaded"}
        passes = [o for o in order if o[0] == "pass"]
        assert passes[0] == ("pass", 64, ["p0", "p1", "p2", "p3"])              # a2 overtook b (other key) but nothing else
        assert passes[1] == ("pass", 128, ["p10"])
        i_sw = order.index(("switch_done",))
        later = [o for o in order[i_sw:] if o[0] == "pass"]
        assert later == [("pass", 64, ["p20", "p21"])] and passes[2:] == later   # nothing behind the switch ran before it
    finally:
        pool.shutdown()


def test_a_failed_pass_reaches_every_drained_future_and_the_queue_accounting_survives():
    def run_pass(key, items):
        if key[0] == 64:
            raise ValueError("pass failed")

    w = _fake_worker(run_pass)
    pool = _pool_with(w)
    try:
        held = _Held(pool)
        futs = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(4)]
        held.release()
        for f in futs:
            with pytest.raises(ValueError, match="pass failed"):
                f.result(10)
        pool.q.join()
        ok = pool.submit_job(minipool.GenerationJob(req=_req(0, size="128x128")))
        assert ok.result(10)[1] == 1000
    finally:
        pool.shutdown()


def test_without_a_bound_queue_or_with_drain_off_every_job_is_its_own_pass(monkeypatch):
    w = _fake_worker(lambda key, items: None)
    pool = minipool.MiniPool(lambda worker_id: w, {"m": "synthetic"}, "m")
    try:
        held = _Held(pool)
        futs = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(3)]
        held.release()
        [f.result(10) for f in futs]
        assert [len(p[2]) for p in w._engine.passes] == [1, 1, 1]
    finally:
        pool.shutdown()


def test_the_reference_singleton_is_found_without_binding(monkeypatch):
    """Behind ``get_worker_pool()`` (backends/worker_pool.py:425-469) nothing has to be bound: the worker looks the loaded
    module's ``_worker_pool`` up and uses its ``q`` when that pool currently holds this worker."""
    w = _fake_worker(lambda key, items: None)
    pool = minipool.MiniPool(lambda worker_id: w, {"m": "synthetic"}, "m")
    mod = types.ModuleType("backends.worker_pool")
    mod._worker_pool = pool
    monkeypatch.setitem(sys.modules, "backends.worker_pool", mod)
    try:
        assert w._pool_queue() is pool.q
        held = _Held(pool)
        futs = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(4)]
        held.release()
        [f.result(10) for f in futs]
        assert [len(p[2]) for p in w._engine.passes] == [4]
        other = _fake_worker(lambda key, items: None)
        assert other._pool_queue() is None                 # a worker the pool does not hold never touches its queue
        monkeypatch.setenv("LCM_DRAIN_QUEUE", "0")
        assert w._pool_queue() is None
    finally:
        pool.shutdown()


def test_dispatcher_survives_the_window_race_between_two_lanes():
    """with a batching window and two lanes, both lanes could wait on the same head; one took the batch, the other
    popped from an empty queue and its thread died -- if it was lane 0, every later request hung.  Timing of the report:
    run_batch 20 ms, window 30 ms, second submit 35 ms after the first."""
    from sdlcm_amd.backends.batching import MicroBatcher

    def run(key, items, lane):
        time.sleep(0.02)
        return list(items)

    mb = MicroBatcher(run, max_batch=8, window_ms=30, lanes=2)
    try:
        for rnd in range(6):
            f1 = mb.submit("k", 2 * rnd)
            time.sleep(0.035)
            f2 = mb.submit("k", 2 * rnd + 1)
            assert f1.result(5) == 2 * rnd and f2.result(5) == 2 * rnd + 1
        assert all(t.is_alive() for t in mb._threads)
    finally:
        mb.close()


def test_other_lanes_serve_when_lane0_is_gone():
    from sdlcm_amd.backends.batching import MicroBatcher
    mb = MicroBatcher(lambda key, items, lane: [(lane, x) for x in items], max_batch=8, lanes=2)
    try:
        mb._lane0_alive = lambda: False
        assert mb.submit("k", 1).result(5)[1] == 1
    finally:
        mb.close()


def test_factory_binds_the_named_queue_to_every_worker_it_creates(monkeypatch):
    """A pool built by hand names its queue once (worker_factory.set_pool_queue); workers created later -- the pool re-creates
    its worker on every mode switch -- are bound by the factory."""
    import queue
    from sdlcm_amd.backends import hip_worker, worker_factory
    made = []

    class W:
        def __init__(self, worker_id):
            self.q = None
            made.append(self)

        def bind_queue(self, q):
            self.q = q
    monkeypatch.setattr(hip_worker, "HipLcmWorker", W)
    monkeypatch.setenv("MODEL", "synthetic")
    q = queue.Queue()
    first = worker_factory.create_hip_worker(worker_id=0)
    assert first.q is None
    worker_factory.set_pool_queue(q, first)
    try:
        assert first.q is q and worker_factory.create_hip_worker(worker_id=0).q is q
    finally:
        worker_factory.set_pool_queue(None)
    assert worker_factory.create_hip_worker(worker_id=0).q is None


def test_run_job_returns_only_when_every_pass_it_started_has_left_the_gpu():
    """Whatever is next in the queue -- a mode switch or another generation job -- the pool's loop gets control back only when all
    passes of the call are done: nothing queued behind a barrier starts before the work queued in front of it is finished."""
    gate = threading.Event()

    def run_pass(key, items):
        if any(it[0].prompt == "p15" for it in items):        # the second batch of the first call: held on "the GPU"
            assert gate.wait(10)

    for barrier_next in (False, True):
        gate.clear()
        w = _fake_worker(run_pass)
        pool = _pool_with(w)
        try:
            held = _Held(pool)
            first = [pool.submit_job(minipool.GenerationJob(req=_req(i))) for i in range(16)]            # drained as 8 + 8
            nxt = pool.submit_job(minipool.ModeSwitchJob(target_mode="m") if barrier_next
                                  else minipool.GenerationJob(req=_req(99, size="128x128")))
            held.release()
            [f.result(10) for f in first[1:8]]                    # the first batch is through (first[0] = the call itself)
            time.sleep(0.3)
            assert not nxt.done() and not first[0].done() and not first[15].done()
            gate.set()
            [f.result(10) for f in first]
            nxt.result(10)
            pool.q.join()
        finally:
            gate.set()
            pool.shutdown()


def test_burst_tail_is_padded_to_a_plan_size_when_that_is_cheaper():
    """A complete set of 15 (8 on lane 0) leaves 7: one pass of 8 with the last item repeated instead of 4 + 2 + 1 one after
    the other; the repeated item's result is dropped, every real item gets its own.  5 stays 4 + 1; without the burst flag
    nothing is padded."""
    import threading
    from sdlcm_amd.backends.batching import MicroBatcher
    gate, seen = threading.Event(), []

    def run(key, items, lane=0):
        seen.append((lane, list(items)))
        if lane == 0:
            gate.wait(10)
        return [i * 10 for i in items]
    mb = MicroBatcher(run, max_batch=8, lanes=2)
    try:
        futs = [mb.submit("k", i, burst=True) for i in range(15)]
        res = [f.result(timeout=10) for f in futs[8:]]
        assert res == [i * 10 for i in range(8, 15)]
        lane1 = [it for ln, it in seen if ln == 1]
        assert lane1 == [[8, 9, 10, 11, 12, 13, 14, 14]]
        gate.set()
        assert [f.result(timeout=10) for f in futs[:8]] == [i * 10 for i in range(8)]
        assert sorted(mb.batches) == [7, 8]
    finally:
        gate.set()
        mb.close()
    one = MicroBatcher(lambda key, items: [i + 1 for i in items], max_batch=8, lanes=1)
    try:
        assert one._split_cost(5) < one._cost(8) and one._split_cost(7) > one._cost(8) and one._split_cost(3) > one._cost(4)
    finally:
        one.close()


def test_dispatcher_stress_every_future_resolves_once_with_its_own_result():
    """Random mix of open arrivals and complete (burst) sets over three keys and two lanes, passes of random length, a failing
    key: every future resolves exactly once, with its own item's result or the pass's error; batch sizes stay plan sizes;
    repeated (padding) items never reach a caller."""
    import random
    import threading
    import time as _t
    from sdlcm_amd.backends.batching import MicroBatcher
    rng = random.Random(7)
    seen_sizes, lock = [], threading.Lock()

    def run(key, items, lane=0):
        with lock:
            seen_sizes.append(len(items))
        _t.sleep(rng.random() * 0.004)
        if key == "bad":
            raise ValueError("pass failed")
        return [(key, i) for i in items]
    mb = MicroBatcher(run, max_batch=8, lanes=2)
    futs = []
    try:
        n = 0
        for rnd in range(60):
            key = rng.choice(["a", "a", "b", "bad"])
            if rng.random() < 0.5:
                k = rng.randint(1, 16)
                futs += [(key, n + j, mb.submit(key, n + j, burst=True)) for j in range(k)]
            else:
                k = rng.randint(1, 5)
                for j in range(k):
                    futs.append((key, n + j, mb.submit(key, n + j)))
                    _t.sleep(rng.random() * 0.001)
            n += k
            if rng.random() < 0.3:
                _t.sleep(rng.random() * 0.01)
        for key, i, f in futs:
            if key == "bad":
                with pytest.raises(ValueError, match="pass failed"):
                    f.result(timeout=30)
            else:
                assert f.result(timeout=30) == (key, i)
        assert set(seen_sizes) <= {1, 2, 4, 8}
        assert sum(mb.batches) == sum(1 for key, _, _ in futs if key != "bad")
    finally:
        mb.close()
