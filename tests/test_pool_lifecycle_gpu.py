"""The worker under the reference pool's lifecycle (backends/worker_pool.py:204-341), driven by a small pool-shaped
consumer written for this test: the worker is built by an injected factory on the MAIN thread, jobs run on a separate
worker thread, a mode switch tears the worker down ON that thread with the pool's own teardown -- ``del worker;
gc.collect(); torch.cuda.empty_cache()``, no ``close()`` (worker_pool.py:270-276) -- and builds the next mode's worker
there; a failing job leaves the worker usable (worker_pool.py:333-336).  Checked: results do not change across a
teardown / rebuild, and the unloaded mode's device memory really is returned."""
import gc
import os
import queue
import threading
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Optional

import pytest
import torch

pytestmark = pytest.mark.gpu


@dataclass
class _StyleLora:
    style: Optional[str] = None
    level: int = 0


@dataclass
class _Req:
    prompt: str
    size: str = "128x128"
    num_inference_steps: int = 2
    guidance_scale: float = 1.0
    seed: Optional[int] = None
    style_lora: _StyleLora = field(default_factory=_StyleLora)


class _Job:
    def __init__(self, req=None, switch_to=None):
        self.req, self.switch_to, self.fut = req, switch_to, Future()

    def execute(self, worker):
        return worker.run_job(self)


class _MiniPool:
    """The part of WorkerPool this test needs: one queue, one worker thread, factory injection, mode switching."""

    def __init__(self, factory, first_mode):
        self.factory, self.q, self.stop = factory, queue.Queue(), threading.Event()
        self.mode, self.worker = None, None
        self._load(first_mode)                           # main thread, like WorkerPool.__init__
        self.thread = threading.Thread(target=self._loop, daemon=True, name="WorkerThread")
        self.thread.start()

    def _load(self, mode):
        if self.worker is not None:
            del self.worker                              # worker_pool.py:270-276, verbatim teardown
            self.worker = None
            gc.collect()
            torch.cuda.empty_cache()
        os.environ["MODEL"] = mode
        self.worker = self.factory(worker_id=0)
        self.mode = mode

    def _loop(self):
        while not self.stop.is_set():
            try:
                job = self.q.get(timeout=0.2)
            except queue.Empty:
                continue
            try:
                if job.switch_to is not None:
                    if job.switch_to != self.mode:
                        self._load(job.switch_to)        # on the worker thread
                    job.fut.set_result(self.mode)
                else:
                    job.fut.set_result(job.execute(self.worker))
            except Exception as e:
                job.fut.set_exception(e)
            finally:
                self.q.task_done()

    def submit(self, job):
        self.q.put(job)
        return job.fut

    def shutdown(self):
        self.q.join()
        self.stop.set()
        self.thread.join(10)
        del self.worker
        self.worker = None
        gc.collect()
        torch.cuda.empty_cache()


def _mem():
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated()


def test_pool_shaped_lifecycle_across_threads_and_modes():
    from sdlcm_amd.backends import hip_worker
    from sdlcm_amd.backends.worker_factory import create_hip_worker
    os.environ.setdefault("MODEL_ROOT", "/nonexistent")
    old_model = os.environ.get("MODEL")
    gc.collect()
    torch.cuda.empty_cache()
    base = _mem()
    pool = _MiniPool(create_hip_worker, "synthetic")
    try:
        jobs = [_Job(_Req(prompt=f"a lighthouse {i}", seed=100 + i)) for i in range(3)]
        first = [pool.submit(j).result(timeout=600) for j in jobs]
        assert all(p[:8] == b"\x89PNG\r\n\x1a\n" and s == 100 + i for i, (p, s) in enumerate(first))
        sd15_mem = _mem() - base
        assert sd15_mem > 1.5e9                          # the SD1.5 weights are resident
        # a failing job (worker_pool.py:333-336): the future carries the error, the worker keeps serving
        bad = pool.submit(_Job(_Req(prompt="x", size="bogus", seed=1)))
        with pytest.raises(RuntimeError, match="Invalid size"):
            bad.result(timeout=60)
        del bad                                          # the stored exception's traceback holds run_job's frame (and so the worker)
        again = pool.submit(_Job(_Req(prompt="a lighthouse 0", seed=100))).result(timeout=600)
        assert again == first[0]
        # mode switch sd15 -> sdxl on the worker thread: the old engine must actually be released
        assert pool.submit(_Job(switch_to="synthetic-sdxl")).result(timeout=900) == "synthetic-sdxl"
        assert isinstance(pool.worker, hip_worker.HipLcmSDXLWorker)
        assert not [k for k in hip_worker._ENGINES if k[0] == "sd15" and hip_worker._ENGINES[k]() is not None]
        x1 = pool.submit(_Job(_Req(prompt="a lighthouse 0", seed=100))).result(timeout=900)
        sdxl_mem = _mem() - base
        assert sdxl_mem < 5.135e9 * 1.9 + 2.5e9, f"SD1.5 engine still resident next to SDXL? {sdxl_mem / 1e9:.2f} GB"
        # ... and back: same bytes as before the round trip (fresh engine, fresh graphs, same numbers)
        assert pool.submit(_Job(switch_to="synthetic")).result(timeout=900) == "synthetic"
        back = [pool.submit(_Job(_Req(prompt=f"a lighthouse {i}", seed=100 + i))).result(timeout=600) for i in range(3)]
        assert back == first
        assert _mem() - base < sd15_mem + 0.3e9, "memory grows across mode switches"
        assert x1[0] != first[0][0]
    finally:
        pool.shutdown()
        if old_model is None:
            os.environ.pop("MODEL", None)
        else:
            os.environ["MODEL"] = old_model
    gc.collect()
    torch.cuda.empty_cache()
    left = _mem() - base
    assert left < 1.3e9, f"{left / 1e9:.2f} GB still allocated after the pool shut down (only the split-K workspace may stay)"
    assert not [k for k, v in hip_worker._ENGINES.items() if v() is not None]
