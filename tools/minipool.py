"""A pool-shaped single-consumer harness for tests and ``bench.py``'s ``extra_worker`` leg -- NOT part of the product.

The reference's control plane (``WorkerPool``, backends/worker_pool.py:135-419) stays the reference's; this is the smallest
stand-in with the same shape towards a worker, so that the worker can be measured and tested on a GPU box, where
``/root/reference`` does not exist:

  * the worker is built by an injected factory, called as ``factory(worker_id=0)`` on the thread that loads the mode
    (worker_pool.py:228) after ``MODEL_ROOT`` / ``MODEL`` were put into the environment (:221-222);
  * ONE consumer thread named ``WorkerThread`` takes jobs from ``self.q`` (a bounded ``queue.Queue``) and runs
    ``job.execute(worker)`` synchronously (:294-341); a job's result / exception goes to ``job.fut`` unless the future is
    already resolved (:323-336); ``task_done`` after every job (:337-339);
  * a mode-switch job for the current mode is a no-op (:309-314); otherwise the old worker is dropped with ``del; gc.collect();
    torch.cuda.empty_cache()`` -- no ``close()`` (:258-278) -- and the next one is built on the consumer thread;
  * ``submit_job`` is ``put_nowait`` (a full queue raises ``queue.Full``, :343-366); ``shutdown`` joins the queue, stops the
    thread, unloads the worker (:396-418).

That it behaves like the real pool is itself tested: ``tests/pool_scenario.py`` is run against the reference's ``WorkerPool``
in the build container (``tests/golden/make_contract_golden.py`` -> ``worker_contract.json["pool"]``) and replayed against this
class by ``tests/test_host_logic.py::test_minipool_replays_the_reference_pool_recording``.
"""
from __future__ import annotations

import gc
import os
import queue
import threading
from concurrent.futures import Future
from typing import Any, Callable, Dict, Optional


class Job:
    def __init__(self):
        self.fut: Future = Future()

    def execute(self, worker) -> Any:
        raise NotImplementedError


class GenerationJob(Job):
    """Carries ``req``; executed as ``worker.run_job(job)`` (worker_pool.py:75-88)."""

    def __init__(self, req):
        super().__init__()
        self.req = req

    def execute(self, worker):
        if worker is None:
            raise RuntimeError("No worker available for generation")
        return worker.run_job(self)


class ModeSwitchJob(Job):
    def __init__(self, target_mode: str, on_complete: Optional[Callable] = None):
        super().__init__()
        self.target_mode, self.on_complete = target_mode, on_complete

    def execute(self, worker):
        if self.on_complete:
            self.on_complete(self.target_mode)
        return {"mode": self.target_mode, "status": "switched"}


class CustomJob(Job):
    def __init__(self, handler: Callable, args: tuple = (), kwargs: Optional[dict] = None):
        super().__init__()
        self.handler, self.args, self.kwargs = handler, args, kwargs or {}

    def execute(self, worker):
        return self.handler(*self.args, **self.kwargs)


def _empty_cache():
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:
        pass


class MiniPool:
    def __init__(self, worker_factory, modes: Dict[str, str], default_mode: str, model_root: str = "/models", queue_max: int = 64,
                 on_event: Optional[Callable] = None):
        """``modes``: mode name -> MODEL value.  ``on_event(name, **info)``: optional hook the scenario recorder uses."""
        self.queue_max = queue_max
        self.q: "queue.Queue[Job]" = queue.Queue(maxsize=queue_max)
        self._stop = threading.Event()
        self._worker = None
        self._worker_thread: Optional[threading.Thread] = None
        self._current_mode: Optional[str] = None
        self._worker_factory, self._modes, self._model_root = worker_factory, dict(modes), model_root
        self._on_event = on_event or (lambda *a, **k: None)
        self._load_mode(default_mode)

    # -- lifecycle ------------------------------------------------------------------------------------------------
    def _load_mode(self, mode_name: str):
        model = self._modes[mode_name]                   # KeyError for unknown modes, like ModeConfigManager.get_mode
        if self._worker is not None:
            self._unload_current_worker()
        os.environ["MODEL_ROOT"] = self._model_root
        os.environ["MODEL"] = model
        self._worker = self._worker_factory(worker_id=0)
        self._on_event("register", mode=mode_name)
        self._current_mode = mode_name
        self._start_worker_thread()

    def _unload_current_worker(self):
        if self._worker is None:
            return
        if self._current_mode:
            self._on_event("unregister", mode=self._current_mode)
        del self._worker
        self._worker = None
        gc.collect()
        _empty_cache()

    def _start_worker_thread(self):
        if self._worker_thread is not None and self._worker_thread.is_alive():
            return
        self._worker_thread = threading.Thread(target=self._worker_loop, daemon=True, name="WorkerThread")
        self._worker_thread.start()

    def _worker_loop(self):
        while not self._stop.is_set():
            try:
                job = self.q.get(timeout=0.2)
            except queue.Empty:
                continue
            try:
                if isinstance(job, ModeSwitchJob):
                    if self._current_mode == job.target_mode:
                        result = {"mode": job.target_mode, "status": "already_loaded"}
                    else:
                        result = job.execute(self._worker)
                        self._load_mode(job.target_mode)
                else:
                    result = job.execute(self._worker)
                if not job.fut.done():
                    job.fut.set_result(result)
            except Exception as e:
                if not job.fut.done():
                    job.fut.set_exception(e)
            finally:
                self.q.task_done()

    # -- API ------------------------------------------------------------------------------------------------------
    def submit_job(self, job: Job) -> Future:
        try:
            self.q.put_nowait(job)
            return job.fut
        except queue.Full:
            raise queue.Full(f"Job queue full (max: {self.queue_max}). Try again later or increase QUEUE_MAX.")

    def switch_mode(self, mode_name: str) -> Future:
        self._modes[mode_name]
        return self.submit_job(ModeSwitchJob(target_mode=mode_name))

    def get_current_mode(self):
        return self._current_mode

    def get_queue_size(self) -> int:
        return self.q.qsize()

    def shutdown(self):
        self.q.join()
        self._stop.set()
        if self._worker_thread and self._worker_thread.is_alive():
            self._worker_thread.join(timeout=5.0)
        self._unload_current_worker()
