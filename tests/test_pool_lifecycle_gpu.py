"""The worker under the reference pool's lifecycle (backends/worker_pool.py:204-341), driven by tools/minipool.MiniPool -- the
pool-shaped harness that replays a session recorded from the reference's own WorkerPool: the worker is built by an injected
factory on the MAIN thread, jobs run on a separate worker thread, a mode switch tears the worker down ON that thread with the pool's own teardown -- ``del worker;
gc.collect(); torch.cuda.empty_cache()``, no ``close()`` (worker_pool.py:270-276) -- and builds the next mode's worker
there; a failing job leaves the worker usable (worker_pool.py:333-336).  Checked: results do not change across a
teardown / rebuild, and the unloaded mode's device memory really is returned."""
import gc
import os
import sys
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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import minipool  # noqa: E402  (the pool-shaped harness that replays a recording of the reference's WorkerPool:
#                                         tests/test_pool_batching.py::test_minipool_replays_the_reference_pool_recording)

MODES = {"synthetic": "synthetic", "synthetic-sdxl": "synthetic-sdxl"}


def _Job(req):
    return minipool.GenerationJob(req=req)


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
    pool = minipool.MiniPool(create_hip_worker, MODES, "synthetic", model_root=os.environ["MODEL_ROOT"])
    pool.submit = pool.submit_job
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
        assert pool.switch_mode("synthetic-sdxl").result(timeout=900) == {"mode": "synthetic-sdxl", "status": "switched"}
        assert pool.get_current_mode() == "synthetic-sdxl" and isinstance(pool._worker, hip_worker.HipLcmSDXLWorker)
        assert not [k for k in hip_worker._ENGINES if k[0] == "sd15" and hip_worker._ENGINES[k]() is not None]
        x1 = pool.submit(_Job(_Req(prompt="a lighthouse 0", seed=100))).result(timeout=900)
        sdxl_mem = _mem() - base
        assert sdxl_mem < 5.135e9 * 1.9 + 2.5e9, f"SD1.5 engine still resident next to SDXL? {sdxl_mem / 1e9:.2f} GB"
        # ... and back: same bytes as before the round trip (fresh engine, fresh graphs, same numbers)
        assert pool.switch_mode("synthetic").result(timeout=900)["status"] == "switched"
        assert pool.switch_mode("synthetic").result(timeout=900)["status"] == "already_loaded"
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
