"""Factory usable as ``WorkerPool(worker_factory=create_hip_worker)`` (backends/worker_pool.py:147-181);
mirrors ``create_cuda_worker`` / ``detect_worker_type`` (backends/worker_factory.py:17-100)."""
from __future__ import annotations

import json
import os


def detect_worker_type() -> str:
    """cross_attention_dim -> 'sd15' | 'sdxl' (backends/worker_factory.py:55-67)."""
    root = (os.environ.get("MODEL_ROOT") or "").strip()
    name = (os.environ.get("MODEL") or "").strip()
    if name.startswith("synthetic"):
        return "sdxl" if name.endswith("sdxl") else "sd15"
    if not root:
        raise RuntimeError("MODEL_ROOT environment variable is required")
    if not name:
        raise RuntimeError("MODEL environment variable is required")
    path = os.path.join(root, name)
    cfgp = os.path.join(path, "unet", "config.json")
    if os.path.isfile(path) and path.endswith(".safetensors"):
        # single file: read only the header, as utils/model_detector.py:232-284 does
        from safetensors import safe_open
        cad = None
        with safe_open(path, framework="pt") as f:
            for k in f.keys():
                if k.endswith("attn2.to_k.weight"):
                    cad = f.get_slice(k).get_shape()[1]
                    break
    elif os.path.exists(cfgp):
        with open(cfgp) as f:
            cad = json.load(f).get("cross_attention_dim")
    else:
        raise RuntimeError(f"Model not found: {path}")
    if cad in (2048, 1280):
        return "sdxl"
    if cad in (768, 1024):
        return "sd15"
    raise RuntimeError(f"Unknown cross_attention_dim={cad}")


def pick_device(worker_id: int, device_count: int, env=None) -> str:
    """Which GPU worker ``worker_id`` runs on.  The reference maps worker i to accelerator core i mod N
    (server/lcm_sr_server.py:140-152 for the NPU cores; one CUDA device otherwise, backends/cuda_worker.py:53): with
    ``LCM_DEVICES=all`` worker i takes ``cuda:{i % device_count}``, with ``LCM_DEVICES=0,2,5`` the i-th entry of that list
    (cyclically); without it every worker takes HIP_DEVICE / CUDA_DEVICE / cuda:0 as before (workers on one GPU share one
    resident engine).  Pure function of its arguments."""
    env = os.environ if env is None else env
    spec = (env.get("LCM_DEVICES") or "").strip().lower()
    if spec:
        if device_count <= 0:
            raise RuntimeError("LCM_DEVICES is set but no GPU is visible")
        if spec == "all":
            ids = list(range(device_count))
        else:
            try:
                ids = [int(x) for x in spec.replace(";", ",").split(",") if x.strip() != ""]
            except ValueError:
                raise RuntimeError(f"LCM_DEVICES={spec!r}: expected 'all' or a comma-separated list of device indices")
            bad = [i for i in ids if i < 0 or i >= device_count]
            if bad or not ids:
                raise RuntimeError(f"LCM_DEVICES={spec!r}: device indices {bad or ids} outside the {device_count} visible GPUs")
        return f"cuda:{ids[int(worker_id) % len(ids)]}"
    return (env.get("HIP_DEVICE") or env.get("CUDA_DEVICE") or "cuda:0").strip()


_POOL_QUEUE = None          # weakref to the queue the caller's single consumer thread takes jobs from (set_pool_queue)


def set_pool_queue(q, worker=None) -> None:
    """For a ``WorkerPool`` built by hand (dependency injection; ``get_worker_pool()`` is found without this): name its job
    queue ONCE -- ``set_pool_queue(pool.q, pool._worker)`` -- and every worker this factory creates afterwards (mode switches
    re-create it, backends/worker_pool.py:316-320) drains that queue into batched passes (``HipLcmWorker.run_job``).
    ``worker``: the one the pool's constructor already created.  ``q=None`` forgets the queue."""
    global _POOL_QUEUE
    import weakref
    _POOL_QUEUE = weakref.ref(q) if q is not None else None
    if worker is not None and hasattr(worker, "bind_queue"):
        worker.bind_queue(q)


def create_hip_worker(worker_id: int):
    kind = detect_worker_type()
    from .hip_worker import HipLcmSDXLWorker, HipLcmWorker
    if kind == "sdxl":
        w = HipLcmSDXLWorker(worker_id=worker_id)                    # backends/worker_factory.py:93
    else:
        w = HipLcmWorker(worker_id=worker_id)                        # backends/worker_factory.py:97
    q = _POOL_QUEUE() if _POOL_QUEUE is not None else None
    if q is not None:
        w.bind_queue(q)
    return w
