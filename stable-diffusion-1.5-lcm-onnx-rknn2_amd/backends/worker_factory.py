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


def create_hip_worker(worker_id: int):
    kind = detect_worker_type()
    from .hip_worker import HipLcmSDXLWorker, HipLcmWorker
    if kind == "sdxl":
        return HipLcmSDXLWorker(worker_id=worker_id)                 # backends/worker_factory.py:93
    return HipLcmWorker(worker_id=worker_id)                         # backends/worker_factory.py:97
