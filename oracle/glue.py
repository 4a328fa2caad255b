"""Pipeline glue restated from the reference (numpy, host side).  Oracle only."""
from __future__ import annotations

import numpy as np
import torch


def parse_size(size) -> tuple[int, int]:
    """'WxH' -> (width, height).  backends/cuda_worker.py:204-208 (error text)."""
    try:
        w_str, h_str = str(size).lower().split("x")
        return int(w_str), int(h_str)
    except Exception:
        raise RuntimeError(f"Invalid size '{size}', expected 'WIDTHxHEIGHT'")


def guidance_scale_embedding(w: np.ndarray, embedding_dim: int = 256, dtype=np.float32) -> np.ndarray:
    """backends/rknnlcm.py:651-677.  ``w`` = guidance_scale - 1 (rknnlcm.py:572)."""
    w = np.asarray(w, dtype=dtype) * 1000
    half = embedding_dim // 2
    f = np.exp(np.arange(half, dtype=dtype) * -(np.log(10000.0) / (half - 1)))
    e = w[:, None] * f[None, :]
    e = np.concatenate([np.sin(e), np.cos(e)], axis=1)
    if embedding_dim % 2 == 1:
        e = np.pad(e, [(0, 0), (0, 1)])
    return e


def prepare_latents(seed: int, height: int, width: int, n_noise: int, init_noise_sigma: float = 1.0):
    """Per-request RNG stream (backends/cuda_worker.py:212-213,
    backends/rknnlcm.py:423-447): one CPU generator seeded with ``seed``; draws in
    order latents[1,4,h,w] then one [1,4,h,w] per non-final step."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    shape = (1, 4, height // 8, width // 8)
    lat = torch.randn(shape, generator=g, dtype=torch.float32) * init_noise_sigma
    noise = [torch.randn(shape, generator=g, dtype=torch.float32) for _ in range(n_noise)]
    return lat, noise


def postprocess_u8(image: np.ndarray) -> np.ndarray:
    """NCHW float in ~[-1,1] -> NHWC uint8.  backends/rknnlcm.py:223,232-236,259."""
    x = np.clip(np.asarray(image) / 2 + 0.5, 0, 1)
    x = x.transpose((0, 2, 3, 1))
    return (x * 255).round().astype("uint8")


def downsample_to_8x8(lat: np.ndarray) -> np.ndarray:
    """[1,4,h,w] -> [1,4,8,8]; block mean when divisible (== adaptive_avg_pool2d,
    backends/cuda_worker.py:299), nearest otherwise.  backends/rknn_worker.py:223-248."""
    lat = np.asarray(lat)
    if lat.shape[0] != 1:
        lat = lat[:1]
    _, _, h, w = lat.shape
    if h == 8 and w == 8:
        return lat
    if h % 8 == 0 and w % 8 == 0:
        return lat.reshape(1, 4, 8, h // 8, 8, w // 8).mean(axis=(3, 5))
    ys = np.linspace(0, h - 1, 8).round().astype(np.int64)
    xs = np.linspace(0, w - 1, 8).round().astype(np.int64)
    return lat[:, :, ys][:, :, :, xs]


def latents_blob(lat: np.ndarray) -> bytes:
    """512-byte fp16 LE C-order [1,4,8,8] blob.  backends/cuda_worker.py:299-304."""
    t = torch.as_tensor(np.asarray(lat, dtype=np.float32))
    l8 = torch.nn.functional.adaptive_avg_pool2d(t, (8, 8)).to(torch.float16).contiguous()
    return l8.numpy().astype(np.float16, copy=False).tobytes(order="C")


def latent_to_nchw(x) -> np.ndarray:
    """Latents in any of the layouts the NPU runtime may hand back -> numpy NCHW (backends/rknn_worker.py:182-220): 4-D only;
    axis 1 of size 4 is taken as channels, else a last axis of size 4 (NHWC), else the first axis of size 4 moves to position 1.
    Pinned by tests/golden/worker_contract.npz (recorded from the reference)."""
    if x is None:
        raise ValueError("latent is None")
    if hasattr(x, "detach") and hasattr(x, "cpu") and hasattr(x, "numpy"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    if x.ndim != 4:
        raise ValueError(f"latent must be 4D, got shape={x.shape}")
    if x.shape[1] == 4:
        return x
    if x.shape[-1] == 4:
        return np.transpose(x, (0, 3, 1, 2))
    if 4 in x.shape:
        c = list(x.shape).index(4)
        axes = [a for a in range(4) if a != c]
        axes.insert(1, c)
        return np.transpose(x, axes)
    raise ValueError(f"cannot interpret latent layout, shape={x.shape}")


def check_size(width: int, height: int) -> None:
    """The size rule of the pipeline's check_inputs (backends/rknnlcm.py:380-381): both sides divisible by 8."""
    if height % 8 != 0 or width % 8 != 0:
        raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

