"""``HipLcmWorker`` -- the MI355X drop-in for the reference's ``DiffusersCudaWorker``
(backends/cuda_worker.py:20-304): same constructor, attributes, ``run_job`` /
``run_job_with_latents`` contract and error behaviour, with the arithmetic of
``self.pipe(...)`` (cuda_worker.py:221-229) replaced by the hipGraph of HIP kernels in
``pipeline.LcmHipPipeline``.

Env (as the reference, backends/cuda_worker.py:43-61):
  MODEL_ROOT, MODEL   checkpoint location (diffusers directory).  MODEL=synthetic (or
                      LCM_HIP_SYNTHETIC=1) selects seeded synthetic weights of the SD1.5 architecture --
                      no checkpoint ships with the reference.
  CUDA_DEVICE / HIP_DEVICE   default cuda:0 (torch's name for the HIP device)
  CUDA_DTYPE                 only fp16 is implemented on the HIP path (the reference's default)
"""
from __future__ import annotations

import io
import os
import struct
import zlib
from typing import Tuple

import numpy as np
import torch

from ..lib import LcmHipError
from ..pipeline import LcmHipPipeline
from ..prompt import HipPromptEncoder
from ..scheduler import LCMSchedule
from .. import weights as _weights


def parse_size(size) -> Tuple[int, int]:
    try:
        w_str, h_str = str(size).lower().split("x")
        return int(w_str), int(h_str)
    except Exception:
        raise RuntimeError(f"Invalid size '{size}', expected 'WIDTHxHEIGHT'")     # cuda_worker.py:204-208


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(rgb) -> bytes:
    """uint8 [H,W,3] -> PNG file bytes (the reference's ``img.save(buf, format="PNG")``, cuda_worker.py:234-239).

    With the sampler at ~22 ms the PIL encoder (40-70 ms for 512x512) would dominate run_job, so the file is
    written directly: scanline filter 2 ("Up", one vectorised numpy subtraction) + zlib level LCM_PNG_COMPRESS
    (default 1) in a single IDAT -- lossless, deterministic, ~4x faster at ~15 % larger files.
    LCM_PNG_ENCODER=pil restores the PIL path."""
    if os.environ.get("LCM_PNG_ENCODER", "").lower() == "pil":
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="PNG", compress_level=int(os.environ.get("LCM_PNG_COMPRESS", "6")))
        return buf.getvalue()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    if c != 3:
        raise ValueError(f"encode_png expects RGB, got {c} channels")
    flat = rgb.reshape(h, w * 3)
    raw = np.empty((h, 1 + w * 3), np.uint8)
    raw[:, 0] = 2                      # filter type Up
    raw[0, 1:] = flat[0]
    np.subtract(flat[1:], flat[:-1], out=raw[1:, 1:])      # uint8 wrap-around == mod 256
    comp = zlib.compress(raw.tobytes(), int(os.environ.get("LCM_PNG_COMPRESS", "1")))
    return (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + _png_chunk(b"IDAT", comp) + _png_chunk(b"IEND", b""))


class HipLcmWorker:
    def __init__(self, worker_id: int):
        self.worker_id = worker_id
        model_root = (os.environ.get("MODEL_ROOT") or "").strip()
        model_name = (os.environ.get("MODEL") or "").strip()
        synthetic = model_name == "synthetic" or os.environ.get("LCM_HIP_SYNTHETIC", "0").lower() in ("1", "true", "yes", "on")
        if not synthetic:
            if not model_root:
                raise RuntimeError("MODEL_ROOT is required for BACKEND=hip")
            if not model_name:
                raise RuntimeError("MODEL is required for BACKEND=hip")
        dtype_str = os.environ.get("CUDA_DTYPE", "fp16").lower().strip()
        if dtype_str != "fp16":
            raise RuntimeError(f"CUDA_DTYPE={dtype_str}: the HIP backend computes in fp16 (fp32 accumulate) only")
        device = (os.environ.get("HIP_DEVICE") or os.environ.get("CUDA_DEVICE") or "cuda:0").strip()
        if not torch.cuda.is_available():
            raise LcmHipError("HipLcmWorker needs an MI355X; no CPU fallback exists on this path")
        sched = LCMSchedule()
        if synthetic:
            usd, ucfg = _weights.synthetic_unet(), None
            vsd, vcfg = _weights.synthetic_vae(), None
            ckpt_root, clip_sd = None, None
            ckpt = "synthetic"
        else:
            ckpt = os.path.join(model_root, model_name)
            clip_sd = None
            if os.path.isdir(ckpt) and os.path.exists(os.path.join(ckpt, "model_index.json")):
                usd, ucfg, vsd, vcfg = _weights.load_diffusers_dir(ckpt)          # cuda_worker.py:66-77
                format_name = "diffusers"
            elif os.path.isfile(ckpt) and ckpt.endswith(".safetensors"):
                usd, ucfg, vsd, vcfg, clip_sd = _weights.load_single_file(ckpt)   # cuda_worker.py:78-85
                format_name = "single-file"
            else:
                raise RuntimeError(f"{ckpt}: expected a diffusers directory (model_index.json) or a .safetensors file "
                                   "(.ckpt pickles are not loaded: they execute code)")
            if int(ucfg.get("cross_attention_dim", 768)) not in (768, 1024):
                raise RuntimeError(f"cross_attention_dim={ucfg['cross_attention_dim']}: SDXL UNets are not "
                                   "supported by the SD1.5 HIP worker")                # cuda_worker.py:114-116
            sched = LCMSchedule.from_config_file(os.path.join(ckpt, "scheduler", "scheduler_config.json"))
            ckpt_root = ckpt if format_name == "diffusers" else None
        self.pipe = LcmHipPipeline(usd, vsd, ucfg, vcfg, device=device, schedule=sched)
        # CLIP text encoder on the same kernels (checkpoint text_encoder/ when present, else synthetic CLIP-L weights)
        with torch.cuda.stream(self.pipe.stream):
            self._encode = HipPromptEncoder(device, ckpt_root, clip_sd)
        if self._encode.enc.D != self.pipe.unet.ctx_dim:
            raise RuntimeError(f"text encoder width {self._encode.enc.D} != UNet cross_attention_dim {self.pipe.unet.ctx_dim}")
        self.device = device
        self.dtype = torch.float16
        self._neg = None
        print(f"[hip] worker {worker_id} loaded: {os.path.basename(ckpt)} on {device} dtype=fp16 "
              f"unet={self.pipe.unet.weight_bytes() / 1e9:.2f}GB vae={self.pipe.vae.weight_bytes() / 1e9:.2f}GB "
              f"clip={self._encode.enc.weight_bytes() / 1e9:.2f}GB ({self._encode.source})")

    # ------------------------------------------------------------------------------------------
    def _generate(self, job):
        req = job.req
        width, height = parse_size(req.size)
        seed = int(req.seed) if getattr(req, "seed", None) is not None else int(torch.randint(0, 100_000_000, (1,)).item())
        sl = getattr(req, "style_lora", None)
        level = int(getattr(sl, "level", 0) or 0) if sl else 0
        if level > 0 and getattr(sl, "style", None):
            # LoRA style adapters are a later row; requests run unstyled.
            print(f"[hip] style_lora '{sl.style}' level {level} ignored (LoRA merge not implemented)")
        g = float(req.guidance_scale)
        neg = None
        with torch.cuda.stream(self.pipe.stream):
            pe = self._encode([req.prompt])
            if g > 1.0 and not self.pipe.unet.has_cond:
                neg = self._encode([""])
        out = self.pipe.generate(pe, [seed], width, height, int(req.num_inference_steps), g, negative_embeds=neg)
        return out, seed

    def run_job(self, job) -> Tuple[bytes, int]:
        out, seed = self._generate(job)
        return encode_png(out["rgb"][0]), seed

    def run_job_with_latents(self, job) -> Tuple[bytes, int, bytes]:
        # The reference re-runs the whole pipeline for the latents (cuda_worker.py:255-283); the sampler is
        # deterministic in the seed, so the same pass's final latents are identical and are pooled on device.
        out, seed = self._generate(job)
        return encode_png(out["rgb"][0]), seed, out["pool8"][:1].tobytes(order="C")

    def close(self):
        pipe = getattr(self, "pipe", None)
        if pipe is not None:
            pipe.drop_plans()
            self.pipe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
