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
  CUDA_DTYPE                 fp16 (default).  bf16 / fp32 -- which the reference honours -- are REFUSED unless LCM_HIP_DTYPE=fp16
                             says to run them in this backend's one arithmetic (fp16 operands, fp32 accumulation)
"""
from __future__ import annotations

import io
import os
import struct
import threading
import weakref
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


_PNG_POOL = None
_PNG_POOL_LOCK = threading.Lock()


def _png_pool(n):
    """Stripe-deflate threads.  The stripe COUNT of an image (and so its bytes) follows LCM_PNG_THREADS alone; the pool is
    wider (up to the host's cores, at most 16) so that the images of a drained batch deflate side by side."""
    global _PNG_POOL
    n = max(n, min(16, os.cpu_count() or 1))
    with _PNG_POOL_LOCK:
        if _PNG_POOL is None or _PNG_POOL._max_workers < n:
            from concurrent.futures import ThreadPoolExecutor
            _PNG_POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix="lcm-png")
        return _PNG_POOL


_FINISH_POOL = None


def _finish_pool():
    """Threads that finish the jobs a ``run_job`` call drained from the pool's queue (PNG encode, set the job's future,
    ``task_done``).  Separate from the stripe pool: a finisher blocks on its image's stripes."""
    global _FINISH_POOL
    with _PNG_POOL_LOCK:
        if _FINISH_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _FINISH_POOL = ThreadPoolExecutor(max_workers=min(16, max(4, os.cpu_count() or 4)), thread_name_prefix="lcm-finish")
        return _FINISH_POOL


def _deflate_stripe(args):
    buf, level, last = args
    co = zlib.compressobj(level, zlib.DEFLATED, -15)          # raw deflate: the stripes share one zlib wrapper
    return co.compress(buf) + co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH)


def encode_png(rgb) -> bytes:
    """uint8 [H,W,3] -> PNG file bytes (the reference's ``img.save(buf, format="PNG")``, cuda_worker.py:234-239).

    With the sampler at ~20 ms the PIL encoder (40-70 ms for 512x512) would dominate run_job, so the file is written by the
    library's own writer (csrc/png.cpp, ``lcm_png_encode_rgb8``): scanline filter "Up", one dynamic-Huffman deflate block per
    stripe over literals and distance-1 runs, LCM_PNG_THREADS stripes (default 8) compressed in parallel, Adler-32 / CRC-32
    combined from the stripes -- lossless, deterministic (the bytes depend on the image and the stripe count, not on timing),
    0.4 ms for 512x512 where zlib level 1 on 8 threads took 3 ms and PIL 40-70.
    LCM_PNG_ENCODER=zlib: the round-3 writer (numpy filter + zlib level LCM_PNG_COMPRESS on a Python thread pool, pigz-style
    stripes); =pil: PIL."""
    enc = os.environ.get("LCM_PNG_ENCODER", "").lower()
    if enc == "pil":
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="PNG", compress_level=int(os.environ.get("LCM_PNG_COMPRESS", "6")))
        return buf.getvalue()
    if enc != "zlib":
        import ctypes
        from .. import lib as _lib
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, c = rgb.shape
        if c != 3:
            raise ValueError(f"encode_png expects RGB, got {c} channels")
        L = _lib.load()
        nthr = max(1, int(os.environ.get("LCM_PNG_THREADS", "8")))
        stripes = min(nthr, max(1, h // 64))                   # stripes of at least 64 scanlines
        cap = int(L.lcm_png_bound(w, h, stripes))
        out = np.empty(cap, np.uint8)
        n = ctypes.c_longlong(0)
        _lib.check(L.lcm_png_encode_rgb8(rgb.ctypes.data, w, h, w * 3, stripes, out.ctypes.data, cap, ctypes.byref(n)),
                   "lcm_png_encode_rgb8")
        return out[:n.value].tobytes()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    if c != 3:
        raise ValueError(f"encode_png expects RGB, got {c} channels")
    flat = rgb.reshape(h, w * 3)
    raw = np.empty((h, 1 + w * 3), np.uint8)
    raw[:, 0] = 2                      # filter type Up
    raw[0, 1:] = flat[0]
    np.subtract(flat[1:], flat[:-1], out=raw[1:, 1:])      # uint8 wrap-around == mod 256
    level = int(os.environ.get("LCM_PNG_COMPRESS", "1"))
    nthr = max(1, int(os.environ.get("LCM_PNG_THREADS", "8")))
    nstripes = min(nthr, max(1, h // 64))                  # stripes of at least 64 scanlines
    if nstripes == 1:
        comp = zlib.compress(raw.tobytes(), level)
    else:
        mv = memoryview(raw).cast("B")
        row = 1 + w * 3
        cuts = [(h * i // nstripes) * row for i in range(nstripes + 1)]
        parts = list(_png_pool(nthr).map(_deflate_stripe, [(mv[cuts[i]:cuts[i + 1]], level, i == nstripes - 1) for i in range(nstripes)]))
        comp = b"\x78\x01" + b"".join(parts) + struct.pack(">I", zlib.adler32(mv) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + _png_chunk(b"IDAT", comp) + _png_chunk(b"IEND", b""))


class _Engine:
    """Everything resident for one (family, device, checkpoint): the pipeline (weights, launch plans, captured graphs), the
    text encoders, the style adapters and the micro-batching dispatcher.  The reference builds one pipeline -- and one
    weight copy -- per worker object (``DiffusersCudaWorker.__init__``, backends/cuda_worker.py:41-121; its ``WorkerPool`` creates
    exactly one, ``worker_id=0``, backends/worker_pool.py:228; the legacy ``PipelineService`` one per ``NUM_WORKERS``,
    forced to 1 on CUDA, server/lcm_sr_server.py:190-193).  Here the workers a caller creates for one GPU share one engine,
    so N threads blocking in ``run_job`` become batched passes -- and behind the single-consumer ``WorkerPool`` the batch comes
    from draining the pool's queue (``HipLcmWorker._drain``).  LCM_SHARE_ENGINE=0: one engine per worker.

    Lifetime: workers hold the engine, the engine never holds a worker, the registry and the dispatcher thread hold it
    weakly.  The pool's teardown -- ``del worker; gc.collect(); torch.cuda.empty_cache()`` (backends/worker_pool.py:270-276),
    no ``close()`` -- therefore releases the weights, plans, graphs and pinned buffers of the mode being unloaded."""

    def __init__(self, family_cls):
        self.family_cls = family_cls
        self.refs = 0
        self.batcher = None
        self.batch_sizes = (1,)
        self.active_style = None
        self.pipe = None
        self.encode = None           # SD1.5: HipPromptEncoder
        self.enc, self.tok = [], []  # SDXL: two encoders / tokenizers
        self.styles = {}
        self.device = None
        self.dtype = torch.float16
        self.timing = [] if os.environ.get("LCM_WORKER_TIMING", "0") == "1" else None
        # Lanes: sampler instances of the pipeline that run concurrently (own stream, scratch, graphs, text-encoder
        # buffers; shared weights).  Two batch-1 passes in flight take ~1.4x the time of one, so at low load -- where
        # nothing queues up to be coalesced -- the second lane is worth ~45 % more images/s.  LCM_LANES=1: one at a time.
        self.n_lanes = max(1, int(os.environ.get("LCM_LANES", "2") or 1))
        self.lane_locks = [threading.Lock() for _ in range(self.n_lanes)]
        self.lane_encoders = {}      # lane -> per-lane views of the text encoders
        self._style_cv = threading.Condition()
        self._style_users = 0        # passes currently running with `active_style` merged into the shared weights

    # ---- style LoRAs (backends/cuda_worker.py:123-196) -----------------------------------------
    def _want_style(self, style_id, level):
        from .styles import STYLE_REGISTRY
        if style_id and int(level) > 0 and style_id in self.styles:
            return (style_id, STYLE_REGISTRY[style_id].weight_for(level))
        return None

    def apply_style(self, style_id, level, stream=None):
        """Exclusive style selection; level 0 / unknown style = off.  Re-merges only when the selection changes."""
        want = self._want_style(style_id, level)
        if want == self.active_style:
            return
        with torch.cuda.stream(stream or self.pipe.stream):
            if self.active_style is not None and (want is None or want[0] != self.active_style[0]):
                self.styles[self.active_style[0]].apply(0.0)
            if want is not None:
                self.styles[want[0]].apply(want[1])
            torch.cuda.current_stream().synchronize()     # the merge must have landed before any lane replays a graph
        self.active_style = want

    def _enter_style(self, style_id, level, stream):
        """The merged weights are shared by all lanes: a pass may start when the style it needs is the merged one, or
        when no other pass is running (then it re-merges)."""
        want = self._want_style(style_id, level)
        with self._style_cv:
            while True:
                if want == self.active_style:
                    break
                if self._style_users == 0:
                    self.apply_style(style_id, level, stream)
                    break
                self._style_cv.wait()
            self._style_users += 1

    def _leave_style(self):
        with self._style_cv:
            self._style_users -= 1
            self._style_cv.notify_all()

    def _lane_encoders(self, lane):
        """(encode callable | None, [encoder views]) of a lane: same weights, activation buffers of their own."""
        e = self.lane_encoders.get(lane)
        if e is None:
            if lane == 0:
                e = (self.encode, list(self.enc))
            else:
                enc = None
                if self.encode is not None:
                    import copy
                    enc = copy.copy(self.encode)
                    enc.enc = self.encode.enc.view()
                e = (enc, [x.view() for x in self.enc])
            self.lane_encoders[lane] = e
        return e

    def run_batch(self, key, items, lane=0):
        """One batched sampler pass for ``items`` = [(req, seed[, noise])], all of ``key``, on lane ``lane``;
        -> per-item (rgb, pool8 row)."""
        width, height, steps, g, style_id, level = key
        import time as _t
        t0 = _t.perf_counter()
        lane = lane % self.n_lanes
        with self.lane_locks[lane]:                  # a lane (stream, scratch, graphs) runs one pass at a time
            pipe = self.pipe
            if pipe is None:
                raise RuntimeError("worker engine is closed")
            stream = pipe.lane(lane).stream
            self._enter_style(style_id, level, stream)   # lazy: no re-merge while consecutive batches use the same style
            try:
                reqs = [it[0] for it in items]
                with torch.cuda.stream(stream):
                    pe, kw = self.family_cls._conditioning(self, reqs, width, height, g, lane)
                t1 = _t.perf_counter()
                noises = [it[2] for it in items] if all(len(it) > 2 and it[2] is not None for it in items) else None
                out = pipe.generate(pe, [it[1] for it in items], width, height, steps, g, noises=noises, lane=lane, **kw)
            finally:
                self._leave_style()
        t2 = _t.perf_counter()
        if self.timing is not None:                  # LCM_WORKER_TIMING=1: (batch, conditioning s, sampler call s, end time)
            self.timing.append((len(items), t1 - t0, t2 - t1, t2))
        return [(out["rgb"][i], out["pool8"][i:i + 1]) for i in range(len(items))]

    def start_batcher(self):
        mb = int(os.environ.get("LCM_MICROBATCH", "8") or 0)
        if mb <= 1 and self.n_lanes <= 1:
            return
        from .batching import MicroBatcher
        ref = weakref.ref(self)

        def dispatch(key, items, lane=0):            # the dispatcher threads must not keep the engine alive
            eng = ref()
            if eng is None:
                raise RuntimeError("worker engine is gone")
            return eng.run_batch(key, items, lane)
        self.batcher = MicroBatcher(dispatch, max_batch=max(mb, 1), window_ms=float(os.environ.get("LCM_MICROBATCH_WINDOW_MS", "0") or 0),
                                    lanes=self.n_lanes)
        self.batch_sizes = tuple(self.batcher.sizes)
        weakref.finalize(self, self.batcher.close)   # engine collected without close(): stop the dispatcher threads

    def release(self) -> bool:
        """One worker less; the last one out shuts the engine down.  -> True when it did."""
        with _ENGINES_LOCK:
            self.refs -= 1
            if self.refs > 0:
                return False
            for k in [k for k, v in _ENGINES.items() if v() is self or v() is None]:
                del _ENGINES[k]
        self.shutdown()
        return True

    def shutdown(self):
        b, self.batcher = self.batcher, None
        if b is not None:
            b.close()
        for lk in self.lane_locks:
            lk.acquire()
        try:
            pipe, self.pipe = self.pipe, None
            if pipe is not None:
                pipe.close()                          # graphs dropped, split-K workspaces unregistered
            self.encode, self.enc, self.tok, self.styles, self.active_style = None, [], [], {}, None
            self.lane_encoders = {}
        finally:
            for lk in self.lane_locks:
                lk.release()


_ENGINES: dict = {}              # key -> weakref.ref(_Engine)
_ENGINES_LOCK = threading.Lock()


class HipLcmWorker:
    """SD1.5-family worker (drop-in for DiffusersCudaWorker, backends/cuda_worker.py:20-304)."""

    FAMILY = "sd15"

    def __init__(self, worker_id: int):
        self.worker_id = worker_id
        self._engine = None
        model_root = (os.environ.get("MODEL_ROOT") or "").strip()
        model_name = (os.environ.get("MODEL") or "").strip()
        synthetic = model_name.startswith("synthetic") or os.environ.get("LCM_HIP_SYNTHETIC", "0").lower() in ("1", "true", "yes", "on")
        if not synthetic:
            if not model_root:
                raise RuntimeError("MODEL_ROOT is required for BACKEND=hip")
            if not model_name:
                raise RuntimeError("MODEL is required for BACKEND=hip")
        dtype_str = os.environ.get("CUDA_DTYPE", "fp16").lower().strip()
        if dtype_str not in ("fp16", "bf16", "fp32"):
            raise RuntimeError(f"Unknown CUDA_DTYPE={dtype_str}, expected fp16, bf16 or fp32")      # cuda_worker.py:55-61
        if dtype_str != "fp16" and os.environ.get("LCM_HIP_DTYPE", "").lower().strip() != "fp16":
            # The reference honours bf16 / fp32 (torch dtype of the whole pipeline, cuda_worker.py:55-61).  This backend has ONE
            # arithmetic: fp16 operands, fp32 accumulation, fp32 sampler state -- inside north_star's tolerance of the fp32 pipeline
            # (max |delta| < 1e-2 on the decoded image; measured ~2e-3), but not what the setting asks for.  So it is refused,
            # not silently reinterpreted; LCM_HIP_DTYPE=fp16 states the override explicitly.
            raise RuntimeError(f"CUDA_DTYPE={dtype_str} is not available with BACKEND=hip: the HIP kernels compute with fp16 operands and "
                               f"fp32 accumulation only.  Set CUDA_DTYPE=fp16, or LCM_HIP_DTYPE=fp16 to run this mode in that arithmetic "
                               f"(max |delta| vs the fp32 pipeline < 1e-2 on the decoded image)")
        if not torch.cuda.is_available():
            raise LcmHipError("HipLcmWorker needs an MI355X; no CPU fallback exists on this path")
        from .worker_factory import pick_device
        device = pick_device(worker_id, torch.cuda.device_count())      # LCM_DEVICES=all: worker i -> GPU i mod N
        share = os.environ.get("LCM_SHARE_ENGINE", "1").lower() not in ("0", "false", "no", "off")
        from .styles import STYLE_REGISTRY
        ekey = (self.FAMILY, device, "synthetic" if synthetic else os.path.join(model_root, model_name),
                tuple(sorted((sid, sd.path()) for sid, sd in STYLE_REGISTRY.items())))
        with _ENGINES_LOCK:
            ref = _ENGINES.get(ekey) if share else None
            eng = ref() if ref is not None else None
            if eng is not None and eng.pipe is not None:
                eng.refs += 1
            else:
                eng = None
        if eng is not None:                          # a sibling worker already holds this checkpoint on this GPU
            self._engine = eng
            print(f"[hip] worker {worker_id} ({self.FAMILY}) attached to the resident engine on {device} ({eng.refs} workers)")
            return
        sched = LCMSchedule()
        clip_sd = None
        if synthetic:
            usd, ucfg, vsd, vcfg = self._synthetic_weights()
            ckpt_root, ckpt, format_name = None, "synthetic", "synthetic"
        else:
            ckpt = os.path.join(model_root, model_name)
            if os.path.isdir(ckpt) and os.path.exists(os.path.join(ckpt, "model_index.json")):
                usd, ucfg, vsd, vcfg = _weights.load_diffusers_dir(ckpt)          # cuda_worker.py:66-77
                format_name = "diffusers"
            elif os.path.isfile(ckpt) and ckpt.endswith(".safetensors"):
                loader = _weights.load_single_file if self.FAMILY == "sd15" else _weights.load_single_file_sdxl
                usd, ucfg, vsd, vcfg, clip_sd = loader(ckpt)                       # cuda_worker.py:78-85 / :330-352
                format_name = "single-file"
            else:
                raise RuntimeError(f"{ckpt}: expected a diffusers directory (model_index.json) or a .safetensors file"
                                   " (.ckpt pickles are not loaded: they execute code)")
            cad = int(ucfg.get("cross_attention_dim", 768))
            if (cad in (2048, 1280)) != (self.FAMILY == "sdxl"):
                raise RuntimeError(f"Loaded UNet with cross_attention_dim={cad} into the {self.FAMILY} worker "
                                   "(SD1.5: 768/1024, SDXL: 2048)")                  # cuda_worker.py:114-116
            sched = LCMSchedule.from_config_file(os.path.join(ckpt, "scheduler", "scheduler_config.json"))
            ckpt_root = ckpt if format_name == "diffusers" else None
        if self.FAMILY == "sd15" and vcfg is not None:
            # only the SDXL pipeline honours the VAE's force_upcast (StableDiffusionXLPipeline.upcast_vae); SD1.5 decodes in
            # the pipeline dtype whatever the checkpoint's vae/config.json says
            vcfg = dict(vcfg, force_upcast=False, residual_scale=1.0)
        eng = _Engine(type(self))
        eng.device = device
        eng.pipe = LcmHipPipeline(usd, vsd, ucfg, vcfg, device=device, schedule=sched)
        with torch.cuda.stream(eng.pipe.stream):
            self._load_text_encoders(eng, device, ckpt_root, clip_sd)
        eng.refs = 1
        self._engine = eng
        self._load_styles(eng)
        eng.start_batcher()
        if share:
            with _ENGINES_LOCK:
                _ENGINES[ekey] = weakref.ref(eng)
        print(f"[hip] worker {worker_id} ({self.FAMILY}) loaded: {os.path.basename(ckpt)} ({format_name}) on {device} dtype=fp16 "
              f"unet={eng.pipe.unet.weight_bytes() / 1e9:.2f}GB vae={eng.pipe.vae.weight_bytes() / 1e9:.2f}GB "
              f"text={self._text_bytes(eng) / 1e9:.2f}GB")

    # ---- the reference's attributes (tests/test_sdxl_worker.py:118-130), served from the shared engine --------------
    @property
    def pipe(self):
        return self._engine.pipe if self._engine is not None else None

    @property
    def device(self):
        return self._engine.device if self._engine is not None else None

    @property
    def dtype(self):
        return torch.float16

    @property
    def _styles(self):
        return self._engine.styles if self._engine is not None else {}

    @property
    def _active_style(self):
        return self._engine.active_style

    # ---- style LoRAs (backends/cuda_worker.py:123-196) -----------------------------------------
    @classmethod
    def _load_styles(cls, eng):
        from ..lora import StyleAdapters
        from .styles import STYLE_REGISTRY
        cad = int(eng.pipe.unet.ctx_dim)
        for sid, sd in STYLE_REGISTRY.items():
            if sd.required_cross_attention_dim is not None and int(sd.required_cross_attention_dim) != cad:
                print(f"[hip] skip style '{sid}': incompatible cross_attention_dim (model={cad} style={sd.required_cross_attention_dim})")
                continue
            path = sd.path()
            if not os.path.exists(path):
                print(f"[hip] style '{sid}': LoRA file {path} not found; style disabled")
                continue
            try:
                from safetensors.torch import load_file
                with torch.cuda.stream(eng.pipe.stream):
                    st = StyleAdapters(eng.pipe.unet, cls._text_encoders(eng), load_file(path))
                eng.styles[sid] = st
                print(f"[hip] loaded style LoRA: {sid} -> {path} ({len(st.modules)} UNet + {st.text_modules} text-encoder modules, "
                      f"{len(st.skipped)} tensors skipped, {st.nbytes() / 1e6:.0f} MB)")
            except Exception as e:
                print(f"[hip] FAILED to load style LoRA {sid}: {e!r}")

    def _apply_style(self, style_id, level):
        """The reference's hook (backends/cuda_worker.py:149-196).  The merged weights are shared by every lane, so -- like
        ``_enter_style`` -- the re-merge waits until no pass is in flight: weights are never rewritten under a running graph."""
        eng = self._engine
        with eng._style_cv:
            while eng._style_users > 0 and eng._want_style(style_id, level) != eng.active_style:
                eng._style_cv.wait()
            eng.apply_style(style_id, level)

    # ---- family hooks (operate on the engine: no worker object is ever captured by shared state) ----------------------
    def _synthetic_weights(self):
        return _weights.synthetic_unet(), None, _weights.synthetic_vae(), None

    @staticmethod
    def _load_text_encoders(eng, device, ckpt_root, clip_sd):
        # CLIP text encoder on the same kernels (checkpoint text_encoder/ when present, else synthetic CLIP-L weights)
        eng.encode = HipPromptEncoder(device, ckpt_root, clip_sd)
        if eng.encode.enc.D != eng.pipe.unet.ctx_dim:
            raise RuntimeError(f"text encoder width {eng.encode.enc.D} != UNet cross_attention_dim {eng.pipe.unet.ctx_dim}")

    @staticmethod
    def _text_encoders(eng):
        return [eng.encode.enc]

    @staticmethod
    def _text_bytes(eng):
        return eng.encode.enc.weight_bytes()

    @staticmethod
    def _conditioning(eng, reqs, width, height, guidance, lane=0):
        """-> (prompt_embeds [B,77,ctx], kwargs for LcmHipPipeline.generate) for B requests of one batch."""
        B = len(reqs)
        encode, _ = eng._lane_encoders(lane)
        pe = encode([r.prompt for r in reqs])
        neg = None
        if guidance > 1.0 and not eng.pipe.unet.has_cond:
            neg = encode([""]).expand(B, -1, -1)
        return pe, dict(negative_embeds=neg)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _job_key(req):
        """What must agree for jobs to share one batched pass: geometry, step count, guidance and style merge."""
        width, height = parse_size(req.size)
        sl = getattr(req, "style_lora", None)
        style_id = getattr(sl, "style", None) if sl else None
        level = int(getattr(sl, "level", 0) or 0) if sl else 0
        if not style_id or level <= 0:
            style_id, level = None, 0
        return (width, height, int(req.num_inference_steps), float(req.guidance_scale), style_id, level)

    def _run_batch(self, key, items):
        return self._engine.run_batch(key, items)

    def _prepare(self, req, key):
        """-> (req, seed, noise): the seed policy of cuda_worker.py:210-213 and the request's RNG stream."""
        eng = self._engine
        seed = int(req.seed) if getattr(req, "seed", None) is not None else int(torch.randint(0, 100_000_000, (1,)).item())
        # the request's RNG stream (initial latents, then one draw per remaining step) is drawn HERE, on the caller's
        # thread: pool threads do it in parallel and the GPU dispatcher's serial path shrinks by ~0.5 ms per request
        from ..pipeline import draw_noise
        noise = None
        if key[0] % 8 == 0 and key[1] % 8 == 0 and key[0] > 0 and key[1] > 0 and key[2] >= 1:
            noise = draw_noise(seed, key[1] // 8, key[0] // 8, key[2] - 1, eng.pipe.sched.init_noise_sigma)
        return (req, seed, noise)

    def _submit(self, job):
        eng = self._engine
        if eng is None or eng.pipe is None:
            raise RuntimeError("worker is closed")
        req = job.req
        key = self._job_key(req)                     # raises the reference's size error in the caller's thread
        item = self._prepare(req, key)
        b = eng.batcher
        if b is None:
            return eng.run_batch(key, [item])[0], item[1]
        return b.submit(key, item).result(), item[1]

    # ---- batching behind the reference's single-consumer pool ------------------------------------------
    def bind_queue(self, q) -> None:
        """Tell the worker which ``queue.Queue`` its caller's consumer thread takes jobs from (``WorkerPool.q``,
        backends/worker_pool.py:171).  Not needed behind ``get_worker_pool()``: that singleton is found by itself."""
        self._pool_q = q

    def _pool_queue(self):
        q = getattr(self, "_pool_q", None)
        if q is not None:
            return q
        if os.environ.get("LCM_DRAIN_QUEUE", "1").lower() in ("0", "false", "no", "off"):
            return None
        # the reference's process-wide pool (backends/worker_pool.py:421-469): looked up, never imported -- only when the
        # reference's module is already loaded and its pool is the one currently holding THIS worker
        import sys
        mod = sys.modules.get("backends.worker_pool")
        pool = getattr(mod, "_worker_pool", None) if mod is not None else None
        if pool is not None and getattr(pool, "_worker", None) is self:
            return getattr(pool, "q", None)
        return None

    def _drain(self, q, job, key, limit):
        """Take from the pool's queue, under its own mutex, the queued jobs that can share this call's sampler pass.

        ``WorkerPool._worker_loop`` is ONE thread that calls ``job.execute(worker)`` -> ``worker.run_job(job)`` and blocks
        (backends/worker_pool.py:294-341, :84-88): ``run_job`` never sees a second job, so the batch has to be collected
        here.  Rules: walk the queue from its head; a job qualifies when it is of the same class as the running one (the
        pool would have called ``run_job`` for it too), carries ``.req`` and an unresolved ``.fut``, and its request agrees on
        (size, steps, guidance, style); generation jobs with another key keep their place; the walk STOPS at the first
        job of any other kind (a ``ModeSwitchJob`` / ``CustomJob`` is a barrier: nothing queued behind it is overtaken).
        A drained job is finished here exactly as the loop would have (worker_pool.py:328-339): its future gets
        ``(png, seed)`` or the exception, and ``q.task_done()`` is called for it."""
        taken = []
        if limit <= 0:
            return taken
        cls = type(job)
        with q.mutex:
            dq = q.queue
            idx = []
            for i, j in enumerate(dq):
                if type(j) is not cls or not hasattr(j, "req") or getattr(j, "fut", None) is None:
                    break                             # barrier
                try:
                    same = (not j.fut.done()) and self._job_key(j.req) == key
                except Exception:
                    same = False                      # malformed request: it raises from its own run_job call, in its turn
                if same:
                    idx.append(i)
                    if len(idx) >= limit:
                        break
            for i in reversed(idx):
                taken.append(dq[i])
                del dq[i]
            if idx:
                q.not_full.notify(len(idx))           # space for blocked producers (Queue.put)
        taken.reverse()
        return taken

    def _finish_drained(self, q, job, fut, seed):
        """Runs on a finisher thread when the drained job's pass is done: what the pool's loop does for a job
        (backends/worker_pool.py:328-339)."""
        try:
            rgb, _ = fut.result()
            res = (encode_png(rgb), seed)
            if not job.fut.done():
                job.fut.set_result(res)
        except BaseException as e:                    # noqa
            try:
                if not job.fut.done():
                    job.fut.set_exception(e)
            except Exception:
                pass
        finally:
            q.task_done()

    def _gather_wait(self, q, want):
        """Closed-loop callers come back together: the clients of the previous call's batch get their results within a few
        milliseconds of each other and re-submit.  If that call drained k jobs, give about as many time to arrive before taking
        the batch -- otherwise the first arrival runs alone and the rest wait a whole pass for a small batch.  The window is a
        tenth of what the previous call took (a batched pass takes 35-125 ms), at most LCM_DRAIN_WINDOW_MS (default 12); a lone
        caller (k == 0) never waits."""
        import time as _t
        k = getattr(self, "_last_drained", 0)
        if k <= 0:
            return
        cap = float(os.environ.get("LCM_DRAIN_WINDOW_MS", "12") or 0) * 1e-3
        win = min(cap, max(0.002, 0.1 * getattr(self, "_last_call_s", 0.0)))
        if cap <= 0:
            return
        # ... but only while jobs keep coming: callers that come back together (closed loop) fill the queue within a millisecond,
        # a job every few hundred microseconds; under steady independent arrivals nothing comes for milliseconds and the wait is
        # pure added service time (open loop at 80 / 100 requests/s: p50 51 / 83 ms without it against 61 / 103 with).  So the
        # wait ends as soon as the queue has been still for 0.6 ms.  The target is one MORE job than last time: the previous
        # call's OWN caller gets its result last (the pool resolves that future after run_job returns,
        # backends/worker_pool.py:330-332) and re-submits a thread wake-up later -- taken without it, its job waits a whole pass
        # and then leads the next call, again one short: a stable state of 15 + 1 instead of 16.
        now = _t.perf_counter()
        deadline, last_n, last_t = now + win, q.qsize(), now
        while last_n < want(k + 1) and now < deadline:
            _t.sleep(0.0002)
            now = _t.perf_counter()
            n = q.qsize()
            if n != last_n:
                last_n, last_t = n, now
            elif now - last_t > 0.0006:
                break

    def run_job(self, job) -> Tuple[bytes, int]:
        eng = self._engine
        q = self._pool_queue() if eng is not None and getattr(eng, "batcher", None) is not None else None
        if q is None:
            (rgb, _), seed = self._submit(job)
            return encode_png(rgb), seed
        b = eng.batcher
        limit = b.max_batch * max(1, b.lanes) - 1
        import time as _t
        t_in = _t.perf_counter()
        self._gather_wait(q, lambda k: min(k, limit))
        t_call = _t.perf_counter()
        if q.empty():
            self._last_drained = 0
            (rgb, _), seed = self._submit(job)
            return encode_png(rgb), seed
        if eng.pipe is None:
            raise RuntimeError("worker is closed")
        key = self._job_key(job.req)
        # as many as the lanes can have in flight as full batches (two batch-8 passes on two lanes: 127 against 116 images/s)
        others = self._drain(q, job, key, limit)
        self._last_drained = len(others)
        if not others:
            (rgb, _), seed = self._submit(job)
            return encode_png(rgb), seed
        pending, fin = [], _finish_pool()
        try:
            items = list(fin.map(lambda j: self._prepare(j.req, key), [job] + others))
            t_prep = _t.perf_counter()
            futs = [b.submit(key, it, burst=True) for it in items]        # a complete set: the lanes split it at once
        except BaseException as e:                    # noqa  nothing was started: the drained jobs fail like the running one
            for j in others:
                if not j.fut.done():
                    j.fut.set_exception(e)
                q.task_done()
            raise
        for j, f, it in zip(others, futs[1:], items[1:]):
            done = threading.Event()
            pending.append(done)

            def _cb(f, j=j, seed=it[1], done=done):
                done.set()                            # the GPU side of this job is over
                fin.submit(self._finish_drained, q, j, f, seed)
            f.add_done_callback(_cb)
        try:
            rgb, _ = futs[0].result()
            t_gpu = _t.perf_counter()
            png = encode_png(rgb)
            if getattr(eng, "timing", None) is not None:      # LCM_WORKER_TIMING=1: ("call", jobs, gather s, drain + prepare s, pass s, PNG s)
                eng.timing.append(("call", len(items), t_call - t_in, t_prep - t_call, t_gpu - t_prep, _t.perf_counter() - t_gpu))
            return png, items[0][1]
        finally:
            # return to the pool's loop only when every pass this call started has left the GPU (a mode switch may be next in
            # the queue); the drained jobs' PNGs may still be deflating on the finisher threads -- no GPU state involved.
            # (Returning as soon as the call's own job is done, to keep the second lane fed, was measured at 8 / 16 / 24
            # closed-loop clients: 79 / 100 / 99 images/s against 86 / 98 / 102 -- the callers fall out of step and the batches
            # shrink; not kept.)
            for ev in pending:
                ev.wait()
            self._last_call_s = _t.perf_counter() - t_call

    def run_job_with_latents(self, job) -> Tuple[bytes, int, bytes]:
        # The reference re-runs the whole pipeline for the latents (cuda_worker.py:255-283); the sampler is
        # deterministic in the seed, so the same pass's final latents are identical and are pooled on device.
        (rgb, pool8), seed = self._submit(job)
        return encode_png(rgb), seed, pool8.tobytes(order="C")

    def run_jobs(self, jobs):
        """Batched convenience entry (not in the reference's protocol): jobs that agree on ``_job_key`` run as one pass
        (chunked to the plan batch sizes); PNGs are encoded on a small thread pool.  -> [(png, seed)] in job order."""
        from concurrent.futures import ThreadPoolExecutor
        seeds, groups = [], {}
        for i, job in enumerate(jobs):
            req = job.req
            seeds.append(int(req.seed) if getattr(req, "seed", None) is not None else int(torch.randint(0, 100_000_000, (1,)).item()))
            groups.setdefault(self._job_key(req), []).append(i)
        rgbs = [None] * len(jobs)
        sizes = self._engine.batch_sizes
        for key, idx in groups.items():
            while idx:
                n = max(s for s in sizes if s <= len(idx))
                part, idx = idx[:n], idx[n:]
                for i, (rgb, _) in zip(part, self._engine.run_batch(key, [(jobs[i].req, seeds[i]) for i in part])):
                    rgbs[i] = rgb
        with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
            pngs = list(ex.map(encode_png, rgbs))
        return list(zip(pngs, seeds))

    def close(self):
        """Optional (the reference's pool never calls it): detach from the engine now instead of at garbage collection."""
        eng, self._engine = getattr(self, "_engine", None), None
        if eng is not None:
            eng.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipLcmSDXLWorker(HipLcmWorker):
    """SDXL worker (drop-in for DiffusersSDXLCudaWorker, backends/cuda_worker.py:307-614): two text encoders
    (CLIP-L hidden_states[-2] | OpenCLIP-bigG hidden_states[-2] -> 2048; pooled bigG text_embeds), size/crop time ids,
    classifier-free guidance when guidance_scale > 1 (negative conditioning = zeros, force_zeros_for_empty_prompt)."""

    FAMILY = "sdxl"

    def _synthetic_weights(self):
        from ..config import SDXL_UNET, unet_config, vae_config
        ucfg = unet_config(SDXL_UNET)
        vcfg = vae_config(dict(scaling_factor=0.13025, sample_size=1024, force_upcast=True))
        return (_weights.synthetic_state_dict(_weights.unet_param_spec(ucfg), 0), ucfg,
                _weights.synthetic_state_dict(_weights.vae_param_spec(vcfg), 1), vcfg)

    @staticmethod
    def _load_text_encoders(eng, device, ckpt_root, clip_sd):
        from ..clip import CLIP_BIGG, CLIP_L, ClipTextHip, HashTokenizer, load_clip_dir, synthetic_clip
        from ..prompt import make_tokenizer
        eng.enc, eng.tok = [], []
        for idx, (sub, tsub, cfg0, seed) in enumerate((("text_encoder", "tokenizer", CLIP_L, 2),
                                                       ("text_encoder_2", "tokenizer_2", CLIP_BIGG, 3))):
            d = os.path.join(ckpt_root, sub) if ckpt_root else None
            real = True
            if clip_sd is not None and clip_sd[idx] is not None:          # carried inside a single-file checkpoint
                sd = clip_sd[idx]
                D = sd["embeddings.token_embedding.weight"].shape[1]
                cfg = dict(cfg0, hidden_size=D, vocab_size=sd["embeddings.token_embedding.weight"].shape[0],
                           num_hidden_layers=1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers.")),
                           intermediate_size=sd["encoder.layers.0.mlp.fc1.weight"].shape[0], num_attention_heads=D // 64)
                if "text_projection.weight" in sd:
                    cfg["projection_dim"] = sd["text_projection.weight"].shape[0]
            elif d and os.path.isdir(d):
                sd, cfg = load_clip_dir(d)
                cfg = dict(cfg0, **cfg)
            else:
                sd, cfg, real = synthetic_clip(cfg0, seed=seed), cfg0, False
            eng.enc.append(ClipTextHip(sd, cfg, device=device))
            eng.tok.append(make_tokenizer(ckpt_root, tsub, cfg["vocab_size"], real))
        if eng.enc[0].D + eng.enc[1].D != eng.pipe.unet.ctx_dim:
            raise RuntimeError("text encoder widths do not add up to the UNet cross_attention_dim")

    @staticmethod
    def _text_encoders(eng):
        return list(eng.enc)

    @staticmethod
    def _text_bytes(eng):
        return sum(e.weight_bytes() for e in eng.enc)

    @staticmethod
    def _conditioning(eng, reqs, width, height, guidance, lane=0):
        prompts = [r.prompt for r in reqs]
        _, enc = eng._lane_encoders(lane)
        h1 = enc[0].forward(eng.tok[0](prompts), output="penultimate")
        h2, pooled = enc[1].forward(eng.tok[1](prompts), output="penultimate", pooled=True)
        pe = torch.cat([h1, h2], dim=-1)
        tids = torch.tensor([[float(height), float(width), 0.0, 0.0, float(height), float(width)]] * len(reqs))
        kw = dict(added=(pooled, tids))
        if guidance > 1.0 and not eng.pipe.unet.has_cond:
            kw["negative_embeds"] = torch.zeros_like(pe)
            kw["negative_added"] = (torch.zeros_like(pooled), tids)
        return pe, kw
