"""The LCM sampler on the MI355X: the arithmetic under ``self.pipe(...)`` (backends/cuda_worker.py:221-229)
as one hipGraph of hand-written HIP kernels.

Per request batch: prompt embeddings [B,77,768] + per-request seeds -> uint8 RGB [B,H,W,3] (+ final latents).
Order of operations follows backends/rknnlcm.py:450-647 (prompt embeds -> guidance embedding -> timesteps ->
latents from the request's generator -> [UNet -> LCMScheduler.step] x n -> /scaling_factor -> VAE -> u8).

RNG contract: each request owns a CPU ``torch.Generator`` seeded with its seed; draws are
latents[1,4,h,w] then one noise tensor per non-final step.  All draws happen on the host BEFORE the graph is
launched, so the captured graph is RNG-free and replays for any seed.
"""
from __future__ import annotations

import os
import threading

import numpy as np
import torch

from . import ops
from .config import TEXT_SEQ_LEN, VAE_SCALE_FACTOR
from .lib import LcmHipError
from .model import UNetHip, VAEDecoderHip
from .scheduler import LCMSchedule


def guidance_scale_embedding(w: np.ndarray, dim: int) -> np.ndarray:
    """w = guidance_scale - 1 per request (backends/rknnlcm.py:572, :651-677)."""
    w = np.asarray(w, dtype=np.float32) * 1000
    half = dim // 2
    f = np.exp(np.arange(half, dtype=np.float32) * -(np.log(10000.0) / (half - 1)))
    e = w[:, None] * f[None, :]
    e = np.concatenate([np.sin(e), np.cos(e)], axis=1)
    if dim % 2 == 1:
        e = np.pad(e, [(0, 0), (0, 1)])
    return e.astype(np.float32)


def sinusoid_host(values: np.ndarray, dim: int) -> np.ndarray:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) of a flat array: [cos | sin] per value (SDXL add_time_proj)."""
    half = dim // 2
    f = np.exp(-np.log(10000.0) * np.arange(half, dtype=np.float32) / half)
    e = np.asarray(values, dtype=np.float32).reshape(-1, 1) * f[None, :]
    return np.concatenate([np.cos(e), np.sin(e)], axis=1).astype(np.float32)


def check_size(width: int, height: int) -> None:
    """What diffusers' check_inputs raises for the reference ("`height` and `width` have to be divisible by 8 ...";
    backends/rknnlcm.py:380-381 has the same rule and text), as an LcmHipError; pinned by tests/golden/worker_contract.json."""
    if width % 8 or height % 8 or width <= 0 or height <= 0:
        raise LcmHipError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")


def draw_noise(seed: int, h: int, w: int, n_extra: int, sigma: float = 1.0):
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    shape = (1, 4, h, w)
    lat = torch.randn(shape, generator=g, dtype=torch.float32) * sigma
    extra = [torch.randn(shape, generator=g, dtype=torch.float32) for _ in range(n_extra)]
    return lat, extra


_DEFAULT_WS = {}


def _default_workspace(device):
    """The device-wide split-K workspace of launches outside any pipeline lane (eager launches of tests / tools on foreign
    streams): one process-lifetime tensor per device, never a lane's."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _DEFAULT_WS.get(key)
    if t is None:
        t = _DEFAULT_WS[key] = torch.empty(int(os.environ.get("LCM_SPLITK_DEFAULT_WS_MB", "256")) << 18, dtype=torch.float32, device=device)
    return t


def _new_workspace(device, mb=None):
    # sized for a batch of 8 at 768x768 (472 MB) / SDXL 1024x1024 (fp32 [splits][M][N] of the largest split layer); a launch
    # that needs more fails loudly (a silently smaller split factor would change the numbers)
    return torch.empty(int(mb if mb is not None else os.environ.get("LCM_SPLITK_WS_MB", "1024")) << 18, dtype=torch.float32, device=device)


class _Lane:
    """One sampler instance of a pipeline: a stream, executors with scratch of their own (weights shared), its plans /
    captured graphs and its split-K workspace.  Lanes of one pipeline run concurrently: a single batch-1 pass is a
    chain of ~1700 latency-bound launches that leaves a third of the MI355X idle, so two requests in flight on two
    lanes finish in less than twice the time of one."""

    def __init__(self, pipe, index, unet, vae):
        self.index = index
        self.unet, self.vae = unet, vae
        self.stream = ops.acquire_stream(pipe.device)       # a handle of this lane's own (not torch's recycled pool)
        self.plans = {}
        # The split-K workspace belongs to THIS lane of THIS pipeline (the library looks it up by launch stream): two
        # pipelines on one GPU -- an SD1.5 and an SDXL engine, or unshared engines of several pool workers -- run on threads
        # with no common lock and must never share fp32 slabs.  It lives as long as the lane (i.e. as the captured graphs that
        # bake its pointer in) and is unregistered by LcmHipPipeline.close().
        self.splitk_ws = _new_workspace(pipe.device)
        ops.set_stream_workspace(self.stream, self.splitk_ws)
        # side stream: launches that fork off the main chain inside one pass (the resnets' conv_shortcut GEMMs), with a
        # split-K workspace of its own -- they run concurrently with the main stream's split layers
        self.side = ops.acquire_stream(pipe.device)
        self.side_ws = _new_workspace(pipe.device, mb=256)
        ops.set_stream_workspace(self.side, self.side_ws)
        self.unet.side_stream = self.side
        self.vae.side_stream = self.side


class _Plan:
    """Buffers + captured graph for one (B, h, w, steps, cfg) key."""

    def __init__(self, pipe, B, h, w, steps, do_cfg, lane=None):
        # Every zero-fill below must be ordered before the first use on the lane's (non-blocking) stream: allocate
        # under that stream, or a fill still queued on the null stream can land AFTER the request's uploads.
        self.lane = lane if lane is not None else pipe.lanes[0]
        with torch.cuda.stream(self.lane.stream):
            self._init(pipe, B, h, w, steps, do_cfg)

    def _init(self, pipe, B, h, w, steps, do_cfg):
        dev = pipe.device
        self.B, self.h, self.w, self.steps, self.do_cfg = B, h, w, steps, do_cfg
        UB = 2 * B if do_cfg else B
        self.UB = UB
        self.ehs = torch.zeros(UB * TEXT_SEQ_LEN, pipe.unet.ctx_dim, dtype=torch.float16, device=dev)
        self.wemb = torch.zeros(UB, pipe.unet.cfg.get("time_cond_proj_dim") or 8, dtype=torch.float16, device=dev)
        self.add_in = (torch.zeros(UB, pipe.unet.added_dim, dtype=torch.float16, device=dev)
                       if pipe.unet.has_added else None)          # SDXL: [pooled text embeds | sinusoid(time ids)]
        self.lat0 = torch.zeros(B, 4, h, w, dtype=torch.float32, device=dev)         # request input
        self.lat = torch.zeros(UB, 4, h, w, dtype=torch.float32, device=dev)          # sampler state
        self.noise = torch.zeros(max(steps - 1, 1), B, 4, h, w, dtype=torch.float32, device=dev)
        self.eps = torch.zeros(UB, h, w, 4, dtype=torch.float32, device=dev)
        self.rgb = torch.zeros(B, h * VAE_SCALE_FACTOR, w * VAE_SCALE_FACTOR, 3, dtype=torch.uint8, device=dev)
        self.pool8 = torch.zeros(B, 4, 8, 8, dtype=torch.float16, device=dev)
        self.img_f32 = None
        self.guidance = 1.0
        self.graph = None
        # pinned staging for H2D / D2H
        self.h_lat = torch.zeros(B, 4, h, w, dtype=torch.float32).pin_memory()
        self.h_noise = torch.zeros(max(steps - 1, 1), B, 4, h, w, dtype=torch.float32).pin_memory()
        self.h_rgb = torch.zeros(B, h * VAE_SCALE_FACTOR, w * VAE_SCALE_FACTOR, 3, dtype=torch.uint8).pin_memory()
        self.h_pool8 = torch.zeros(B, 4, 8, 8, dtype=torch.float16).pin_memory()
        self.h_latout = torch.zeros(B, 4, h, w, dtype=torch.float32).pin_memory()


class LcmHipPipeline:
    def __init__(self, unet_sd, vae_sd, unet_cfg=None, vae_cfg=None, device="cuda:0", schedule: LCMSchedule | None = None,
                 use_graph=True):
        if not torch.cuda.is_available():
            raise LcmHipError("LcmHipPipeline needs an MI355X (torch.cuda.is_available() is False); "
                              "there is no CPU fallback on the product path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.unet = UNetHip(unet_sd, unet_cfg, self.device)
        self.vae = VAEDecoderHip(vae_sd, vae_cfg, self.device)
        self.sched = schedule or LCMSchedule()
        self.use_graph = use_graph
        self._tuned_keys = set()
        self._build_lock = threading.RLock()      # tuning / eager warm-up / capture of a plan: one lane at a time
        self.lanes = [_Lane(self, 0, self.unet, self.vae)]
        self.stream = self.lanes[0].stream
        self._plans = self.lanes[0].plans
        # fp32 scratch for deterministic split-K of the deep-K / small-M layers (low-res UNet levels at batch 1): per lane
        # (above); launches on streams that are no lane's (tests, tools) use a device-wide default that no lane shares.
        self._splitk_ws = self.lanes[0].splitk_ws
        ops.set_workspace(_default_workspace(self.device))
        if "LCM_CONV_IMPL" in os.environ:            # A/B switches for kernel work
            ops.set_conv_impl(int(os.environ["LCM_CONV_IMPL"]))
        if "LCM_GN_FUSED_BYTES" in os.environ:
            ops.set_gn_fused_bytes(int(os.environ["LCM_GN_FUSED_BYTES"]))
        if "LCM_PERSIST_N" in os.environ:
            ops.set_persist_n(int(os.environ["LCM_PERSIST_N"]))
        if "LCM_HALO_PIPE" in os.environ:
            ops.set_halo_pipe_threshold(int(os.environ["LCM_HALO_PIPE"]))
        if "LCM_ATTN_KSPLIT" in os.environ:          # 0: unsplit streaming attention at every length (changes the bits of the >= 1024-key levels)
            ops.set_attention_ksplit(int(os.environ["LCM_ATTN_KSPLIT"]))
        if "LCM_KERNEL_VARIANT" in os.environ:
            ops.set_kernel_variant(int(os.environ["LCM_KERNEL_VARIANT"]))

    # ------------------------------------------------------------------------------------------
    def lane(self, index: int) -> _Lane:
        """Lane ``index`` (created on first use: executor views + stream + workspace; weights are shared)."""
        with self._build_lock:
            while len(self.lanes) <= index:
                self.lanes.append(_Lane(self, len(self.lanes), self.unet.view(), self.vae.view()))
        return self.lanes[index]

    def _enqueue(self, P: _Plan, guidance: float, want_float=False, taps=None):
        """Enqueue the whole sampler on the current stream (this is what gets captured)."""
        B, UB, h, w = P.B, P.UB, P.h, P.w
        unet, vae = P.lane.unet, P.lane.vae
        ts = self.sched.timesteps(P.steps)
        if P.do_cfg:
            P.lat[:B].copy_(P.lat0)
            P.lat[B:].copy_(P.lat0)
        else:
            P.lat.copy_(P.lat0)
        kv = unet.encode_context(P.ehs, UB)
        aug = unet.encode_added(P.add_in, UB) if unet.has_added else None
        wemb = P.wemb if unet.has_cond else None
        # the time-embedding MLP + all time_emb_proj of EVERY step ahead of the loop (they depend on the schedule and the
        # request's guidance only): 5 launches per pass instead of 4 per step
        ta_all = unet.time_embed_all([int(t) for t in ts], wemb, UB, aug) if len(ts) <= unet.MAX_HOISTED_STEPS else None
        for i, t in enumerate(ts):
            unet.forward(P.lat, int(t), kv, wemb, UB, h, w, P.eps, taps=taps if i == 0 else None, aug=aug,
                         ta=ta_all[i * UB:(i + 1) * UB] if ta_all is not None else None)
            coef, last = self.sched.step_coefficients(ts, i)
            noise = P.noise[min(i, P.noise.shape[0] - 1)]
            if P.do_cfg:   # rows [0,B) = negative prompt, [B,2B) = prompt
                ops.scheduler_step(P.eps[B:], P.lat[B:], noise, coef, last, B, h, w, eps_uncond=P.eps[:B], guidance=guidance)
                P.lat[:B].copy_(P.lat[B:])
            else:
                ops.scheduler_step(P.eps, P.lat, noise, coef, last, B, h, w)
        final = P.lat[B:] if P.do_cfg else P.lat
        ops.latents_pool8(final, P.pool8, B, h, w)
        if want_float and P.img_f32 is None:
            P.img_f32 = torch.zeros(B, h * 8, w * 8, 3, dtype=torch.float32, device=self.device)
        vae.decode(final, B, h, w, P.rgb, img_f32=P.img_f32 if want_float else None, taps=taps)
        return final

    def plan(self, B, h, w, steps, do_cfg=False, guidance=None, lane=0) -> _Plan:
        # classifier-free guidance bakes the guidance value into the captured step kernels: one plan per value
        key = (B, h, w, steps, do_cfg, round(float(guidance), 4) if do_cfg and guidance is not None else None)
        L = self.lane(lane)
        P = L.plans.get(key)
        if P is None:
            with self._build_lock:
                P = L.plans.get(key)
                if P is None:
                    P = _Plan(self, B, h, w, steps, do_cfg, L)
                    L.plans[key] = P
        return P

    def tune(self, P: _Plan, verbose=False):
        """Autotune the launch plan of every contraction shape this plan touches (once per shape per process)."""
        if os.environ.get("LCM_AUTOTUNE", "1") == "0" or getattr(P, "tuned", False):
            return
        from . import autotune
        stream = P.lane.stream
        with self._build_lock, torch.cuda.stream(stream):
            self._enqueue(P, 1.0)                    # allocate scratch, warm caches
            with ops.recording() as recs:            # thread-local: another lane's eager launches never land in here
                self._enqueue(P, 1.0)
            stream.synchronize()
            todo = [r for r in recs if r[0] is not None and r[0] not in self._tuned_keys]
            # in situ every layer's weights come from HBM and its input was written by the previous kernel, never by a
            # previous run of the same layer: time the candidates with cold caches (autotune._time_cold); measured
            # +2.5 % at batch 1 and +1.7 % at batch 8 over warm back-to-back timing.  LCM_AUTOTUNE_COLD=0: warm timing
            cold = os.environ.get("LCM_AUTOTUNE_COLD", "1") != "0"
            # LCM_TUNE_SPLITS=1: offline table generation only (tools/make_plans.py) -- a serving process never times splits
            res = autotune.autotune(todo, P.lane.splitk_ws.numel() * 4, verbose=verbose, cold=cold,
                                    tune_splits=os.environ.get("LCM_TUNE_SPLITS", "0") == "1")
            self._tuned_keys.update(res.keys())
            stream.synchronize()
        P.tuned = True

    def drop_plans(self):
        for L in self.lanes:
            for P in L.plans.values():
                if P.graph is not None:
                    P.graph.close()
            L.plans.clear()

    def close(self):
        """Drop the captured graphs and take this pipeline's workspaces out of the library's per-stream table (the library
        keeps raw pointers).  Idempotent; also run when the pipeline is collected."""
        lanes, self.lanes = getattr(self, "lanes", []), []
        for L in lanes:
            for P in L.plans.values():
                if P.graph is not None:
                    P.graph.close()
            L.plans.clear()
            try:
                torch.cuda.synchronize(self.device)
                ops.set_stream_workspace(L.stream, L.splitk_ws, forget=True)      # only if the entry is still this lane's
                ops.set_stream_workspace(L.side, L.side_ws, forget=True)
                ops.release_stream(L.stream)
                ops.release_stream(L.side)
                L.stream = L.side = None
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def generate(self, prompt_embeds, seeds, width, height, steps, guidance_scale=1.0, negative_embeds=None,
                 want_float=False, taps=None, latents=None, added=None, negative_added=None, noises=None, lane=0):
        """prompt_embeds: [B,77,ctx] (any float dtype, host or device); seeds: B ints.  noises: optional per-request
        ``draw_noise(seed, h, w, steps - 1, init_noise_sigma)`` results drawn ahead by the callers (the worker's pool
        threads draw them in parallel, off the dispatcher's serial path); None: drawn here from the seeds.
        lane: which of the pipeline's concurrent sampler instances runs the request (calls on different lanes may overlap;
        calls on one lane must be serialised by the caller).
        Returns dict(rgb uint8 [B,H,W,3] (host), latents fp32 [B,4,h,w] (host), pool8 fp16 [B,4,8,8] (host))."""
        torch.cuda.set_device(self.device)        # the pool may call from a thread other than the constructing one
        pe = torch.as_tensor(prompt_embeds)
        B = pe.shape[0]
        check_size(width, height)
        h, w = height // VAE_SCALE_FACTOR, width // VAE_SCALE_FACTOR
        steps = int(steps)
        has_cond = self.unet.has_cond
        do_cfg = (guidance_scale > 1.0) and not has_cond
        if do_cfg and negative_embeds is None:
            raise LcmHipError("classifier-free guidance needs negative_embeds")
        if self.unet.has_added and added is None:
            raise LcmHipError("this UNet needs added=(pooled_text_embeds [B,P], time_ids [B,6]) (SDXL text_time embedding)")
        P = self.plan(B, h, w, steps, do_cfg, guidance_scale, lane=lane)
        stream = P.lane.stream
        with torch.cuda.stream(stream):
            # ---- host-side request state -> device (outside the graph) ----
            for b, s in enumerate(seeds):
                if latents is not None:
                    P.h_lat[b].copy_(torch.as_tensor(latents[b]).reshape(4, h, w))
                    extra = []
                else:
                    l0, extra = noises[b] if noises is not None else draw_noise(s, h, w, steps - 1, self.sched.init_noise_sigma)
                    P.h_lat[b].copy_(l0[0])
                for i, n in enumerate(extra):
                    P.h_noise[i, b].copy_(n[0])
            P.lat0.copy_(P.h_lat, non_blocking=True)
            P.noise.copy_(P.h_noise, non_blocking=True)
            pe16 = pe.to(torch.float16).reshape(B * TEXT_SEQ_LEN, -1)
            if do_cfg:
                ne16 = torch.as_tensor(negative_embeds).to(torch.float16).reshape(B * TEXT_SEQ_LEN, -1)
                P.ehs[:B * TEXT_SEQ_LEN].copy_(ne16, non_blocking=True)
                P.ehs[B * TEXT_SEQ_LEN:].copy_(pe16, non_blocking=True)
            else:
                P.ehs.copy_(pe16, non_blocking=True)
            if self.unet.has_added:
                def _add_rows(a):
                    pooled, tids = a
                    pooled = torch.as_tensor(pooled).to(torch.float32).reshape(B, -1).cpu()
                    sin = sinusoid_host(np.asarray(tids, dtype=np.float32).reshape(-1), self.unet.cfg["addition_time_embed_dim"])
                    return torch.cat([pooled, torch.from_numpy(sin).reshape(B, -1)], dim=1).to(torch.float16)
                rows = _add_rows(added)
                if do_cfg:
                    neg_rows = _add_rows(negative_added if negative_added is not None else (torch.zeros(B, rows.shape[1] - 6 * self.unet.cfg["addition_time_embed_dim"]), added[1]))
                    P.add_in[:B].copy_(neg_rows, non_blocking=True)
                    P.add_in[B:].copy_(rows, non_blocking=True)
                else:
                    P.add_in.copy_(rows, non_blocking=True)
            if has_cond:
                gs = np.full((B,), float(guidance_scale) - 1.0, dtype=np.float32)
                P.wemb.copy_(torch.from_numpy(guidance_scale_embedding(gs, P.wemb.shape[1])).to(torch.float16),
                             non_blocking=True)
            # ---- the sampler: eager once (allocates scratch), then captured + replayed ----
            eager = (not self.use_graph) or taps is not None or want_float
            if eager:
                with self._build_lock:               # eager launches allocate scratch
                    final = self._enqueue(P, guidance_scale, want_float=want_float, taps=taps)
            else:
                if P.graph is None:
                    with self._build_lock:
                        if P.graph is None:
                            self.tune(P)
                            self._enqueue(P, guidance_scale)           # warm-up: allocates every scratch buffer
                            stream.synchronize()
                            g = ops.Graph()
                            with g:
                                self._enqueue(P, guidance_scale)
                            P.graph = g
                P.graph.launch()
                final = P.lat[B:] if do_cfg else P.lat
            P.h_rgb.copy_(P.rgb, non_blocking=True)
            P.h_pool8.copy_(P.pool8, non_blocking=True)
            P.h_latout.copy_(final, non_blocking=True)
            stream.synchronize()
        out = dict(rgb=P.h_rgb.numpy().copy(), latents=P.h_latout.numpy().copy(), pool8=P.h_pool8.numpy().copy())
        if want_float:
            out["image"] = P.img_f32.cpu().numpy()   # NHWC float, pre-clamp
        return out

    # hot loop only (device resident inputs already in the plan): used by bench.py
    def replay(self, P: _Plan):
        if P.graph is None:
            raise LcmHipError("plan has no captured graph")
        P.graph.launch()
