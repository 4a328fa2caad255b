"""The reference's hot path end to end on the CPU (torch fp32).  Oracle only.

Restates ``StableDiffusionPipeline.__call__`` + ``LCMScheduler`` as invoked at
backends/cuda_worker.py:221-229, in the order of operations of the numpy twin
backends/rknnlcm.py:450-647: prompt embeds -> guidance embedding -> timesteps ->
latents (generator) -> [UNet -> scheduler.step]*n -> /scaling_factor -> VAE ->
post-process.  Prompt embeddings are an INPUT here (the CLIP text encoder is
row (f1) of SURVEY.md section 8).
"""
from __future__ import annotations

import numpy as np
import torch

from . import glue
from .scheduler import LCMSchedulerOracle
from .unet import UNetOracle
from .vae import VAEDecoderOracle


class LCMPipelineOracle:
    def __init__(self, unet_sd, vae_sd, unet_cfg=None, vae_cfg=None):
        self.unet = UNetOracle(unet_sd, unet_cfg)
        self.vae = VAEDecoderOracle(vae_sd, vae_cfg)
        self.sched = LCMSchedulerOracle()

    @torch.inference_mode()
    def __call__(self, prompt_embeds, width, height, steps, guidance_scale, seed,
                 negative_embeds=None, return_all=False, added=None, negative_added=None):
        """prompt_embeds [1,77,768] float; returns dict(image_u8 NHWC, image float NCHW, latents)."""
        pe = torch.as_tensor(np.asarray(prompt_embeds), dtype=torch.float32)
        B = pe.shape[0]
        assert B == 1, "one request per call, like run_job (backends/cuda_worker.py:201)"
        ts = self.sched.set_timesteps(int(steps))
        lat, noises = glue.prepare_latents(seed, height, width, len(ts) - 1, self.sched.init_noise_sigma)
        tcd = self.unet.cfg.get("time_cond_proj_dim")
        cond = None
        if tcd:
            cond = torch.from_numpy(glue.guidance_scale_embedding(
                np.full((B,), guidance_scale - 1.0, dtype=np.float32), tcd, np.float32))
        do_cfg = guidance_scale > 1.0 and not tcd
        if do_cfg:
            ne = torch.as_tensor(np.asarray(negative_embeds), dtype=torch.float32)
        trace = []
        for i, t in enumerate(ts):
            if do_cfg:
                add2 = None
                if added is not None:
                    na = negative_added if negative_added is not None else (torch.zeros_like(torch.as_tensor(added[0]).float()), added[1])
                    add2 = (torch.cat([torch.as_tensor(na[0]).float(), torch.as_tensor(added[0]).float()]),
                            torch.cat([torch.as_tensor(na[1]).float(), torch.as_tensor(added[1]).float()]))
                e2 = self.unet.forward(torch.cat([lat, lat]), int(t), torch.cat([ne, pe]), None, added=add2)
                eu, et = e2.chunk(2)
                eps = eu + guidance_scale * (et - eu)
            else:
                a1 = None if added is None else (torch.as_tensor(added[0]).float(), torch.as_tensor(added[1]).float())
                eps = self.unet.forward(lat, int(t), pe, cond, added=a1)
            lat, den = self.sched.step(eps, i, lat, noises[i] if i < len(noises) else None)
            if return_all:
                trace.append((eps.clone(), lat.clone()))
        img = self.vae.decode(lat)
        out = dict(image=img.numpy(), image_u8=glue.postprocess_u8(img.numpy()), latents=lat.numpy(),
                   timesteps=ts)
        if return_all:
            out["trace"] = trace
        return out
