#!/usr/bin/env python3
"""Regenerate the launch-plan table shipped with the package (tuned_plans_gfx950.json): tunes every contraction
shape of the standard request shapes with many cold-cache repetitions on the GPU box and writes the winners.
Batch-1 plans come first and are the only ones whose split-K factor is tuned (it becomes the canonical K partition of
that per-image shape); the batched plans after them tune tile / variant around the partition of their per-image shape.
Usage (GPU box): python tools/make_plans.py gpurun_out/tuned_plans_gfx950.json [sd15|sdxl|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "sd15"
os.environ["LCM_TUNE_CACHE"] = out
os.environ["LCM_TUNED_PLANS"] = "0"
os.environ.setdefault("LCM_AUTOTUNE_REPS", "10")
os.environ["LCM_TUNE_SPLITS"] = "1"
import torch
import sdlcm_amd  # noqa
from sdlcm_amd import weights
from sdlcm_amd.pipeline import LcmHipPipeline

t0 = time.time()
if which in ("sd15", "all"):
    pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device="cuda:0")
    for (B, size, steps) in [(1, 512, 4), (1, 768, 4), (1, 256, 4), (1, 128, 4), (1, 192, 4), (1, 64, 4),
                             (2, 512, 4), (4, 512, 4), (8, 512, 4), (2, 256, 4), (3, 128, 4), (8, 768, 4)]:
        P = pipe.plan(B, size // 8, size // 8, steps, False, 1.0)
        pipe.tune(P)
        print(f"sd15 B{B} {size}px tuned, {time.time() - t0:.0f}s", flush=True)
    pipe.drop_plans()
    del pipe
    torch.cuda.empty_cache()
if which in ("sdxl", "all"):
    from sdlcm_amd.config import SDXL_UNET, unet_config, vae_config
    ucfg, vcfg = unet_config(SDXL_UNET), vae_config(dict(scaling_factor=0.13025, sample_size=1024, force_upcast=True))
    pipe = LcmHipPipeline(weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0), weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1),
                          ucfg, vcfg, device="cuda:0")
    for (B, size, steps, cfg) in [(1, 1024, 4, False), (1, 1024, 4, True)]:
        P = pipe.plan(B, size // 8, size // 8, steps, cfg, 5.0 if cfg else 1.0)
        pipe.tune(P)
        print(f"sdxl B{B} {size}px cfg={cfg} tuned, {time.time() - t0:.0f}s", flush=True)
print("done", flush=True)
