#!/usr/bin/env python3
"""Add launch plans for more request shapes to the table shipped with the package WITHOUT touching the entries it has
(their K partitions define the output bits of the sizes already covered): the reference UI's other stock sizes
(lcm-sr-ui/src/utils/constants.js:6-15 -- 640x360, 512x768, 768x512, 1024x1024; 960x540 is not a multiple of 8 and is
refused as diffusers refuses it) and small batches of 768x768.  Shapes not in the table get tile / variant tuned, and
-- for single-image shapes -- their K partition, as tools/make_plans.py does.
Usage (GPU box): python tools/extend_plans.py gpurun_out/tuned_plans_gfx950.json   (then copy it over the packaged file)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
os.environ["LCM_TUNE_CACHE"] = out
os.environ["LCM_TUNED_PLANS"] = "1"
os.environ.setdefault("LCM_AUTOTUNE_REPS", "10")
os.environ["LCM_TUNE_SPLITS"] = "1"
import torch
import sdlcm_amd  # noqa
from sdlcm_amd import lib, weights
from sdlcm_amd.pipeline import LcmHipPipeline

n0 = len(lib.known_plans())
t0 = time.time()
pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device="cuda:0")
for (B, W, H, steps) in [(1, 640, 360, 4), (1, 512, 768, 4), (1, 768, 512, 4), (1, 1024, 1024, 4),
                         (2, 640, 360, 4), (2, 512, 768, 4), (2, 768, 512, 4), (2, 768, 768, 4), (4, 768, 768, 4), (4, 640, 360, 4)]:
    P = pipe.plan(B, H // 8, W // 8, steps, False, 1.0)
    pipe.tune(P)
    print(f"sd15 B{B} {W}x{H} tuned, {time.time() - t0:.0f}s, table {len(lib.known_plans())} entries", flush=True)
print(f"done: {n0} -> {len(lib.known_plans())} entries", flush=True)
