#!/usr/bin/env python3
"""ONE eager (no hipGraph) batch-B 512x512 4-step pass, as a target for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE per
dispatch).  Graph capture + counters crashed on this pool earlier; eager launches are plain dispatches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdlcm_amd  # noqa
from sdlcm_amd import weights
from sdlcm_amd.pipeline import LcmHipPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print("building pipeline", flush=True)
pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device="cuda:0")
pipe.use_graph = False
pe = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(5)).half()
P = pipe.plan(B, 64, 64, 4)
pipe.tune(P)                      # shipped plans: no tuning launches for the standard shapes
print("plans set; running eager passes", flush=True)
for i in range(2):
    pipe.generate(pe, list(range(B)), 512, 512, 4, 1.0, want_float=True)
    print(f"pass {i} done", flush=True)
