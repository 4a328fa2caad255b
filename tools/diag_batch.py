#!/usr/bin/env python3
"""Where does a request inside a batch first differ from its solo run?  (diagnostic for the batch-invariance contract)
   python tools/diag_batch.py [size] [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdlcm_amd  # noqa
from sdlcm_amd import weights
from sdlcm_amd.pipeline import LcmHipPipeline
S = int(sys.argv[1]) if len(sys.argv) > 1 else 768
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device="cuda:0")
pe = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(13)).half()
seeds = [900 + i for i in range(B)]
tb, ts = {}, {}
ob = pipe.generate(pe, seeds, S, S, steps, 1.0, want_float=True, taps=tb)
i = 0
os_ = pipe.generate(pe[i:i + 1], [seeds[i]], S, S, steps, 1.0, want_float=True, taps=ts)
print("latents equal:", np.array_equal(ob["latents"][i], os_["latents"][0]), " rgb equal:", np.array_equal(ob["rgb"][i], os_["rgb"][0]),
      " image max diff:", float(np.abs(ob["image"][i] - os_["image"][0]).max()))
for k in tb:
    a, b = tb[k][i].numpy(), ts[k][0].numpy()
    eq = np.array_equal(a, b)
    print(f"{'==' if eq else '!='} {k:50s} maxdiff {np.abs(a - b).max():.3g}")
