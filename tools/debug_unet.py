#!/usr/bin/env python3
"""Layer-by-layer error report of the HIP UNet/VAE against the oracle (GPU box only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import weights
from sdlcm_amd.pipeline import LcmHipPipeline, guidance_scale_embedding
from oracle.pipeline import LCMPipelineOracle

usd, vsd = weights.synthetic_unet(), weights.synthetic_vae()
hip = LcmHipPipeline(usd, vsd)
ora = LCMPipelineOracle(usd, vsd)
B, h, w, t = 1, 16, 16, 759
g = torch.Generator().manual_seed(11)
lat = torch.randn(B, 4, h, w, generator=g)
pe = torch.randn(B, 77, 768, generator=g).half()
wemb = torch.from_numpy(guidance_scale_embedding(np.zeros(B, np.float32), 256))
ora.unet.taps = {}
ref = ora.unet.forward(lat, t, pe.float(), wemb).numpy()
taps = {}
with torch.cuda.stream(hip.stream):
    kv = hip.unet.encode_context(pe.reshape(B * 77, 768).to(hip.device), B)
    eps = torch.zeros(B, h, w, 4, dtype=torch.float32, device=hip.device)
    hip.unet.forward(lat.to(hip.device), t, kv, wemb.to(hip.device, torch.float16), B, h, w, eps, taps=taps)
    hip.stream.synchronize()
for k, v in taps.items():
    r = ora.unet.taps[k].numpy()
    e = np.abs(v.numpy() - r)
    print(f"{k:40s} max|d|={e.max():.4g} rel={e.max() / (np.abs(r).max() + 1e-9):.4g} refmax={np.abs(r).max():.4g}")
e = np.abs(eps.cpu().numpy().transpose(0, 3, 1, 2) - ref)
print("eps max|d|", e.max(), "ref std", ref.std())
