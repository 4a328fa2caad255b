#!/usr/bin/env python3
"""Does a generate() at one size disturb a later generate() at another size on the same pipeline?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import weights
from sdlcm_amd.pipeline import LcmHipPipeline
from oracle.pipeline import LCMPipelineOracle

usd, vsd = weights.synthetic_unet(), weights.synthetic_vae()
ora = LCMPipelineOracle(usd, vsd)
pe = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(3)).half()
ref = ora(pe.float(), 768, 512, 2, 1.0, 11)


def check(hip, tag):
    out = hip.generate(pe, [11], 768, 512, 2, 1.0, want_float=True)
    a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
    b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
    print(f"{tag}: latents max|d| {np.abs(out['latents'] - ref['latents']).max():.4g}  image max|d| {np.abs(a - b).max():.4g}", flush=True)


for seq in sys.argv[1:]:
    hip = LcmHipPipeline(usd, vsd)
    for s in seq.split(","):
        if s == "768":
            check(hip, f"[{seq}]")
        elif s == "vae2":
            lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(3)) * 0.9
            with torch.cuda.stream(hip.stream):
                rgb = torch.zeros(2, 128, 128, 3, dtype=torch.uint8, device=hip.device)
                img = torch.zeros(2, 128, 128, 3, dtype=torch.float32, device=hip.device)
                hip.vae.decode(lat.to(hip.device), 2, 16, 16, rgb, img_f32=img)
                hip.stream.synchronize()
        elif s == "vae768":      # decode the ORACLE's final latents at 768x512 with taps, report the first bad layer
            ora.vae.taps = {}
            lat = torch.from_numpy(ref["latents"])
            refi = ora.vae.decode(lat).numpy()
            vt = {}
            with torch.cuda.stream(hip.stream):
                rgb = torch.zeros(1, 512, 768, 3, dtype=torch.uint8, device=hip.device)
                img = torch.zeros(1, 512, 768, 3, dtype=torch.float32, device=hip.device)
                hip.vae.decode(lat.to(hip.device), 1, 64, 96, rgb, img_f32=img, taps=vt)
                hip.stream.synchronize()
            for k, v in vt.items():
                r = ora.vae.taps[k].numpy()
                e = np.abs(v.numpy() - r)
                print(f"   VAE {k:42s} max|d|={np.nanmax(e):.4g} refmax={np.abs(r).max():.3g}", flush=True)
            print("   vae768 image max|d|", np.abs(img.cpu().numpy().transpose(0, 3, 1, 2) - refi).max(), flush=True)
        else:
            n = int(s)
            hip.generate(pe, [5], n, n, 2, 1.0, want_float=True)
    del hip
    torch.cuda.empty_cache()
