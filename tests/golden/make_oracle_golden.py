#!/usr/bin/env python3
"""Oracle outputs for the slow BASELINE-size parity cases, computed ONCE in the build container (CPU, fp32) and committed as
small fixtures, so the GPU suite does not spend minutes of CPU oracle time on every run (round 2: 718 s of a 900 s limit).

  python tests/golden/make_oracle_golden.py [case ...]      -> tests/golden/oracle_<case>.npz

What is stored per case: the inputs' seeds (the tests rebuild the identical fp16 embeddings / noise from them), the oracle's
final latents in full (fp32), the oracle's u8 image IN FULL (``u8_full`` [H, W, 3]: every pixel is checked -- a u8 value pins the
oracle's [0, 1] float to +-0.5/255, which the test subtracts from the 1e-2 tolerance), and the decoded float image on a stride-4
pixel grid (fp16, [-1, 1] pre-clamp values clipped to +-4) for an exact float comparison on 1/16 of the pixels.
The oracle is this repo's CPU restatement (oracle/, parity unpinned at the diffusers boundary: oracle/__init__.py); the
weights are the seeded synthetic ones both sides generate.  Small sizes keep a live oracle run in the tests.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sdlcm_amd  # noqa: E402,F401
from sdlcm_amd import weights  # noqa: E402
from oracle.pipeline import LCMPipelineOracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STRIDE = 4


def grid(img_nchw):
    """[1, 3, H, W] -> the stride-4 grid [3, H/4, W/4]."""
    return img_nchw[0][:, ::STRIDE, ::STRIDE]


def save(case, ref, **meta):
    img = np.clip(ref["image"], -4, 4)
    u8 = ref["image_u8"]                       # [1, H, W, 3]
    np.savez_compressed(os.path.join(HERE, f"oracle_{case}.npz"),
                        latents=ref["latents"].astype(np.float32), image_grid=grid(img).astype(np.float16),
                        u8_grid=u8[0][::STRIDE, ::STRIDE].copy(), u8_full=u8[0].copy(), stride=np.int32(STRIDE),
                        **{k: np.asarray(v) for k, v in meta.items()})
    print(f"[golden] {case}: latents {ref['latents'].shape}, grid {grid(img).shape}, "
          f"{os.path.getsize(os.path.join(HERE, f'oracle_{case}.npz')) / 1e3:.0f} KB", flush=True)


def embeds(B, D=768, seed=5):
    return torch.randn(B, 77, D, generator=torch.Generator().manual_seed(seed)).to(torch.float16)


def sd15_cases(which):
    ora = LCMPipelineOracle(weights.synthetic_unet(), weights.synthetic_vae())
    if "sd15_512_4step" in which:          # tests/test_configs_gpu.py::test_config1
        t0 = time.time()
        save("sd15_512_4step", ora(embeds(1, seed=42).float(), 512, 512, 4, 1.0, 42), pe_seed=42, seed=42, steps=4, size=512)
        print(f"  {time.time() - t0:.0f} s", flush=True)
    if "sd15_512_b8_1step" in which:       # test_config2: requests 0 and 7 of the batch, run per request
        pe = embeds(8, seed=77)
        for i in (0, 7):
            save(f"sd15_512_b8_1step_req{i}", ora(pe[i:i + 1].float(), 512, 512, 1, 1.0, 500 + i), pe_seed=77, seed=500 + i, steps=1, size=512)
    if "sd15_768_8step" in which:          # test_config3
        t0 = time.time()
        save("sd15_768_8step", ora(embeds(1, seed=9).float(), 768, 768, 8, 1.0, 31), pe_seed=9, seed=31, steps=8, size=768)
        print(f"  {time.time() - t0:.0f} s", flush=True)


def sdxl_cases(which):
    from sdlcm_amd.config import SDXL_UNET, unet_config, vae_config
    ucfg = unet_config(SDXL_UNET)
    vcfg = vae_config(dict(scaling_factor=0.13025, sample_size=1024, force_upcast=True))
    ora = LCMPipelineOracle(weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0),
                            weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1), ucfg, vcfg)
    g = torch.Generator().manual_seed(8)
    pe = torch.randn(1, 77, 2048, generator=g).half()
    pooled = torch.randn(1, 1280, generator=g).half()
    tids = torch.tensor([[1024.0, 1024.0, 0, 0, 1024.0, 1024.0]])
    for guidance in (1.0, 5.0):
        case = f"sdxl_1024_g{int(guidance)}_1step"
        if case not in which:
            continue
        okw = dict(added=(pooled.float(), tids))
        if guidance > 1:
            okw.update(negative_embeds=torch.zeros_like(pe).float(), negative_added=(torch.zeros_like(pooled).float(), tids))
        t0 = time.time()
        save(case, ora(pe.float(), 1024, 1024, 1, guidance, 21, **okw), gen_seed=8, seed=21, steps=1, size=1024, guidance=guidance)
        print(f"  {time.time() - t0:.0f} s", flush=True)


ALL = ("sd15_512_4step", "sd15_512_b8_1step", "sd15_768_8step", "sdxl_1024_g1_1step", "sdxl_1024_g5_1step")
if __name__ == "__main__":
    which = set(sys.argv[1:]) or set(ALL)
    torch.set_num_threads(int(os.environ.get("LCM_CPU_THREADS", "8")))
    with torch.inference_mode():
        if any(c.startswith("sd15") for c in which):
            sd15_cases(which)
        if any(c.startswith("sdxl") for c in which):
            sdxl_cases(which)
