#!/usr/bin/env python3
"""Generate golden vectors for the pipeline glue from the reference itself.

Runs ONLY in the build container (needs /root/reference); the resulting
``glue_golden.npz`` is committed and is what the tests read.  The reference's
RKNN pipeline (backends/rknnlcm.py) is a numpy restatement of the diffusers LCM
pipeline glue; its NPU runtime (rknnlite) and diffusers are absent here, so the
two packages are registered as empty stub modules before import -- the same
``sys.modules`` technique the reference's own tests use
(tests/test_worker_pool.py:14-17).  Only pure-numpy functions are then called:

  RKNN2LatentConsistencyPipeline.get_guidance_scale_embedding  rknnlcm.py:651-677
  RKNN2LatentConsistencyPipeline.postprocess                   rknnlcm.py:211-264
  RKNN2LatentConsistencyPipeline.prepare_latents               rknnlcm.py:423-447
  rknn_worker._downsample_to_8x8_nchw                          rknn_worker.py:223-248
  rknn_worker.parse_size                                       rknn_worker.py:15-20

No reference source is copied: the npz holds inputs and outputs only.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def register_modules(self, **k):
            for n, v in k.items():
                setattr(self, n, v)

    _stub("rknnlite")
    _stub("rknnlite.api", RKNNLite=_Any)
    d = _stub("diffusers", LCMScheduler=_Any, StableDiffusionPipeline=_Any,
              StableDiffusionXLPipeline=_Any)
    _stub("diffusers.pipelines")
    _stub("diffusers.pipelines.pipeline_utils", DiffusionPipeline=_Any)
    _stub("diffusers.pipelines.stable_diffusion", StableDiffusionPipelineOutput=_Any)
    _stub("diffusers.schedulers", LCMScheduler=_Any)
    d.pipelines = sys.modules["diffusers.pipelines"]
    import transformers
    if not hasattr(transformers, "CLIPFeatureExtractor"):
        transformers.CLIPFeatureExtractor = object
    sys.path.insert(0, REF)
    import backends.rknnlcm as rknnlcm
    import backends.rknn_worker as rknn_worker
    return rknnlcm, rknn_worker


def main():
    rknnlcm, rknn_worker = import_reference()
    P = rknnlcm.RKNN2LatentConsistencyPipeline
    out = {}

    # guidance-scale embedding: w = guidance - 1
    ws = np.array([-1.0, 0.0, 0.5, 7.5], dtype=np.float32)
    pipe = P.__new__(P)
    out["gse_w"] = ws
    out["gse_256"] = pipe.get_guidance_scale_embedding(ws, embedding_dim=256, dtype=np.float32)
    out["gse_255"] = pipe.get_guidance_scale_embedding(ws, embedding_dim=255, dtype=np.float32)

    # postprocess: NCHW float in ~[-1,1] -> NHWC u8, incl. out-of-range + .5/255 ties
    rng = np.random.RandomState(7)
    img = rng.uniform(-1.3, 1.3, size=(2, 3, 5, 7)).astype(np.float32)
    ties = (np.arange(35, dtype=np.float32).reshape(5, 7) + 0.5) / 255.0 * 2 - 1
    img[0, 0] = ties
    out["post_in"] = img
    pils = P.postprocess(img, output_type="pil", do_denormalize=[True, True])
    out["post_u8"] = np.stack([np.asarray(p) for p in pils])
    out["post_np"] = P.postprocess(img, output_type="np", do_denormalize=[True, True])

    # prepare_latents with a torch CPU generator (seed stream contract)
    class _Sched:
        init_noise_sigma = 1.0
    pipe.scheduler = _Sched()
    pipe.vae_scale_factor = 8
    for seed, (h, w) in ((42, (512, 512)), (1234, (64, 64)), (7, (768, 512))):
        g = torch.Generator().manual_seed(seed)
        lat = pipe.prepare_latents(1, 4, h, w, np.float32, g)
        out[f"latents_seed{seed}_{w}x{h}"] = lat.astype(np.float32)
        # the next draw from the same generator = first re-noise sample of LCMScheduler.step
        out[f"noise1_seed{seed}_{w}x{h}"] = torch.randn(lat.shape, generator=g).numpy()

    # 8x8 latent downsample used by run_job_with_latents
    for hw in ((64, 64), (16, 24), (8, 8), (20, 12)):
        lat = rng.randn(1, 4, *hw).astype(np.float32)
        out[f"ds_in_{hw[0]}x{hw[1]}"] = lat
        out[f"ds_out_{hw[0]}x{hw[1]}"] = rknn_worker._downsample_to_8x8_nchw(lat).astype(np.float32)

    out["parse_size_512x768"] = np.array(rknn_worker.parse_size("512x768"))
    out["parse_size_64X64"] = np.array(rknn_worker.parse_size("64X64"))

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
