"""CPU oracle for the LCM Stable-Diffusion hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU fp32 / numpy restatement of the algorithm the
reference runs through ``self.pipe(...)`` at backends/cuda_worker.py:221-229
(diffusers ``StableDiffusionPipeline`` + ``LCMScheduler``) and of the numpy twin
of the same glue in backends/rknnlcm.py:450-677.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``stable-diffusion-1.5-lcm-onnx-rknn2_amd/``) never imports it and fails loudly
when the HIP extension is missing.

Parity status
-------------
* glue (guidance embedding, latent preparation / RNG stream, post-process,
  8x8 latent blob, size parsing): PINNED against vectors produced by the
  reference's own numpy functions (tests/golden/make_golden.py ->
  tests/golden/glue_golden.npz).
* LCMScheduler tables / step, UNet2DConditionModel, AutoencoderKL.decode:
  the arithmetic lives in ``diffusers`` (un-pinned in requirements.txt:31, not
  installed, no network) and the reference's tests hold no numeric vector for
  it (tests/test_sdxl_worker.py checks only PNG magic / determinism).  These are
  restated from the published diffusers algorithm and checked against
  closed-form known answers (timesteps [999,759,499,259], boundary-condition
  scalings, ...).  **parity unpinned** at the diffusers boundary.
"""
