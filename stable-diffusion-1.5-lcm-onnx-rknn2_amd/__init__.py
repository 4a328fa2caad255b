"""MI355X-native LCM Stable-Diffusion-1.5 backend (gfx950 HIP kernels behind the reference's
``PipelineWorker`` interface, backends/base.py:29-39).  Import as ``sdlcm_amd`` (see /sdlcm_amd.py)."""
__version__ = "0.1.0"
