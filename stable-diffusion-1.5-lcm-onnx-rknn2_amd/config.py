"""Architecture constants of the hot path (SD1.5 UNet2DConditionModel + AutoencoderKL decoder).

Key names follow the checkpoints' ``config.json`` (the diffusers-dir layout the reference
loads at backends/cuda_worker.py:70-77 and inspects at utils/model_detector.py:293-331).
SURVEY.md Appendix A.4 / A.5.
"""
from __future__ import annotations

SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, attention_head_dim=8, cross_attention_dim=768,
    norm_num_groups=32, norm_eps=1e-5, time_cond_proj_dim=256,
    down_attn=(True, True, True, False),
)

SD15_VAE = dict(
    latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512,
)

TEXT_SEQ_LEN = 77          # CLIP model_max_length (backends/rknnlcm.py:305-312)
VAE_SCALE_FACTOR = 8       # backends/rknnlcm.py:208-209


def unet_config(overrides: dict | None = None) -> dict:
    c = dict(SD15_UNET)
    if overrides:
        c.update(overrides)
    return c


def vae_config(overrides: dict | None = None) -> dict:
    c = dict(SD15_VAE)
    if overrides:
        c.update(overrides)
    return c
