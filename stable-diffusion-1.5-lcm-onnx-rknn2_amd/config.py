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

# SDXL-base UNet (DiffusersSDXLCudaWorker, backends/cuda_worker.py:307-614):
# 3 levels, transformer depth 1/2/10 (no attention at level 0), 64-wide heads, linear proj_in/out,
# "text_time" additional embedding (pooled text 1280 + 6 x 256 sinusoid of the size/crop ids -> 2816 -> MLP).
SDXL_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280),
    layers_per_block=2, attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
    norm_num_groups=32, norm_eps=1e-5, time_cond_proj_dim=None,
    down_attn=(False, True, True), transformer_layers_per_block=(1, 2, 10), use_linear_projection=True,
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)

SD15_VAE = dict(
    latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512,
    force_upcast=False,        # SDXL VAE: True -> the decoder runs with its residual stream scaled by 1/16 (model.py)
    residual_scale=1.0,
)

TEXT_SEQ_LEN = 77          # CLIP model_max_length (backends/rknnlcm.py:305-312)
VAE_SCALE_FACTOR = 8       # backends/rknnlcm.py:208-209


def unet_config(overrides: dict | None = None) -> dict:
    c = dict(SD15_UNET)
    c.setdefault("transformer_layers_per_block", 1)
    c.setdefault("use_linear_projection", False)
    c.setdefault("addition_time_embed_dim", None)
    c.setdefault("projection_class_embeddings_input_dim", None)
    if overrides:
        c.update(overrides)
    return c


def heads_at(cfg: dict, level: int) -> int:
    """Number of attention heads at down-level `level` (diffusers' misnamed ``attention_head_dim``)."""
    h = cfg["attention_head_dim"]
    return int(h[level]) if isinstance(h, (tuple, list)) else int(h)


def depth_at(cfg: dict, level: int) -> int:
    """BasicTransformerBlocks per Transformer2DModel at down-level `level`."""
    t = cfg.get("transformer_layers_per_block", 1)
    return int(t[level]) if isinstance(t, (tuple, list)) else int(t)


def vae_config(overrides: dict | None = None) -> dict:
    c = dict(SD15_VAE)
    if overrides:
        c.update(overrides)
    if c.get("force_upcast") and float(c.get("residual_scale", 1.0)) == 1.0:
        c["residual_scale"] = 1.0 / 16.0       # the fp16 stand-in for the reference's fp32 upcast of the SDXL VAE
    return c
