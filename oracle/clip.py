"""CLIP text encoder checker: ``transformers.CLIPTextModel`` itself (CPU, fp32), instantiated from a config (no
download) and loaded with the same weights as the HIP path.  Oracle only -- the reference calls exactly this class
inside the pipeline (backends/cuda_worker.py:221-229; numpy twin of the call backends/rknnlcm.py:266-367)."""
from __future__ import annotations

import torch


def clip_text_oracle(sd: dict, cfg: dict, ids: torch.Tensor) -> torch.Tensor:
    from transformers import CLIPTextConfig, CLIPTextModel
    c = CLIPTextConfig(**cfg)
    m = CLIPTextModel(c).eval().float()
    want = m.state_dict()
    src = {}
    for k in want:
        kk = k[len("text_model."):] if k.startswith("text_model.") else k
        if kk in sd:
            src[k] = sd[kk].float()
    missing = [k for k in want if k not in src and "position_ids" not in k]
    if missing:
        raise RuntimeError(f"CLIP oracle: weights missing for {missing[:4]}")
    m.load_state_dict(src, strict=False)
    with torch.inference_mode():
        return m(input_ids=ids.long()).last_hidden_state


def clip_text_oracle_sdxl(sd: dict, cfg: dict, ids: torch.Tensor):
    """-> (hidden_states[-2], text_embeds or None) from transformers' CLIPTextModel / CLIPTextModelWithProjection."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    proj = "text_projection.weight" in sd
    c = CLIPTextConfig(**cfg)
    m = (CLIPTextModelWithProjection(c) if proj else CLIPTextModel(c)).eval().float()
    want = m.state_dict()
    src = {}
    for k in want:
        kk = k[len("text_model."):] if k.startswith("text_model.") else k
        if kk in sd:
            src[k] = sd[kk].float()
    missing = [k for k in want if k not in src and "position_ids" not in k]
    if missing:
        raise RuntimeError(f"CLIP oracle: weights missing for {missing[:4]}")
    m.load_state_dict(src, strict=False)
    with torch.inference_mode():
        o = m(input_ids=ids.long(), output_hidden_states=True)
    return o.hidden_states[-2], (o.text_embeds if proj else None)
