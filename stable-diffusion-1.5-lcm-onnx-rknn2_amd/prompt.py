"""Prompt -> encoder_hidden_states [B, 77, 768] for the UNet cross-attention.

The text encoder itself runs on the HIP kernels (``clip.ClipTextHip``).  Tokenisation is host string work:
the checkpoint's ``tokenizer/`` (CLIP BPE vocabulary) through ``transformers.CLIPTokenizer`` when present --
padding / truncation to 77 and int32 ids as backends/rknnlcm.py:305-324 -- else ``clip.HashTokenizer`` (no
vocabulary ships with the reference; synthetic-weight runs only need stable ids).
"""
from __future__ import annotations

import os

import torch

from .clip import ClipTextHip, HashTokenizer, load_clip_dir, synthetic_clip
from .config import TEXT_SEQ_LEN


class _BpeTokenizer:
    def __init__(self, d: str):
        from transformers import CLIPTokenizer
        self.tok = CLIPTokenizer.from_pretrained(d)

    def __call__(self, prompts):
        ids = self.tok(list(prompts), padding="max_length", max_length=TEXT_SEQ_LEN, truncation=True,
                       return_tensors="pt").input_ids
        return ids.to(torch.int32)


class HipPromptEncoder:
    """prompts (list[str]) -> fp16 [B, 77, D] on the device, computed by the native CLIP text encoder."""

    def __init__(self, device, ckpt_root: str | None = None, clip_sd: dict | None = None):
        te = os.path.join(ckpt_root, "text_encoder") if ckpt_root else None
        if clip_sd is not None:                      # text encoder carried inside a single-file checkpoint
            sd = clip_sd
            cfg = dict(num_hidden_layers=1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers.")),
                       hidden_size=sd["embeddings.token_embedding.weight"].shape[1],
                       vocab_size=sd["embeddings.token_embedding.weight"].shape[0],
                       intermediate_size=sd["encoder.layers.0.mlp.fc1.weight"].shape[0],
                       num_attention_heads=sd["embeddings.token_embedding.weight"].shape[1] // 64)
            self.source = "single-file"
        elif te and os.path.isdir(te):
            sd, cfg = load_clip_dir(te)
            self.source = "checkpoint"
        else:
            sd, cfg = synthetic_clip(), None
            self.source = "synthetic"
        self.enc = ClipTextHip(sd, cfg, device=device)
        tk = os.path.join(ckpt_root, "tokenizer") if ckpt_root else None
        if tk and os.path.isdir(tk):
            self.tokenize = _BpeTokenizer(tk)
        else:
            self.tokenize = HashTokenizer(self.enc.cfg["vocab_size"])

    def __call__(self, prompts):
        return self.enc.forward(self.tokenize(prompts))
