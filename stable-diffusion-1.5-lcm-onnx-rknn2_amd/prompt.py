"""Prompt -> encoder_hidden_states [77, 768] for the UNet cross-attention.

SURVEY.md section 8 row (a6)/(f1): the CLIP text encoder is the NEXT row after the UNet/VAE hot path.
Until it is native, two host-side providers exist:

* ``SyntheticPromptEncoder`` -- no tokenizer vocabulary or weights ship with the reference
  (SURVEY.md 0.4), so synthetic-weight runs derive a deterministic unit-variance embedding from the
  prompt text (CLIP ends in a LayerNorm, so unit variance is representative).
* ``ClipPromptEncoder`` -- for a real diffusers-layout checkpoint: ``tokenizer/`` + ``text_encoder/``
  through ``transformers`` (torch ops; plumbing until row f1 lands), padded/truncated to 77 tokens as in
  backends/rknnlcm.py:305-312.
"""
from __future__ import annotations

import os
import zlib

import torch

from .config import TEXT_SEQ_LEN


class SyntheticPromptEncoder:
    def __init__(self, dim: int = 768):
        self.dim = dim

    def __call__(self, prompts):
        out = []
        for p in prompts:
            g = torch.Generator(device="cpu").manual_seed(zlib.crc32(str(p).encode("utf-8")) & 0x7FFFFFFF)
            out.append(torch.randn(TEXT_SEQ_LEN, self.dim, generator=g, dtype=torch.float32))
        return torch.stack(out).to(torch.float16)


class ClipPromptEncoder:
    def __init__(self, root: str, device):
        from transformers import CLIPTextModel, CLIPTokenizer
        self.tok = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
        self.enc = CLIPTextModel.from_pretrained(os.path.join(root, "text_encoder"), torch_dtype=torch.float16).to(device).eval()
        self.device = device

    @torch.inference_mode()
    def __call__(self, prompts):
        ids = self.tok(list(prompts), padding="max_length", max_length=TEXT_SEQ_LEN, truncation=True,
                       return_tensors="pt").input_ids.to(self.device)
        return self.enc(ids)[0].to(torch.float16)
