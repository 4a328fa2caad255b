"""Prompt -> encoder_hidden_states [B, 77, 768] for the UNet cross-attention.

The text encoder itself runs on the HIP kernels (``clip.ClipTextHip``).  Tokenisation is host string work: the CLIP BPE
vocabulary of the checkpoint (``tokenizer/`` / ``tokenizer_2/``) through ``transformers.CLIPTokenizer`` -- padding /
truncation to 77 and int32 ids as backends/rknnlcm.py:305-324; the pad id is the directory's own (SDXL's second tokenizer
pads with id 0).  A single-file checkpoint carries no vocabulary (the reference's ``from_single_file`` fetches it,
backends/cuda_worker.py:78-85; there is no network here): point ``LCM_TOKENIZER_DIR`` at a directory holding ``tokenizer/``
(and ``tokenizer_2/`` for SDXL) or the vocabulary files themselves.  REAL text-encoder weights without a vocabulary are an
error -- hashed word ids would silently condition the image on garbage; only the synthetic-weight runs (no vocabulary
ships with the reference) use ``clip.HashTokenizer``, which just needs stable ids.
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


def make_tokenizer(ckpt_root, sub, vocab_size, real_weights):
    """The tokenizer that goes with a text encoder: ``<ckpt_root>/<sub>``, else ``$LCM_TOKENIZER_DIR/<sub>``, else
    ``$LCM_TOKENIZER_DIR`` itself (first tokenizer only); with synthetic weights and none of those, the hash stand-in."""
    from .lib import LcmHipError
    cands = [os.path.join(ckpt_root, sub)] if ckpt_root else []
    env = os.environ.get("LCM_TOKENIZER_DIR", "")
    if env:
        cands.append(os.path.join(env, sub))
        if sub == "tokenizer":
            cands.append(env)
    for d in cands:
        if os.path.isfile(os.path.join(d, "vocab.json")) or os.path.isfile(os.path.join(d, "tokenizer.json")):
            return _BpeTokenizer(d)
    if real_weights:
        raise LcmHipError(f"text encoder weights were loaded from a checkpoint but no CLIP vocabulary was found for '{sub}' "
                          f"(looked in {cands or ['<no checkpoint directory>']}): a single-file checkpoint does not carry one -- set "
                          f"LCM_TOKENIZER_DIR to a directory holding {sub}/vocab.json + merges.txt.  Refusing to tokenise real "
                          f"weights with hashed word ids.")
    return HashTokenizer(vocab_size)


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
        self.tokenize = make_tokenizer(ckpt_root, "tokenizer", self.enc.cfg["vocab_size"], self.source != "synthetic")

    def __call__(self, prompts):
        return self.enc.forward(self.tokenize(prompts))
