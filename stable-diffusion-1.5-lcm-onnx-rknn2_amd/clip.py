"""CLIP text encoder (the prompt -> encoder_hidden_states stage of the hot path) on the MI355X kernels.

SURVEY.md section 8 row a6 / f1: ``transformers.CLIPTextModel`` as called inside the pipeline at
backends/cuda_worker.py:221-229 (numpy twin of the call: backends/rknnlcm.py:266-367 -- tokenize with padding /
truncation to 77, ids as int32, take last_hidden_state).  12 pre-LN layers, hidden 768, 12 heads x 64, causal
attention, quick_gelu MLP 3072, final LayerNorm; built from the same kernels as the UNet: token+position gather,
LayerNorm, MFMA GEMM (fused q|k|v, bias / quick-GELU / residual epilogues) and the flash attention kernel with its
causal mask.  Python only sequences launches.
"""
from __future__ import annotations

import os
import zlib

import torch

from . import ops
from .config import TEXT_SEQ_LEN

CLIP_GRAPH = os.environ.get("LCM_CLIP_GRAPH", "1") != "0"

# OpenCLIP ViT-bigG/14 text tower (SDXL text_encoder_2, CLIPTextModelWithProjection): 32 layers, 1280 wide, exact GELU,
# text_projection 1280 -> 1280 on the pooled (EOS) token.
CLIP_BIGG = dict(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                 max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, projection_dim=1280)

CLIP_L = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
              max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5)


def clip_param_spec(cfg=None):
    c = dict(CLIP_L)
    c.update(cfg or {})
    D, F = c["hidden_size"], c["intermediate_size"]
    yield "embeddings.token_embedding.weight", (c["vocab_size"], D), "emb"
    yield "embeddings.position_embedding.weight", (c["max_position_embeddings"], D), "emb"
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield f"{p}.self_attn.{n}.weight", (D, D), "w_res" if n == "out_proj" else "w"
            yield f"{p}.self_attn.{n}.bias", (D,), "bias"
        for n in ("layer_norm1", "layer_norm2"):
            yield f"{p}.{n}.weight", (D,), "gamma"
            yield f"{p}.{n}.bias", (D,), "beta"
        yield f"{p}.mlp.fc1.weight", (F, D), "w"
        yield f"{p}.mlp.fc1.bias", (F,), "bias"
        yield f"{p}.mlp.fc2.weight", (D, F), "w_res"
        yield f"{p}.mlp.fc2.bias", (D,), "bias"
    yield "final_layer_norm.weight", (D,), "gamma"
    yield "final_layer_norm.bias", (D,), "beta"
    if c.get("projection_dim"):
        yield "text_projection.weight", (c["projection_dim"], D), "w"


def synthetic_clip(cfg=None, seed=2):
    from .weights import synthetic_state_dict
    spec = [(n, s, "w" if k == "emb" else k) for n, s, k in clip_param_spec(cfg)]
    sd = synthetic_state_dict(spec, seed)
    # embeddings: unit-ish variance rows (synthetic_state_dict scaled them by 1/sqrt(D))
    for n in ("embeddings.token_embedding.weight", "embeddings.position_embedding.weight"):
        sd[n] = (sd[n].float() * (sd[n].shape[1] ** 0.5) * 0.5).to(torch.float16)
    return sd


def load_clip_dir(d: str):
    """text_encoder/ of a diffusers checkpoint -> (state dict without the 'text_model.' prefix, config)."""
    import json
    from safetensors.torch import load_file
    with open(os.path.join(d, "config.json")) as f:
        j = json.load(f)
    cfg = {k: j[k] for k in CLIP_L if k in j}
    for fn in ("model.fp16.safetensors", "model.safetensors"):
        p = os.path.join(d, fn)
        if os.path.exists(p):
            sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v.to(torch.float16)
                  for k, v in load_file(p).items()}
            if "text_projection.weight" in sd:
                cfg["projection_dim"] = sd["text_projection.weight"].shape[0]
            return sd, cfg
    raise FileNotFoundError(f"no model*.safetensors under {d}")


class ClipTextHip:
    def __init__(self, sd: dict, cfg=None, device="cuda"):
        self.cfg = c = dict(CLIP_L)
        c.update(cfg or {})
        if c["hidden_act"] not in ("quick_gelu", "gelu"):
            raise ValueError(f"unsupported CLIP hidden_act {c['hidden_act']!r}")
        self.device = torch.device(device)
        self.D, self.F, self.heads, self.L = c["hidden_size"], c["intermediate_size"], c["num_attention_heads"], c["num_hidden_layers"]
        self.act = 2 if c["hidden_act"] == "quick_gelu" else 3
        dev = lambda t: t.to(device=self.device, dtype=torch.float16).contiguous()
        self.w = {"tok": dev(sd["embeddings.token_embedding.weight"]), "pos": dev(sd["embeddings.position_embedding.weight"]),
                  "fln.g": dev(sd["final_layer_norm.weight"]), "fln.b": dev(sd["final_layer_norm.bias"])}
        for i in range(self.L):
            p = f"encoder.layers.{i}"
            self.w[f"{i}.qkv.w"] = dev(torch.cat([sd[f"{p}.self_attn.{n}_proj.weight"] for n in "qkv"], 0))
            self.w[f"{i}.qkv.b"] = dev(torch.cat([sd[f"{p}.self_attn.{n}_proj.bias"] for n in "qkv"], 0))
            self.w[f"{i}.o.w"], self.w[f"{i}.o.b"] = dev(sd[f"{p}.self_attn.out_proj.weight"]), dev(sd[f"{p}.self_attn.out_proj.bias"])
            for n, t in (("ln1", "layer_norm1"), ("ln2", "layer_norm2")):
                self.w[f"{i}.{n}.g"], self.w[f"{i}.{n}.b"] = dev(sd[f"{p}.{t}.weight"]), dev(sd[f"{p}.{t}.bias"])
            self.w[f"{i}.fc1.w"], self.w[f"{i}.fc1.b"] = dev(sd[f"{p}.mlp.fc1.weight"]), dev(sd[f"{p}.mlp.fc1.bias"])
            self.w[f"{i}.fc2.w"], self.w[f"{i}.fc2.b"] = dev(sd[f"{p}.mlp.fc2.weight"]), dev(sd[f"{p}.mlp.fc2.bias"])
        self.has_proj = "text_projection.weight" in sd
        if self.has_proj:
            self.w["proj"] = dev(sd["text_projection.weight"])
        self._buf = {}
        self._graphs, self._ids = {}, {}

    def _b(self, name, *shape):
        t = self._buf.get((name, shape))
        if t is None:
            t = torch.empty(shape, dtype=torch.float16, device=self.device)
            self._buf[(name, shape)] = t
        return t

    def weight_bytes(self):
        return sum(t.numel() * 2 for t in self.w.values())

    def view(self):
        """Same encoder (shared weights), own activation buffers: one per pipeline lane."""
        v = object.__new__(type(self))
        v.__dict__.update(self.__dict__)
        v._buf = {}
        v._graphs, v._ids = {}, {}
        return v

    @torch.inference_mode()
    def forward(self, ids: torch.Tensor, output="last", pooled=False):
        """ids: int [B, S<=77] (host or device).
        output="last": final_layer_norm(last layer)           -> fp16 [B, S, D]   (SD1.5: last_hidden_state)
        output="penultimate": hidden_states[-2], no final LN   -> fp16 [B, S, D]   (SDXL, both encoders)
        pooled=True additionally returns text_projection(final_layer_norm(last layer)[EOS token]) fp16 [B, P]
        (CLIPTextModelWithProjection.text_embeds; EOS = position of the largest id, as transformers does for CLIP).

        The SD1.5 form (output="last", no pooled vector) replays a hipGraph per (B, S): 86 launches through ctypes cost the
        request's thread ~1 ms, one graph launch ~0.05 (LCM_CLIP_GRAPH=0: always eager).  Same kernels, same launch
        parameters, same bits."""
        B, S = ids.shape
        # (the legacy default stream cannot be captured: callers outside a lane's stream run eagerly)
        if CLIP_GRAPH and output == "last" and not pooled and torch.cuda.current_stream(self.device) != torch.cuda.default_stream(self.device):
            key = (B, S)
            ids_d = self._ids.get(key)
            if ids_d is None:
                ids_d = self._ids[key] = torch.empty(B, S, dtype=torch.int32, device=self.device)
            ids_d.copy_(ids.to(dtype=torch.int32), non_blocking=False)
            g = self._graphs.get(key)
            if g is None:
                self._encode(ids_d, B, S, "last")                  # eager once: allocates the activation buffers
                g = ops.Graph()
                with g:
                    self._encode(ids_d, B, S, "last")
                self._graphs[key] = g
            g.launch()
            return self._b("fin", B * S, self.D).reshape(B, S, self.D).clone()     # the buffer belongs to the next call
        ids_d = ids.to(device=self.device, dtype=torch.int32).contiguous()
        hidden, fin = self._encode(ids_d, B, S, output, want_fin=pooled)
        if output == "last":
            hidden = hidden.clone()
        if not pooled:
            return hidden
        if not self.has_proj:
            raise ValueError("pooled output needs text_projection weights (CLIPTextModelWithProjection)")
        D = self.D
        eos = ids.to("cpu").long().argmax(dim=-1) + torch.arange(B) * S          # row of the EOS token per prompt
        rows = fin.index_select(0, eos.to(self.device)).contiguous()               # gather = data movement only
        P = self.w["proj"].shape[0]
        te = torch.empty(B, P, dtype=torch.float16, device=self.device)
        for b0 in range(0, B, 16):
            nb = min(16, B - b0)
            ops.linear_smallm(rows[b0:b0 + nb], self.w["proj"], te[b0:b0 + nb], nb, P, D)
        return hidden, te

    def _encode(self, ids_d, B, S, output, want_fin=False):
        """The kernel sequence (what a graph captures).  -> (hidden [B,S,D], final-LN rows [M,D] or None); both in buffers
        of this view."""
        D, F, H = self.D, self.F, self.heads
        M, d = B * S, D // H
        w = self.w
        x = self._b("x", M, D)
        ops.embed_tokens(ids_d, w["tok"], w["pos"], x, B, S, D)
        n, qkv, a, h = self._b("n", M, D), self._b("qkv", M, 3 * D), self._b("a", M, D), self._b("h", M, F)
        pooled = want_fin
        pen = None
        last_needed = self.L if (output == "last" or pooled) else self.L - 1
        for i in range(last_needed):
            if output == "penultimate" and i == self.L - 1:
                pen = x.clone()                                    # hidden_states[-2]: input of the last layer (device copy)
            ops.layernorm(x, w[f"{i}.ln1.g"], w[f"{i}.ln1.b"], n, M, D, self.cfg["layer_norm_eps"])
            ops.gemm(n, w[f"{i}.qkv.w"], qkv, bias=w[f"{i}.qkv.b"], img_rows=S)
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, B, H, S, S, d, ldq=3 * D, ldk=3 * D, ldv=3 * D,
                          ldo=D, causal=True)
            ops.gemm(a, w[f"{i}.o.w"], x, bias=w[f"{i}.o.b"], res=x, img_rows=S)
            ops.layernorm(x, w[f"{i}.ln2.g"], w[f"{i}.ln2.b"], n, M, D, self.cfg["layer_norm_eps"])
            ops.gemm(n, w[f"{i}.fc1.w"], h, bias=w[f"{i}.fc1.b"], epilogue=self.act, img_rows=S)
            ops.gemm(h, w[f"{i}.fc2.w"], x, bias=w[f"{i}.fc2.b"], res=x, img_rows=S)
        if output == "penultimate":
            hidden = (pen if pen is not None else x.clone()).reshape(B, S, D)
        else:
            hidden = None
        fin = None
        if output == "last" or pooled:
            fin = self._b("fin", M, D)
            ops.layernorm(x, w["fln.g"], w["fln.b"], fin, M, D, self.cfg["layer_norm_eps"])
            if output == "last":
                hidden = fin.reshape(B, S, D)
        return hidden, fin


class HashTokenizer:
    """Stand-in tokenizer for synthetic-weight runs (no vocabulary ships with the reference): words map
    to stable ids by CRC32; BOS 49406, EOS/pad 49407, padded / truncated to 77 as backends/rknnlcm.py:305-312."""

    def __init__(self, vocab_size=49408, bos=49406, eos=49407):
        self.vocab, self.bos, self.eos = vocab_size, bos, eos

    def __call__(self, prompts):
        rows = []
        for p in prompts:
            toks = [zlib.crc32(w.encode("utf-8")) % (self.bos - 1) + 1 for w in str(p).lower().split()]
            ids = [self.bos] + toks[:TEXT_SEQ_LEN - 2] + [self.eos]
            rows.append(ids + [self.eos] * (TEXT_SEQ_LEN - len(ids)))
        return torch.tensor(rows, dtype=torch.int32)
