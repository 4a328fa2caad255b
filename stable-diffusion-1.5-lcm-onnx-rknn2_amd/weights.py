"""Parameter inventory, synthetic weights and checkpoint loading for the hot path.

No checkpoint ships with the reference, so parity and perf runs use
seeded synthetic weights of the real architecture; a real diffusers-dir checkpoint
(``unet/diffusion_pytorch_model.safetensors`` + ``vae/...``; layout per
backends/base.py:44-58, backends/cuda_worker.py:66-77) loads through the same names.

State dicts use the diffusers parameter names; values are fp16 (the precision the reference
runs at, CUDA_DTYPE default fp16 -- backends/cuda_worker.py:55-61).
"""
from __future__ import annotations

import json
import os

import torch

from .config import depth_at, unet_config, vae_config

# kind: conv (OIHW), lin (out,in), vec (bias / norm affine)


def _resnet(p, cin, cout, temb):
    yield p + ".norm1.weight", (cin,), "gamma"
    yield p + ".norm1.bias", (cin,), "beta"
    yield p + ".conv1.weight", (cout, cin, 3, 3), "w"
    yield p + ".conv1.bias", (cout,), "bias"
    if temb:
        yield p + ".time_emb_proj.weight", (cout, temb), "w"
        yield p + ".time_emb_proj.bias", (cout,), "bias"
    yield p + ".norm2.weight", (cout,), "gamma"
    yield p + ".norm2.bias", (cout,), "beta"
    yield p + ".conv2.weight", (cout, cout, 3, 3), "w_res"
    yield p + ".conv2.bias", (cout,), "bias"
    if cin != cout:
        yield p + ".conv_shortcut.weight", (cout, cin, 1, 1), "w"
        yield p + ".conv_shortcut.bias", (cout,), "bias"


def _transformer(p, c, ctx, depth=1, linear=False):
    pshape = (c, c) if linear else (c, c, 1, 1)
    yield p + ".norm.weight", (c,), "gamma"
    yield p + ".norm.bias", (c,), "beta"
    yield p + ".proj_in.weight", pshape, "w"
    yield p + ".proj_in.bias", (c,), "bias"
    for k in range(depth):
        t = f"{p}.transformer_blocks.{k}"
        for n in ("norm1", "norm2", "norm3"):
            yield f"{t}.{n}.weight", (c,), "gamma"
            yield f"{t}.{n}.bias", (c,), "beta"
        for a, kdim in (("attn1", c), ("attn2", ctx)):
            yield f"{t}.{a}.to_q.weight", (c, c), "w"
            yield f"{t}.{a}.to_k.weight", (c, kdim), "w"
            yield f"{t}.{a}.to_v.weight", (c, kdim), "w"
            yield f"{t}.{a}.to_out.0.weight", (c, c), "w_res"
            yield f"{t}.{a}.to_out.0.bias", (c,), "bias"
        yield f"{t}.ff.net.0.proj.weight", (8 * c, c), "w"
        yield f"{t}.ff.net.0.proj.bias", (8 * c,), "bias"
        yield f"{t}.ff.net.2.weight", (c, 4 * c), "w_res"
        yield f"{t}.ff.net.2.bias", (c,), "bias"
    yield p + ".proj_out.weight", pshape, "w_res"
    yield p + ".proj_out.bias", (c,), "bias"


def unet_param_spec(cfg: dict | None = None):
    cfg = unet_config(cfg)
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    yield "conv_in.weight", (boc[0], cfg["in_channels"], 3, 3), "w"
    yield "conv_in.bias", (boc[0],), "bias"
    yield "time_embedding.linear_1.weight", (temb, boc[0]), "w"
    yield "time_embedding.linear_1.bias", (temb,), "bias"
    yield "time_embedding.linear_2.weight", (temb, temb), "w"
    yield "time_embedding.linear_2.bias", (temb,), "bias"
    if cfg.get("time_cond_proj_dim"):
        yield "time_embedding.cond_proj.weight", (boc[0], cfg["time_cond_proj_dim"]), "w"
    if cfg.get("addition_time_embed_dim"):
        yield "add_embedding.linear_1.weight", (temb, cfg["projection_class_embeddings_input_dim"]), "w"
        yield "add_embedding.linear_1.bias", (temb,), "bias"
        yield "add_embedding.linear_2.weight", (temb, temb), "w"
        yield "add_embedding.linear_2.bias", (temb,), "bias"
    nb = len(boc)
    lin = bool(cfg.get("use_linear_projection"))
    skip_ch = [boc[0]]
    ch = boc[0]
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            yield from _resnet(f"down_blocks.{i}.resnets.{j}", ch, boc[i], temb)
            ch = boc[i]
            if cfg["down_attn"][i]:
                yield from _transformer(f"down_blocks.{i}.attentions.{j}", ch, ctx, depth_at(cfg, i), lin)
            skip_ch.append(ch)
        if i < nb - 1:
            yield f"down_blocks.{i}.downsamplers.0.conv.weight", (ch, ch, 3, 3), "w"
            yield f"down_blocks.{i}.downsamplers.0.conv.bias", (ch,), "bias"
            skip_ch.append(ch)
    yield from _resnet("mid_block.resnets.0", ch, ch, temb)
    yield from _transformer("mid_block.attentions.0", ch, ctx, depth_at(cfg, nb - 1), lin)
    yield from _resnet("mid_block.resnets.1", ch, ch, temb)
    rboc = tuple(reversed(boc))
    up_attn = tuple(reversed(cfg["down_attn"]))
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            s = skip_ch.pop()
            yield from _resnet(f"up_blocks.{i}.resnets.{j}", ch + s, rboc[i], temb)
            ch = rboc[i]
            if up_attn[i]:
                yield from _transformer(f"up_blocks.{i}.attentions.{j}", ch, ctx, depth_at(cfg, nb - 1 - i), lin)
        if i < nb - 1:
            yield f"up_blocks.{i}.upsamplers.0.conv.weight", (ch, ch, 3, 3), "w"
            yield f"up_blocks.{i}.upsamplers.0.conv.bias", (ch,), "bias"
    yield "conv_norm_out.weight", (ch,), "gamma"
    yield "conv_norm_out.bias", (ch,), "beta"
    yield "conv_out.weight", (cfg["out_channels"], ch, 3, 3), "w_out"
    yield "conv_out.bias", (cfg["out_channels"],), "bias"


def vae_param_spec(cfg: dict | None = None):
    cfg = vae_config(cfg)
    boc = cfg["block_out_channels"]
    lc = cfg["latent_channels"]
    yield "post_quant_conv.weight", (lc, lc, 1, 1), "w"
    yield "post_quant_conv.bias", (lc,), "bias"
    top = boc[-1]
    yield "decoder.conv_in.weight", (top, lc, 3, 3), "w"
    yield "decoder.conv_in.bias", (top,), "bias"
    yield from _resnet("decoder.mid_block.resnets.0", top, top, 0)
    a = "decoder.mid_block.attentions.0"
    yield a + ".group_norm.weight", (top,), "gamma"
    yield a + ".group_norm.bias", (top,), "beta"
    for n in ("to_q", "to_k", "to_v"):
        yield f"{a}.{n}.weight", (top, top), "w"
        yield f"{a}.{n}.bias", (top,), "bias"
    yield a + ".to_out.0.weight", (top, top), "w_res"
    yield a + ".to_out.0.bias", (top,), "bias"
    yield from _resnet("decoder.mid_block.resnets.1", top, top, 0)
    ch = top
    rboc = tuple(reversed(boc))
    nb = len(boc)
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            yield from _resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, rboc[i], 0)
            ch = rboc[i]
        if i < nb - 1:
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (ch, ch, 3, 3), "w"
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (ch,), "bias"
    yield "decoder.conv_norm_out.weight", (ch,), "gamma"
    yield "decoder.conv_norm_out.bias", (ch,), "beta"
    yield "decoder.conv_out.weight", (cfg["out_channels"], ch, 3, 3), "w_out"
    yield "decoder.conv_out.bias", (cfg["out_channels"],), "bias"


def count_params(spec) -> int:
    n = 0
    for _, shape, _ in spec:
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


def synthetic_state_dict(spec, seed: int, res_scale: float = 0.25) -> dict:
    """Seeded, variance-preserving weights stored as fp16.

    w: N(0, 1/fan_in); residual-branch output layers additionally x ``res_scale`` so the
    4-step feedback loop stays well-conditioned; norm gamma ~ 1 +- 0.1, beta ~ +-0.1.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape, kind in spec:
        if kind in ("w", "w_res", "w_out"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = fan_in ** -0.5
            if kind == "w_res":
                std *= res_scale
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif kind == "gamma":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == "beta":
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:  # bias
            t = 0.02 * torch.randn(shape, generator=g, dtype=torch.float32)
        sd[name] = t.to(torch.float16)
    return sd


def synthetic_unet(cfg=None, seed=0):
    return synthetic_state_dict(unet_param_spec(cfg), seed)


def synthetic_vae(cfg=None, seed=1):
    return synthetic_state_dict(vae_param_spec(cfg), seed)


# ---------------------------------------------------------------------------------------
# real checkpoints (diffusers directory layout)
# ---------------------------------------------------------------------------------------
def _load_safetensors_dir(d: str) -> dict:
    from safetensors.torch import load_file
    for fn in ("diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors"):
        p = os.path.join(d, fn)
        if os.path.exists(p):
            return {k: v.to(torch.float16) for k, v in load_file(p).items()}
    raise FileNotFoundError(f"no diffusion_pytorch_model*.safetensors under {d}")


def load_diffusers_dir(root: str):
    """-> (unet_sd, unet_cfg, vae_sd, vae_cfg) from a diffusers-layout checkpoint directory."""
    def cfg_of(sub, base):
        with open(os.path.join(root, sub, "config.json")) as f:
            j = json.load(f)
        out = dict(base)
        for k in base:
            if k in j and j[k] is not None:
                out[k] = tuple(j[k]) if isinstance(j[k], list) else j[k]
        return j, out

    ju, ucfg = cfg_of("unet", unet_config())
    ucfg["time_cond_proj_dim"] = ju.get("time_cond_proj_dim")
    if "down_block_types" in ju:
        ucfg["down_attn"] = tuple("CrossAttn" in t for t in ju["down_block_types"])
    _, vcfg = cfg_of("vae", vae_config())
    usd = _load_safetensors_dir(os.path.join(root, "unet"))
    vsd = _load_safetensors_dir(os.path.join(root, "vae"))
    vsd = {k: v for k, v in vsd.items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    # older VAE checkpoints name the mid attention query/key/value/proj_attn
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    for k in list(vsd):
        for old, new in ren.items():
            tag = f".attentions.0.{old}."
            if tag in k:
                vsd[k.replace(tag, f".attentions.0.{new}.")] = vsd.pop(k)
    for name, shape, _ in list(unet_param_spec(ucfg)) :
        if name not in usd or tuple(usd[name].shape) != tuple(shape):
            raise RuntimeError(f"checkpoint/graph mismatch at unet '{name}': expected {shape}, "
                               f"got {tuple(usd[name].shape) if name in usd else None}")
    for name, shape, _ in list(vae_param_spec(vcfg)):
        if name not in vsd:
            raise RuntimeError(f"checkpoint/graph mismatch at vae '{name}'")
        if tuple(vsd[name].shape) != tuple(shape):
            vsd[name] = vsd[name].reshape(shape)   # linear attn stored as 1x1 conv or vice versa
    return usd, ucfg, vsd, vcfg


# ---------------------------------------------------------------------------------------
# single-file (LDM / original Stable Diffusion layout) checkpoints
# ---------------------------------------------------------------------------------------
_LDM_RES = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2",
            "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}


def _ldm_res(rest: str) -> str:
    for old, new in _LDM_RES.items():
        if rest.startswith(old + "."):
            return new + rest[len(old):]
    raise KeyError(rest)


def _ldm_unet_key(k: str, layers_per_block: int = 2, down_attn=(True, True, True, False)):
    """'model.diffusion_model.' key (prefix stripped) -> diffusers UNet2DConditionModel key (or None to drop).
    Same correspondence the reference relies on when it calls from_single_file (backends/cuda_worker.py:78-85) and
    inspects at utils/model_detector.py:232-284."""
    p = k.split(".")
    per = layers_per_block + 1
    if p[0] == "time_embed":
        return {"0": "time_embedding.linear_1", "2": "time_embedding.linear_2"}[p[1]] + "." + p[2]
    if p[0] == "label_emb":                                                   # SDXL text_time additional embedding
        return {"0": "add_embedding.linear_1", "2": "add_embedding.linear_2"}[p[2]] + "." + p[3]
    if p[0] == "input_blocks":
        i = int(p[1])
        if i == 0:
            return "conv_in." + p[3]
        b, l = (i - 1) // per, (i - 1) % per
        if l == layers_per_block:
            return f"down_blocks.{b}.downsamplers.0.conv." + p[-1]          # input_blocks.i.0.op.{weight,bias}
        if p[2] == "0":
            return f"down_blocks.{b}.resnets.{l}." + _ldm_res(".".join(p[3:]))
        return f"down_blocks.{b}.attentions.{l}." + ".".join(p[3:])
    if p[0] == "middle_block":
        j = int(p[1])
        if j == 1:
            return "mid_block.attentions.0." + ".".join(p[2:])
        return f"mid_block.resnets.{0 if j == 0 else 1}." + _ldm_res(".".join(p[2:]))
    if p[0] == "output_blocks":
        i = int(p[1])
        b, l = i // per, i % per
        up_attn = tuple(reversed(down_attn))
        if p[2] == "0":
            return f"up_blocks.{b}.resnets.{l}." + _ldm_res(".".join(p[3:]))
        if p[2] == "1" and up_attn[b] and p[3] != "conv":
            return f"up_blocks.{b}.attentions.{l}." + ".".join(p[3:])
        return f"up_blocks.{b}.upsamplers.0.conv." + p[-1]                   # output_blocks.i.{1|2}.conv.*
    if p[0] == "out":
        return {"0": "conv_norm_out", "2": "conv_out"}[p[1]] + "." + p[2]
    return None


def _ldm_vae_key(k: str, n_up: int = 4):
    """'first_stage_model.' key (prefix stripped) -> diffusers AutoencoderKL key (decoder side only)."""
    if k.startswith("post_quant_conv."):
        return k
    if not k.startswith("decoder."):
        return None
    p = k.split(".")[1:]
    ren = {"nin_shortcut": "conv_shortcut"}
    if p[0] in ("conv_in", "conv_out"):
        return "decoder." + ".".join(p)
    if p[0] == "norm_out":
        return "decoder.conv_norm_out." + p[1]
    if p[0] == "mid":
        if p[1].startswith("block_"):
            return f"decoder.mid_block.resnets.{int(p[1][-1]) - 1}." + ".".join(ren.get(t, t) for t in p[2:])
        a = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}[p[2]]
        return f"decoder.mid_block.attentions.0.{a}." + p[3]
    if p[0] == "up":
        b = n_up - 1 - int(p[1])
        if p[2] == "block":
            return f"decoder.up_blocks.{b}.resnets.{p[3]}." + ".".join(ren.get(t, t) for t in p[4:])
        return f"decoder.up_blocks.{b}.upsamplers.0.conv." + p[-1]
    return None


def load_single_file(path: str):
    """Original-layout .safetensors checkpoint -> (unet_sd, unet_cfg, vae_sd, vae_cfg, clip_sd | None).
    Architecture numbers are inferred from tensor shapes, then audited against the graph like the directory loader."""
    from safetensors.torch import load_file
    raw = load_file(path)
    usd, vsd, csd = {}, {}, {}
    for k, v in raw.items():
        if k.startswith("model.diffusion_model."):
            kk = k[len("model.diffusion_model."):]
            nk = "time_embedding.cond_proj.weight" if "cond_proj" in kk else _ldm_unet_key(kk)
            if nk:
                usd[nk] = v.to(torch.float16)
        elif k.startswith("first_stage_model."):
            nk = _ldm_vae_key(k[len("first_stage_model."):])
            if nk:
                vsd[nk] = v.to(torch.float16)
        elif k.startswith("cond_stage_model.transformer."):
            kk = k[len("cond_stage_model.transformer."):]
            kk = kk[len("text_model."):] if kk.startswith("text_model.") else kk
            if "position_ids" not in kk:
                csd[kk] = v.to(torch.float16)
    if "conv_in.weight" not in usd:
        raise RuntimeError(f"{path}: no model.diffusion_model.* tensors (not an original-layout SD checkpoint)")
    boc = (usd["conv_in.weight"].shape[0], usd["down_blocks.1.resnets.0.conv1.weight"].shape[0],
           usd["down_blocks.2.resnets.0.conv1.weight"].shape[0], usd["down_blocks.3.resnets.0.conv1.weight"].shape[0])
    ucfg = unet_config(dict(block_out_channels=boc,
                            cross_attention_dim=usd["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape[1],
                            time_cond_proj_dim=(usd["time_embedding.cond_proj.weight"].shape[1]
                                                if "time_embedding.cond_proj.weight" in usd else None)))
    vboc = (vsd["decoder.up_blocks.3.resnets.0.conv1.weight"].shape[0], vsd["decoder.up_blocks.2.resnets.0.conv1.weight"].shape[0],
            vsd["decoder.up_blocks.1.resnets.0.conv1.weight"].shape[0], vsd["decoder.conv_in.weight"].shape[0])
    vcfg = vae_config(dict(block_out_channels=vboc))
    for name, shape, _ in unet_param_spec(ucfg):
        if name not in usd:
            raise RuntimeError(f"checkpoint/graph mismatch at unet '{name}': missing")
        if tuple(usd[name].shape) != tuple(shape):
            usd[name] = usd[name].reshape(shape)            # linear stored as 1x1 conv or vice versa
    for name, shape, _ in vae_param_spec(vcfg):
        if name not in vsd:
            raise RuntimeError(f"checkpoint/graph mismatch at vae '{name}': missing")
        if tuple(vsd[name].shape) != tuple(shape):
            vsd[name] = vsd[name].reshape(shape)
    return usd, ucfg, vsd, vcfg, (csd or None)


def _openclip_text_key(k: str):
    """'conditioner.embedders.1.model.' key (prefix stripped; OpenCLIP text tower as SDXL single files carry it) ->
    list of (transformers CLIPTextModelWithProjection key without 'text_model.', transform) pairs."""
    if k == "token_embedding.weight":
        return [("embeddings.token_embedding.weight", None)]
    if k == "positional_embedding":
        return [("embeddings.position_embedding.weight", None)]
    if k.startswith("ln_final."):
        return [("final_layer_norm." + k.split(".")[1], None)]
    if k == "text_projection":
        return [("text_projection.weight", "T")]                              # x @ P  ->  Linear weight P^T
    if k.startswith("transformer.resblocks."):
        p = k.split(".")
        base = f"encoder.layers.{p[2]}."
        rest = ".".join(p[3:])
        simple = {"ln_1": "layer_norm1", "ln_2": "layer_norm2", "mlp.c_fc": "mlp.fc1", "mlp.c_proj": "mlp.fc2",
                  "attn.out_proj": "self_attn.out_proj"}
        for old, new in simple.items():
            if rest.startswith(old + "."):
                return [(base + new + rest[len(old):], None)]
        if rest in ("attn.in_proj_weight", "attn.in_proj_bias"):
            kind = rest.rsplit("_", 1)[1]
            return [(base + f"self_attn.{n}_proj.{kind}", ("chunk", i)) for i, n in enumerate("qkv")]
    return []                                                                  # logit_scale, attn_mask ...


def load_single_file_sdxl(path: str):
    """Original-layout SDXL .safetensors -> (unet_sd, unet_cfg, vae_sd, vae_cfg, [clip_l_sd, clip_bigg_sd]).
    UNet: model.diffusion_model.* (3 levels, label_emb = text_time embedding); VAE: first_stage_model.*; text encoders:
    conditioner.embedders.0.transformer.text_model.* (CLIP-L, transformers names) and conditioner.embedders.1.model.*
    (OpenCLIP bigG: fused in_proj split into q/k/v, text_projection transposed).  The reference reaches this through
    StableDiffusionXLPipeline.from_single_file (backends/cuda_worker.py:330-352)."""
    from safetensors.torch import load_file
    from .config import SDXL_UNET
    raw = load_file(path)
    if not any(k.startswith("model.diffusion_model.label_emb.") for k in raw):
        raise RuntimeError(f"{path}: not an original-layout SDXL checkpoint (no model.diffusion_model.label_emb.*)")
    usd, vsd, c1, c2 = {}, {}, {}, {}
    down_attn = SDXL_UNET["down_attn"]
    for k, v in raw.items():
        if k.startswith("model.diffusion_model."):
            nk = _ldm_unet_key(k[len("model.diffusion_model."):], 2, down_attn)
            if nk:
                usd[nk] = v.to(torch.float16)
        elif k.startswith("first_stage_model."):
            nk = _ldm_vae_key(k[len("first_stage_model."):])
            if nk:
                vsd[nk] = v.to(torch.float16)
        elif k.startswith("conditioner.embedders.0.transformer."):
            kk = k[len("conditioner.embedders.0.transformer."):]
            kk = kk[len("text_model."):] if kk.startswith("text_model.") else kk
            if "position_ids" not in kk:
                c1[kk] = v.to(torch.float16)
        elif k.startswith("conditioner.embedders.1.model."):
            for nk, tf in _openclip_text_key(k[len("conditioner.embedders.1.model."):]):
                t = v
                if tf == "T":
                    t = v.t()
                elif tf is not None:
                    t = v.chunk(3, dim=0)[tf[1]]
                c2[nk] = t.to(torch.float16).contiguous()
    if "conv_in.weight" not in usd or "add_embedding.linear_1.weight" not in usd:
        raise RuntimeError(f"{path}: not an original-layout SDXL checkpoint (no model.diffusion_model.label_emb.*)")
    boc = (usd["conv_in.weight"].shape[0], usd["down_blocks.1.resnets.0.conv1.weight"].shape[0],
           usd["down_blocks.2.resnets.0.conv1.weight"].shape[0])
    depth = tuple(0 if not down_attn[b] else
                  1 + max(int(k.split(".")[5]) for k in usd if k.startswith(f"down_blocks.{b}.attentions.0.transformer_blocks."))
                  for b in range(3))
    cad = usd["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape[1]
    ucfg = unet_config(dict(SDXL_UNET, block_out_channels=boc, cross_attention_dim=cad,
                            transformer_layers_per_block=tuple(max(1, d) for d in depth),
                            attention_head_dim=tuple(c // 64 for c in boc),
                            projection_class_embeddings_input_dim=usd["add_embedding.linear_1.weight"].shape[1]))
    vboc = (vsd["decoder.up_blocks.3.resnets.0.conv1.weight"].shape[0], vsd["decoder.up_blocks.2.resnets.0.conv1.weight"].shape[0],
            vsd["decoder.up_blocks.1.resnets.0.conv1.weight"].shape[0], vsd["decoder.conv_in.weight"].shape[0])
    vcfg = vae_config(dict(block_out_channels=vboc, scaling_factor=0.13025, sample_size=1024, force_upcast=True))
    for what, sd, spec in (("unet", usd, unet_param_spec(ucfg)), ("vae", vsd, vae_param_spec(vcfg))):
        for name, shape, _ in spec:
            if name not in sd:
                raise RuntimeError(f"checkpoint/graph mismatch at {what} '{name}': missing")
            if tuple(sd[name].shape) != tuple(shape):
                sd[name] = sd[name].reshape(shape)
    return usd, ucfg, vsd, vcfg, [c1 or None, c2 or None]
