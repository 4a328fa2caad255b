"""LoRA style adapters for the HIP UNet.

The reference toggles adapters around every job (``pipe.set_adapters([name],[weight])`` / ``disable_lora()``,
backends/cuda_worker.py:165-196).  Here a style is a set of per-weight deltas (B A) * alpha / rank, pre-packed into the
kernel layouts (fused q|k|v rows, the stacked cross-attention K/V matrix, GEGLU interleave); switching a style re-merges
W' = W + weight * delta IN PLACE with the ``lcm_axpy_f16`` kernel, so captured hipGraphs keep their weight pointers.

Supported targets: every Linear / 1x1-conv of the transformer blocks (proj_in/out, attn1/attn2 to_q/k/v/out, ff);
key styles: kohya (``lora_unet_<path>.lora_down.weight`` / ``.lora_up.weight`` / ``.alpha``) and peft / diffusers
(``unet.<path>.lora_A.weight`` / ``.lora_B.weight``).  Text-encoder entries (``lora_te_`` / ``lora_te1_`` / ``lora_te2_`` /
``text_encoder[_2].``: the q/k/v/out projections and the MLP of every CLIP layer) are merged into the native CLIP
encoders the same way (``ClipLora``), as ``pipe.load_lora_weights`` does for the reference.  LoCon entries on the
ResnetBlock2D convolutions (conv1 / conv2 3x3: down [r,Cin,3,3] x up [Cout,r,1,1]; conv_shortcut 1x1), the resnets'
time_emb_proj, and the down/upsampler convolutions are merged too, each delta packed the way its weight is (tap-major
3x3 rows, the four-phase upsampler layout, the stacked time-embedding matrix), and so are entries on the UNet's conv_in /
conv_out (tap-major rows, the layouts lcm_conv3x3_c4_f32in / lcm_conv3x3_smalln read).
"""
from __future__ import annotations

import re

import torch

from . import ops
from .packing import pack_conv1x1, pack_conv3x3, pack_conv3x3_up2, pack_ff2_cols, pack_geglu
from .weights import unet_param_spec


_CONV_TARGET = re.compile(r"\.resnets\.\d+\.(conv1|conv2|conv_shortcut|time_emb_proj)$|\.(downsamplers|upsamplers)\.0\.conv$|^conv_in$|^conv_out$")


def _target_modules(cfg):
    out = {}
    for name, shape, kind in unet_param_spec(cfg):
        if not name.endswith(".weight"):
            continue
        mod = name[:-7]
        if (".attentions." in name and kind in ("w", "w_res")) or _CONV_TARGET.search(mod):
            out[mod] = tuple(shape)
    return out


def parse_lora(raw: dict, cfg) -> tuple[dict, list]:
    """raw safetensors dict -> ({module_path: (down[r,in], up[out,r], alpha)}, skipped keys)."""
    mods = _target_modules(cfg)
    by_kohya = {"lora_unet_" + m.replace(".", "_"): m for m in mods}
    found, skipped = {}, []
    alphas = {k[:-6]: float(v) for k, v in raw.items() if k.endswith(".alpha")}
    for k, v in raw.items():
        if k.endswith(".alpha"):
            continue
        m = re.match(r"^(.*)\.(lora_down|lora_up|lora_A|lora_B)\.weight$", k)
        if not m:
            skipped.append(k)
            continue
        stem, part = m.group(1), m.group(2)
        mod = by_kohya.get(stem)
        if mod is None:
            s2 = stem[5:] if stem.startswith("unet.") else stem
            s2 = s2.replace(".processor", "")
            mod = s2 if s2 in mods else None
        if mod is None:
            skipped.append(k)
            continue
        e = found.setdefault(mod, {"alpha": alphas.get(stem)})
        # LoCon keeps the 3x3 extent on lora_down ([r, Cin, 3, 3]); lora_up is [Cout, r(,1,1)]
        e["down" if part in ("lora_down", "lora_A") else "up"] = v.float() if (v.dim() == 4 and v.shape[-1] > 1) else v.float().reshape(v.shape[0], -1)
    out = {}
    for mod, e in found.items():
        if "down" in e and "up" in e:
            r = e["down"].shape[0]
            out[mod] = (e["down"], e["up"], e["alpha"] if e["alpha"] is not None else float(r))
    return out, skipped


class LoraStyle:
    """Deltas of one LoRA, packed for a given UNetHip; ``apply(weight)`` re-merges them in place."""

    def __init__(self, unet, raw: dict):
        parsed, self.skipped = parse_lora(raw, unet.cfg)
        self.unet = unet
        self.modules = sorted(parsed)
        deltas = {}

        cdeltas = {}

        def acc(name, rows, d, dc=None):
            w = unet.w[name]
            t = deltas.get(name)
            if t is None:
                t = torch.zeros(w.shape, dtype=torch.float32)
                deltas[name] = t
            t[rows[0]:rows[1]] += d
            if dc is not None:           # LayerNorm-folded consumer: the delta of c = W beta + b that goes with it
                c = cdeltas.setdefault(name, torch.zeros(w.shape[0], dtype=torch.float32))
                c[rows[0]:rows[1]] += dc

        def fold(name, d):
            """A delta of a weight that carries a folded LayerNorm (model.UNetHip._put_ln_fold): columns scaled by gamma,
            plus the matching delta of the constant c = W beta + b.  -> (delta of gamma (*) W, delta of c) or (d, None)."""
            stem = name[:-2]
            if (stem + ".lnw") not in unet.w:
                return d, None
            return d * unet.w[stem + ".lnw"].float().cpu()[None, :], d @ unet.w[stem + ".lnb"].float().cpu()

        for mod, (down, up, alpha) in parsed.items():
            if _CONV_TARGET.search(mod):
                self._conv_delta(unet, mod, down, up, alpha, acc)
                continue
            d = (up @ down) * (alpha / down.shape[0])                        # [out, in]
            m = re.match(r"^(.*\.attentions\.\d+)\.(.*)$", mod)
            p, rest = m.group(1), m.group(2)
            if rest in ("proj_in", "proj_out"):
                acc(f"{p}.{rest}.w", (0, d.shape[0]), d)
                continue
            k, sub = re.match(r"^transformer_blocks\.(\d+)\.(.*)$", rest).groups()
            q, C = f"{p}.{k}", d.shape[0]
            qs = getattr(unet, "qs", {}).get(q, 1.0)                         # to_q rows carry the softmax scale (model.Q_PRESCALE)
            if sub.startswith("attn1.to_") and sub[-1] in "qkv":
                i = "qkv".index(sub[-1])
                acc(q + ".qkv.w", (i * C, (i + 1) * C), *fold(q + ".qkv.w", d * qs if i == 0 else d))
            elif sub == "attn1.to_out.0":
                acc(q + ".o1.w", (0, C), d)
            elif sub == "attn2.to_q":
                acc(q + ".q2.w", (0, C), *fold(q + ".q2.w", d * qs))
            elif sub in ("attn2.to_k", "attn2.to_v"):
                off, Ck = unet.kv_off[q]
                o = off + (Ck if sub.endswith("v") else 0)
                acc("kv_all.w", (o, o + Ck), d)
            elif sub == "attn2.to_out.0":
                acc(q + ".o2.w", (0, C), d)
            elif sub == "ff.net.0.proj":
                dg, dc = fold(q + ".ff1.w", d)
                dp, dcp = pack_geglu(dg, dc)
                acc(q + ".ff1.w", (0, dp.shape[0]), dp, dcp)
            elif sub == "ff.net.2":
                acc(q + ".ff2.w", (0, C), pack_ff2_cols(d))          # columns in the GEGLU output's stored order
        dev = unet.device
        self.delta = {n: t.to(torch.float16).to(dev).contiguous() for n, t in deltas.items()}
        self.base = {n: unet.w[n].clone() for n in self.delta}
        # LayerNorm-folded consumers touched by this style: their constants follow the live weight (ops.ln_fold_refresh)
        self.cdelta = {n: t.to(dev).contiguous() for n, t in cdeltas.items()}
        self.cbase = {n: unet.w[n[:-2] + ".c"].clone() for n in self.cdelta}
        self.current = 0.0

    @staticmethod
    def _conv_delta(unet, mod, down, up, alpha, acc):
        """LoCon / linear entries outside the transformer blocks, packed like the weight they modify (model.UNetHip)."""
        from .model import UPS_PHASES
        r = down.shape[0]
        scale = alpha / r
        if mod.endswith(".time_emb_proj"):
            off, n = unet.temb_off[mod[:-len(".time_emb_proj")]]
            acc("temb_all.w", (off, off + n), (up @ down.reshape(r, -1)) * scale)
            return
        if down.dim() == 4:                       # 3x3 LoCon: delta[o,i,ky,kx] = sum_r up[o,r] down[r,i,ky,kx]
            d4 = torch.einsum("or,rikl->oikl", up.reshape(up.shape[0], r), down) * scale
        else:                                      # 1x1 (conv_shortcut) or a 3x3 target given as a plain matrix
            d2 = (up.reshape(up.shape[0], r) @ down) * scale
            d4 = d2.reshape(d2.shape[0], -1, 1, 1)
        if mod.endswith(".conv_shortcut"):
            acc(mod.replace(".conv_shortcut", ".sc") + ".w", (0, d4.shape[0]), pack_conv1x1(d4))
            return
        if d4.shape[-1] != 3:
            raise ValueError(f"LoRA entry for {mod}: expected a 3x3 LoCon pair, got down {tuple(down.shape)}")
        name = mod + ".w"
        packed = pack_conv3x3_up2(d4) if (".upsamplers." in mod and UPS_PHASES) else pack_conv3x3(d4)
        acc(name, (0, packed.shape[0]), packed)
        if (mod + ".w3") in unet.w:               # the upsampler's plain 3x3 copy (odd-sized targets) follows the style too
            p3 = pack_conv3x3(d4)
            acc(mod + ".w3", (0, p3.shape[0]), p3)

    def nbytes(self):
        return 2 * sum(t.numel() * 2 for t in self.delta.values()) + 2 * sum(t.numel() * 4 for t in self.cdelta.values())

    def apply(self, weight: float):
        weight = float(weight)
        if weight == self.current:
            return
        for n, d in self.delta.items():
            ops.axpy(self.base[n], d, weight, self.unet.w[n])
            if n in self.cdelta:
                stem = n[:-2]
                ops.ln_fold_refresh(self.unet.w[n], self.unet.w[stem + ".g"], self.cbase[n], self.cdelta[n], weight,
                                    self.unet.w[stem + ".c"])
        self.current = weight


# ---- text-encoder LoRA ---------------------------------------------------------------------------
_TE_SUB = {"self_attn.q_proj": ("qkv", 0), "self_attn.k_proj": ("qkv", 1), "self_attn.v_proj": ("qkv", 2),
           "self_attn.out_proj": ("o", None), "mlp.fc1": ("fc1", None), "mlp.fc2": ("fc2", None)}


def parse_te_lora(raw: dict, n_layers: int, index: int) -> tuple[dict, list]:
    """Entries of ``raw`` that target text encoder ``index`` (0: CLIP-L / the only one, 1: SDXL's second) ->
    ({(layer, sub-module): (down, up, alpha)}, matched keys)."""
    kohya = ["lora_te_", "lora_te1_"] if index == 0 else ["lora_te2_"]
    peft = ["text_encoder."] if index == 0 else ["text_encoder_2."]
    names = {}
    for i in range(n_layers):
        for sub in _TE_SUB:
            path = f"text_model.encoder.layers.{i}.{sub}"
            for pre in kohya:
                names[pre + path.replace(".", "_")] = (i, sub)
            for pre in peft:
                names[pre + path] = (i, sub)
    alphas = {k[:-6]: float(v) for k, v in raw.items() if k.endswith(".alpha")}
    found, used = {}, []
    for k, v in raw.items():
        m = re.match(r"^(.*?)\.(lora_down|lora_up|lora_A|lora_B|lora_linear_layer\.down|lora_linear_layer\.up)\.weight$", k)
        if not m or m.group(1) not in names:
            continue
        used.append(k)
        e = found.setdefault(names[m.group(1)], {"alpha": alphas.get(m.group(1))})
        e["down" if m.group(2) in ("lora_down", "lora_A", "lora_linear_layer.down") else "up"] = v.float().reshape(v.shape[0], -1)
    out = {key: (e["down"], e["up"], e["alpha"] if e["alpha"] is not None else float(e["down"].shape[0]))
           for key, e in found.items() if "down" in e and "up" in e}
    return out, used


class ClipLora:
    """Deltas of one LoRA for one native CLIP text encoder (clip.ClipTextHip); same in-place axpy merge as LoraStyle."""

    def __init__(self, enc, raw: dict, index: int = 0):
        parsed, self.used = parse_te_lora(raw, enc.L, index)
        self.enc = enc
        self.modules = sorted(parsed)
        deltas = {}
        for (layer, sub), (down, up, alpha) in parsed.items():
            d = (up @ down) * (alpha / down.shape[0])
            tgt, part = _TE_SUB[sub]
            name = f"{layer}.{tgt}.w"
            w = enc.w[name]
            t = deltas.setdefault(name, torch.zeros(w.shape, dtype=torch.float32))
            if part is None:
                t += d
            else:
                t[part * enc.D:(part + 1) * enc.D] += d
        self.delta = {n: t.to(torch.float16).to(enc.device).contiguous() for n, t in deltas.items()}
        self.base = {n: enc.w[n].clone() for n in self.delta}
        self.current = 0.0

    def nbytes(self):
        return 2 * sum(t.numel() * 2 for t in self.delta.values())

    def apply(self, weight: float):
        weight = float(weight)
        if weight == self.current:
            return
        for n, d in self.delta.items():
            ops.axpy(self.base[n], d, weight, self.enc.w[n])
        self.current = weight


class StyleAdapters:
    """Everything one style file touches: the UNet deltas plus one ClipLora per text encoder that has entries."""

    def __init__(self, unet, encoders, raw: dict):
        self.unet = LoraStyle(unet, raw)
        self.text = [c for c in (ClipLora(e, raw, i) for i, e in enumerate(encoders)) if c.modules]
        te_keys = {k for c in self.text for k in c.used}
        self.modules = self.unet.modules
        self.skipped = [k for k in self.unet.skipped if k not in te_keys]
        self.text_modules = sum(len(c.modules) for c in self.text)

    def nbytes(self):
        return self.unet.nbytes() + sum(c.nbytes() for c in self.text)

    def apply(self, weight: float):
        self.unet.apply(weight)
        for c in self.text:
            c.apply(weight)
