"""UNet2DConditionModel (SD1.5 family) restated with plain torch CPU fp32 ops.  Oracle only.

The reference executes this inside diffusers (backends/cuda_worker.py:221-229;
numpy-twin call backends/rknnlcm.py:588-593).  Spec: SURVEY.md Appendix A.4.
Weights are a flat ``{diffusers_state_dict_name: tensor}`` mapping.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, attention_head_dim=8, cross_attention_dim=768,
    norm_num_groups=32, norm_eps=1e-5, time_cond_proj_dim=256,
    down_attn=(True, True, True, False),
)


def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): [cos | sin]."""
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    e = t.float()[:, None] * f[None, :]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)


class UNetOracle:
    def __init__(self, sd: dict, cfg: dict | None = None):
        self.cfg = dict(SD15_UNET if cfg is None else cfg)
        self.sd = {k: v.float() for k, v in sd.items()}
        self.taps = None  # optional dict to record intermediates (tests)

    # -- helpers -------------------------------------------------------------
    def _w(self, name):
        return self.sd[name]

    def _b(self, name):
        return self.sd.get(name)

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x.detach().clone()

    def conv(self, p, x, stride=1, padding=1):
        return F.conv2d(x, self._w(p + ".weight"), self._b(p + ".bias"), stride=stride, padding=padding)

    def lin(self, p, x):
        return F.linear(x, self._w(p + ".weight"), self._b(p + ".bias"))

    def gn(self, p, x, eps):
        return F.group_norm(x, self.cfg["norm_num_groups"], self._w(p + ".weight"), self._w(p + ".bias"), eps)

    def ln(self, p, x):
        return F.layer_norm(x, (x.shape[-1],), self._w(p + ".weight"), self._w(p + ".bias"), 1e-5)

    # -- blocks --------------------------------------------------------------
    def resnet(self, p, x, temb):
        eps = self.cfg["norm_eps"]
        h = self.conv(p + ".conv1", F.silu(self.gn(p + ".norm1", x, eps)))
        h = h + self.lin(p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
        h = self.conv(p + ".conv2", F.silu(self.gn(p + ".norm2", h, eps)))
        if (p + ".conv_shortcut.weight") in self.sd:
            x = self.conv(p + ".conv_shortcut", x, padding=0)
        out = x + h
        self._tap(p, out)
        return out

    def _heads(self, level):
        h = self.cfg["attention_head_dim"]
        return int(h[level]) if isinstance(h, (tuple, list)) else int(h)

    def _depth(self, level):
        t = self.cfg.get("transformer_layers_per_block", 1)
        return int(t[level]) if isinstance(t, (tuple, list)) else int(t)

    def attention(self, p, x, ctx, heads):
        q, k, v = self.lin(p + ".to_q", x), self.lin(p + ".to_k", ctx), self.lin(p + ".to_v", ctx)
        B, S, C = q.shape
        d = C // heads
        q = q.view(B, S, heads, d).transpose(1, 2)
        k = k.view(B, -1, heads, d).transpose(1, 2)
        v = v.view(B, -1, heads, d).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, S, C)
        return self.lin(p + ".to_out.0", a)

    def transformer(self, p, x, ehs, level):
        """Transformer2DModel: 1x1-conv (SD1.5) or Linear (SDXL, use_linear_projection) projections, `depth` blocks."""
        B, C, H, W = x.shape
        heads, depth = self._heads(level), self._depth(level)
        res = x
        h = self.gn(p + ".norm", x, 1e-6)
        linear = self._w(p + ".proj_in.weight").ndim == 2
        if linear:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = self.lin(p + ".proj_in", h)
        else:
            h = self.conv(p + ".proj_in", h, padding=0)
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        for k in range(depth):
            tb = f"{p}.transformer_blocks.{k}"
            n1 = self.ln(tb + ".norm1", h)
            h = self.attention(tb + ".attn1", n1, n1, heads) + h
            h = self.attention(tb + ".attn2", self.ln(tb + ".norm2", h), ehs, heads) + h
            n = self.ln(tb + ".norm3", h)
            g = self.lin(tb + ".ff.net.0.proj", n)
            a, gate = g.chunk(2, dim=-1)
            h = self.lin(tb + ".ff.net.2", a * F.gelu(gate)) + h
        if linear:
            h = self.lin(p + ".proj_out", h)
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
            out = h + res
        else:
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
            out = self.conv(p + ".proj_out", h, padding=0) + res
        self._tap(p, out)
        return out

    def time_embed(self, t, timestep_cond, added=None):
        ch0 = self.cfg["block_out_channels"][0]
        e = timestep_sinusoid(t, ch0)
        if self.cfg.get("time_cond_proj_dim") and timestep_cond is not None:
            e = e + F.linear(timestep_cond.float(), self._w("time_embedding.cond_proj.weight"))
        e = self.lin("time_embedding.linear_1", e)
        e = self.lin("time_embedding.linear_2", F.silu(e))
        if self.cfg.get("addition_time_embed_dim"):
            # SDXL "text_time": concat(pooled text embeds, sinusoid of the 6 size/crop ids) -> MLP, added to temb
            text_embeds, time_ids = added
            B = time_ids.shape[0]
            tid = timestep_sinusoid(time_ids.reshape(-1), self.cfg["addition_time_embed_dim"]).reshape(B, -1)
            a = torch.cat([text_embeds.float(), tid], dim=-1)
            a = self.lin("add_embedding.linear_2", F.silu(self.lin("add_embedding.linear_1", a)))
            e = e + a
        return e

    # -- forward -------------------------------------------------------------
    @torch.inference_mode()
    def forward(self, sample, t, ehs, timestep_cond=None, added=None):
        cfg = self.cfg
        B = sample.shape[0]
        t = torch.as_tensor(t).reshape(-1).expand(B) if torch.as_tensor(t).numel() == 1 else torch.as_tensor(t)
        temb = self.time_embed(t, timestep_cond, added)
        self._tap("temb", temb)
        x = self.conv("conv_in", sample.float())
        self._tap("conv_in", x)
        ehs = ehs.float()
        skips = [x]
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"]):
                x = self.resnet(f"down_blocks.{i}.resnets.{j}", x, temb)
                if cfg["down_attn"][i]:
                    x = self.transformer(f"down_blocks.{i}.attentions.{j}", x, ehs, i)
                skips.append(x)
            if i < nb - 1:
                x = self.conv(f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
                skips.append(x)
        x = self.resnet("mid_block.resnets.0", x, temb)
        x = self.transformer("mid_block.attentions.0", x, ehs, nb - 1)
        x = self.resnet("mid_block.resnets.1", x, temb)
        up_attn = tuple(reversed(cfg["down_attn"]))
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                x = torch.cat([x, skips.pop()], dim=1)
                x = self.resnet(f"up_blocks.{i}.resnets.{j}", x, temb)
                if up_attn[i]:
                    x = self.transformer(f"up_blocks.{i}.attentions.{j}", x, ehs, nb - 1 - i)
            if i < nb - 1:
                # UNet2DConditionModel.forward sets forward_upsample_size when the sample is not a multiple of
                # 2**num_upsamplers and then hands every up block upsample_size = the next skip's spatial size;
                # Upsample2D.forward: interpolate(size=output_size) in that case, interpolate(scale_factor=2) otherwise
                tgt = tuple(skips[-1].shape[2:])
                if tgt == (2 * x.shape[2], 2 * x.shape[3]):
                    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                else:
                    x = F.interpolate(x, size=tgt, mode="nearest")
                x = self.conv(f"up_blocks.{i}.upsamplers.0.conv", x)
                self._tap(f"up_blocks.{i}.upsamplers.0", x)
        x = F.silu(self.gn("conv_norm_out", x, cfg["norm_eps"]))
        return self.conv("conv_out", x)
