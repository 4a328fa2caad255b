"""AutoencoderKL.decode restated with torch CPU fp32 ops.  Oracle only.

Reference call: inside diffusers from backends/cuda_worker.py:221-229; numpy twin
backends/rknnlcm.py:614-618 (``denoised / scaling_factor`` then decoder).
Spec: SURVEY.md Appendix A.5 / A.6.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SD15_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512)


class VAEDecoderOracle:
    def __init__(self, sd: dict, cfg: dict | None = None):
        self.cfg = dict(SD15_VAE if cfg is None else cfg)
        self.sd = {k: v.float() for k, v in sd.items()}
        self.taps = None

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x.detach().clone()

    def conv(self, p, x, padding=1):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd.get(p + ".bias"), padding=padding)

    def gn(self, p, x):
        return F.group_norm(x, self.cfg["norm_num_groups"], self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-6)

    def lin(self, p, x):
        return F.linear(x, self.sd[p + ".weight"], self.sd.get(p + ".bias"))

    def resnet(self, p, x):
        h = self.conv(p + ".conv1", F.silu(self.gn(p + ".norm1", x)))
        h = self.conv(p + ".conv2", F.silu(self.gn(p + ".norm2", h)))
        if (p + ".conv_shortcut.weight") in self.sd:
            x = self.conv(p + ".conv_shortcut", x, padding=0)
        out = x + h
        self._tap(p, out)
        return out

    def attn(self, p, x):
        B, C, H, W = x.shape
        h = self.gn(p + ".group_norm", x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        q, k, v = self.lin(p + ".to_q", h), self.lin(p + ".to_k", h), self.lin(p + ".to_v", h)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (C ** -0.5), dim=-1) @ v
        a = self.lin(p + ".to_out.0", a)
        out = a.reshape(B, H, W, C).permute(0, 3, 1, 2) + x
        self._tap(p, out)
        return out

    @torch.inference_mode()
    def decode(self, latents, use_tiling=True):
        """AutoencoderKL.decode with ``enable_tiling()`` as the reference sets it (backends/cuda_worker.py:91): the plain
        path unless a latent side exceeds tile_latent_min_size = sample_size / 8, then overlapping tiles."""
        tmin = int(self.cfg.get("sample_size", 512)) // 8
        if use_tiling and (latents.shape[-1] > tmin or latents.shape[-2] > tmin):
            return self.tiled_decode(latents)
        return self.decode_plain(latents)

    @staticmethod
    def _blend_v(a, b, extent):
        extent = min(a.shape[2], b.shape[2], extent)
        for y in range(extent):
            b[:, :, y, :] = a[:, :, -extent + y, :] * (1 - y / extent) + b[:, :, y, :] * (y / extent)
        return b

    @staticmethod
    def _blend_h(a, b, extent):
        extent = min(a.shape[3], b.shape[3], extent)
        for x in range(extent):
            b[:, :, :, x] = a[:, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, x] * (x / extent)
        return b

    @torch.inference_mode()
    def tiled_decode(self, latents, overlap=0.25):
        """diffusers AutoencoderKL.tiled_decode restated: tiles of tile_latent_min_size every 75 % of it, decoded
        independently, linearly blended over 25 % of the tile, cropped and concatenated."""
        sample = int(self.cfg.get("sample_size", 512))
        tl = sample // 8
        stride = int(tl * (1 - overlap))
        extent = int(sample * overlap)
        limit = sample - extent
        H, W = latents.shape[2:]
        rows = []
        for i in range(0, H, stride):
            rows.append([self.decode_plain(latents[:, :, i:i + tl, j:j + tl]).clone() for j in range(0, W, stride)])
        out_rows = []
        for i, row in enumerate(rows):
            res = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend_v(rows[i - 1][j], tile, extent)
                if j > 0:
                    tile = self._blend_h(row[j - 1], tile, extent)
                res.append(tile[:, :, :limit, :limit])
            out_rows.append(torch.cat(res, dim=3))
        return torch.cat(out_rows, dim=2)

    @torch.inference_mode()
    def decode_plain(self, latents):
        """latents: UNet-space [B,4,h,w] -> image [B,3,8h,8w] in ~[-1,1] (plain, untiled path)."""
        cfg = self.cfg
        z = latents.float() / cfg["scaling_factor"]
        z = self.conv("post_quant_conv", z, padding=0)
        x = self.conv("decoder.conv_in", z)
        self._tap("decoder.conv_in", x)
        x = self.resnet("decoder.mid_block.resnets.0", x)
        x = self.attn("decoder.mid_block.attentions.0", x)
        x = self.resnet("decoder.mid_block.resnets.1", x)
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i < nb - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
                self._tap(f"decoder.up_blocks.{i}.upsamplers.0", x)
        x = F.silu(self.gn("decoder.conv_norm_out", x))
        return self.conv("decoder.conv_out", x)
