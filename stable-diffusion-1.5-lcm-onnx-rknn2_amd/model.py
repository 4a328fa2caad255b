"""Host-side executors of the two networks on the hot path: they walk the UNet2DConditionModel /
AutoencoderKL-decoder op graph (diffusers modules reached from
backends/cuda_worker.py:221-229) and enqueue the HIP kernels of liblcmhip.so through ``ops``.

No arithmetic happens here: Python only owns shapes, buffers and launch order; the whole sequence is
captured once into a hipGraph by ``pipeline.LcmHipPipeline`` and replayed per request.

Activations are pixel-major fp16 ``[B*H*W, C]``.  Buffers are allocated on first use and keyed by
(role, shape), so every layer of a resolution level reuses the same few scratch tensors (they stay
resident in the 256 MiB Infinity Cache); only skip connections get buffers of their own.
"""
from __future__ import annotations

import torch

from . import ops
from .config import TEXT_SEQ_LEN, depth_at, heads_at, unet_config, vae_config
from .packing import pack_conv1x1, pack_conv3x3, pack_conv3x3_up2, pack_ff2_cols, pack_geglu


import os

FUSED_GN_STATS = os.environ.get("LCM_FUSED_GN_STATS", "1") != "0"
# Upsample2D (nearest-2x -> conv3x3) as four 2x2 phase convolutions on the low-resolution input (2.25x fewer MACs)
UPS_PHASES = os.environ.get("LCM_UPS_PHASES", "1") != "0"
# GroupNorm-apply(+SiLU) inside the consuming conv's halo staging instead of its own pass over HBM -- where it pays:
# the staged element is transformed once per n-tile of the conv, so few n-tiles (<= 4 with the 160-wide tile, <= 3 with
# 128) and a tensor too large for the Infinity Cache, so that the separate pass really is HBM time (in situ the apply pass
# reads what the producer just wrote: below ~64 MB it is served from MALL and fusing gains nothing; measured +1.5 % at
# batch 8 and +0.2 % at batch 1 with the VAE's 512^2 / 256^2 levels fused; tools/gn_fuse_ab.py re-measures it per shape: profiles/r04_gn_fuse_ab.txt)
# LayerNorm folded into the GEMM that consumes it (norm1 -> q|k|v, norm2 -> attn2.to_q, norm3 -> GEGLU proj): 192 launches
# fewer per 512x512 4-step pass, no LayerNorm output in HBM (ops.gemm_ln / lcm_gemm_ln_f16)
LN_FOLD = os.environ.get("LCM_LN_FOLD", "1") != "0"
# conv_shortcut of a ResnetBlock2D on a forked side stream, concurrent with norm1 -> conv1 -> norm2 (see _Net.resnet).
# OFF by default: measured on the same box, 512x512 4-step batch 1: 45.5 images/s with the 56 fork / join pairs per pass in
# the captured graph against 47.6 without -- a cross-stream edge costs more here than the ~10 us GEMM it takes off the chain.
FORK_SHORTCUT = os.environ.get("LCM_FORK_SHORTCUT", "0") != "0"
FUSE_GN_CONV = os.environ.get("LCM_FUSE_GN_CONV", "1") != "0"
# softmax scale (and log2 e) folded into the to_q weights of the UNet's attention layers (see _pack_transformer)
Q_PRESCALE = os.environ.get("LCM_Q_PRESCALE", "1") != "0"
# AutoencoderKL mid-block attention (one head, d = 512) as ONE fused flash kernel behind one q|k|v GEMM; "0" = the round-1
# GEMM -> softmax -> transpose -> GEMM form with its B x S x S score matrix in HBM (kept for comparison)
VAE_FLASH_ATTN = os.environ.get("LCM_VAE_FLASH_ATTN", "1") != "0"
FUSE_GN_MIN_BYTES = int(os.environ.get("LCM_FUSE_GN_MIN_BYTES", str(64 << 20)))


def _fuse_gn_into_conv(M, Cin, Cout):
    if not (FUSE_GN_CONV and FUSED_GN_STATS) or Cin % 64 or Cout % 64:
        return False
    ntiles = Cout // 160 if Cout % 160 == 0 else -(-Cout // 128)
    return M * Cin * 2 >= FUSE_GN_MIN_BYTES and ntiles <= (4 if Cout % 160 == 0 else 3)


class _Buffers:
    def __init__(self, device):
        self.device = device
        self._b = {}

    def get(self, role, *shape, dtype=torch.float16, zero=False):
        """zero: zero-filled at allocation (buffers whose padding the kernels never write)."""
        key = (role, shape, dtype)
        t = self._b.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._b[key] = t
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._b.values())


def _dev(t, device, dtype=torch.float16):
    return t.to(device=device, dtype=dtype).contiguous()


class _Net:
    def __init__(self, device):
        self.device = device
        self.w = {}
        self.buf = _Buffers(device)
        self._stats = {}
        self.side_stream = None       # set by the pipeline lane: a second stream for launches that fork off the main chain

    def _put(self, name, t, dtype=torch.float16):
        self.w[name] = _dev(t, self.device, dtype)

    def view(self):
        """The same network (shared weight tensors: a style re-merge reaches every view) with scratch of its own -- one per
        pipeline lane, so two captured passes can be in flight at once without sharing an activation buffer."""
        v = object.__new__(type(self))
        v.__dict__.update(self.__dict__)
        v.buf = _Buffers(self.device)
        v._stats = {}
        return v

    def stats(self, role, M, C, hw):
        """Per-role holder of the producer-written GroupNorm statistics of an [M, C] tensor of M // hw images."""
        key = ("stats", role, M, C, hw)
        st = self._stats.get(key)
        if st is None:
            st = ops.Stats(torch.zeros(ops.stats_floats(M, C, hw), dtype=torch.float32, device=self.device))
            self._stats[key] = st
        return st

    def norm(self, x, gamma, beta, out, B, HW, C1, *, x_st=None, x2=None, C2=0, x2_st=None, eps=1e-5, silu=True):
        """GroupNorm(+SiLU) of [x | x2]: from the producers' fused statistics when every source has them
        (finalize + apply, no statistics pass over the data), else the standalone three-kernel form."""
        C = C1 + C2
        ws = self.gn_ws(B, HW, C)
        if FUSED_GN_STATS and x_st is not None and x_st.P > 0 and (x2 is None or (x2_st is not None and x2_st.P > 0)):
            ops.groupnorm_from_stats(x, gamma, beta, out, B, HW, C1, x_st, ws, x2=x2, C2=C2, st2=x2_st, eps=eps, silu=silu)
        else:
            ops.groupnorm(x, gamma, beta, out, B, HW, C1, ws, x2=x2, C2=C2, eps=eps, silu=silu)
        return out

    def _norm_conv_out(self, x, norm, st, hn, out, B, H, W, ch, cout, eps, *, mode, out_f32=None):
        """conv_norm_out -> SiLU -> conv_out.  With the producer's fused statistics and 64-aligned channels the GroupNorm is
        applied inside the conv's staging pass (the normalised tensor never exists in memory); the choice depends on the layer's
        shape only, and both forms feed the conv the same fp16 values."""
        wt = self.w
        if FUSED_GN_STATS and st is not None and st.P > 0 and ch % 64 == 0:
            sc_t, sh_t = ops.groupnorm_tables_from_stats(wt[norm + ".g"], wt[norm + ".b"], B, H * W, ch, st, self.gn_ws(B, H * W, ch), eps=eps)
            ops.conv3x3_smalln(x, wt["conv_out.w"], out, B, H, W, ch, cout, bias=wt["conv_out.b"], mode=mode, out_f32=out_f32,
                               gn_scale=sc_t, gn_shift=sh_t, silu=True)
        else:
            self.norm(x, wt[norm + ".g"], wt[norm + ".b"], hn, B, H * W, ch, x_st=st, eps=eps)
            ops.conv3x3_smalln(hn, wt["conv_out.w"], out, B, H, W, ch, cout, bias=wt["conv_out.b"], mode=mode, out_f32=out_f32)

    def gn_ws(self, B, HW, C):
        """fp32 scratch of the GroupNorm kernels (chunk partials + [B][C] scale / shift tables).  One tensor per
        (B, HW, C), allocated once and never replaced: captured hipGraphs bake the raw pointer in, so growing a shared
        buffer in place would leave earlier plans' graphs writing into freed memory."""
        key = ("gn_ws", B, HW, C)
        t = self._stats.get(key)
        if t is None:
            t = torch.empty(max(ops.groupnorm_ws_bytes(B, HW, C) // 4, 1024), dtype=torch.float32, device=self.device)
            self._stats[key] = t
        return t

    def weight_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.w.values())

    # ---- shared blocks ----------------------------------------------------------------------
    def _pack_resnet(self, sd, p, temb_list=None):
        for n in ("norm1", "norm2"):
            self._put(f"{p}.{n}.g", sd[f"{p}.{n}.weight"])
            self._put(f"{p}.{n}.b", sd[f"{p}.{n}.bias"])
        for n in ("conv1", "conv2"):
            self._put(f"{p}.{n}.w", pack_conv3x3(sd[f"{p}.{n}.weight"]))
            self._put(f"{p}.{n}.b", sd[f"{p}.{n}.bias"])
        if f"{p}.conv_shortcut.weight" in sd:
            self._put(f"{p}.sc.w", pack_conv1x1(sd[f"{p}.conv_shortcut.weight"]))
            self._put(f"{p}.sc.b", sd[f"{p}.conv_shortcut.bias"])
        if temb_list is not None:
            temb_list.append((p, sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"]))

    def resnet(self, p, x, C1, Cout, B, H, W, eps, x2=None, C2=0, rowadd=None, out_role="res_out", x_st=None, x2_st=None):
        """-> (out, out_stats).  x_st / x2_st: fused statistics of the inputs (None: compute them standalone)."""
        HW, M, Cin = H * W, B * H * W, C1 + C2
        w = self.w
        have1 = x_st is not None and x_st.P > 0 and (x2 is None or (x2_st is not None and x2_st.P > 0))
        # conv_shortcut (1x1 over the raw input) does not depend on norm1 -> conv1 -> norm2: with LCM_FORK_SHORTCUT=1 it forks
        # onto the lane's side stream and joins before conv2 adds it as the residual (edges of the captured graph).  Measured
        # slower than the serial chain (see FORK_SHORTCUT), so off by default.
        sc, join = x, None
        if (p + ".sc.w") in w:
            sc = self.buf.get("shortcut", M, Cout)
            side = self.side_stream if FORK_SHORTCUT else None
            if side is not None:
                fork = torch.cuda.Event()
                fork.record()
                with torch.cuda.stream(side):
                    side.wait_event(fork)
                    ops.gemm(x, w[p + ".sc.w"], sc, bias=w[p + ".sc.b"], a2=x2, img_rows=HW)
                    join = torch.cuda.Event()
                    join.record()
            else:
                ops.gemm(x, w[p + ".sc.w"], sc, bias=w[p + ".sc.b"], a2=x2, img_rows=HW)
        h1 = self.buf.get("conv1", M, Cout)
        h1_st = self.stats("conv1", M, Cout, HW)
        if have1 and C1 % 64 == 0 and C2 % 64 == 0 and _fuse_gn_into_conv(M, Cin, Cout):
            sc_t, sh_t = ops.groupnorm_tables_from_stats(w[p + ".norm1.g"], w[p + ".norm1.b"], B, HW, C1, x_st, self.gn_ws(B, HW, Cin),
                                                         C2=C2, st2=x2_st if x2 is not None else None, eps=eps)
            ops.conv3x3_gn(x, w[p + ".conv1.w"], h1, B, H, W, C1, Cout, x2=x2, C2=C2, gn_scale=sc_t, gn_shift=sh_t, silu=True,
                           bias=w[p + ".conv1.b"], rowadd=rowadd, stats=h1_st)
        else:
            hn = self.buf.get("gn", M, Cin)
            self.norm(x, w[p + ".norm1.g"], w[p + ".norm1.b"], hn, B, HW, C1, x_st=x_st, x2=x2, C2=C2, x2_st=x2_st, eps=eps)
            ops.conv3x3(hn, w[p + ".conv1.w"], h1, B, H, W, Cin, Cout, bias=w[p + ".conv1.b"], rowadd=rowadd, stats=h1_st)
        out = self.buf.get(out_role, M, Cout)
        out_st = self.stats(out_role, M, Cout, HW)
        if h1_st.P > 0 and _fuse_gn_into_conv(M, Cout, Cout):
            sc_t, sh_t = ops.groupnorm_tables_from_stats(w[p + ".norm2.g"], w[p + ".norm2.b"], B, HW, Cout, h1_st,
                                                         self.gn_ws(B, HW, Cout), eps=eps)
            if join is not None:
                torch.cuda.current_stream().wait_event(join)
            ops.conv3x3_gn(h1, w[p + ".conv2.w"], out, B, H, W, Cout, Cout, gn_scale=sc_t, gn_shift=sh_t, silu=True,
                           bias=w[p + ".conv2.b"], res=sc, stats=out_st)
        else:
            hn2 = self.buf.get("gn", M, Cout)
            self.norm(h1, w[p + ".norm2.g"], w[p + ".norm2.b"], hn2, B, HW, Cout, x_st=h1_st, eps=eps)
            if join is not None:
                torch.cuda.current_stream().wait_event(join)
            ops.conv3x3(hn2, w[p + ".conv2.w"], out, B, H, W, Cout, Cout, bias=w[p + ".conv2.b"], res=sc, stats=out_st)
        return out, out_st


# ==================================================================================================
class UNetHip(_Net):
    def __init__(self, sd: dict, cfg: dict | None = None, device="cuda"):
        super().__init__(device)
        self.cfg = cfg = unet_config(cfg)
        boc = cfg["block_out_channels"]
        self.ctx_dim = cfg["cross_attention_dim"]
        self.temb_dim = boc[0] * 4
        nb = len(boc)
        temb_list, kv_list = [], []
        self.qs = {}                                                     # per transformer block: the factor carried by its to_q rows
        self._put("conv_in.w", pack_conv3x3(sd["conv_in.weight"]))
        self._put("conv_in.b", sd["conv_in.bias"])
        for n in ("linear_1", "linear_2"):
            self._put(f"te.{n}.w", sd[f"time_embedding.{n}.weight"])
            self._put(f"te.{n}.b", sd[f"time_embedding.{n}.bias"])
        self.has_cond = bool(cfg.get("time_cond_proj_dim"))
        if self.has_cond:
            self._put("te.cond.w", sd["time_embedding.cond_proj.weight"])
        self.has_added = bool(cfg.get("addition_time_embed_dim"))       # SDXL "text_time" embedding
        if self.has_added:
            for n in ("linear_1", "linear_2"):
                self._put(f"add.{n}.w", sd[f"add_embedding.{n}.weight"])
                self._put(f"add.{n}.b", sd[f"add_embedding.{n}.bias"])
            self.added_dim = cfg["projection_class_embeddings_input_dim"]
        for i in range(nb):
            for j in range(cfg["layers_per_block"]):
                self._pack_resnet(sd, f"down_blocks.{i}.resnets.{j}", temb_list)
                if cfg["down_attn"][i]:
                    self._pack_transformer(sd, f"down_blocks.{i}.attentions.{j}", kv_list, depth_at(cfg, i), heads_at(cfg, i))
            if i < nb - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                self._put(p + ".w", pack_conv3x3(sd[p + ".weight"]))
                self._put(p + ".b", sd[p + ".bias"])
        self._pack_resnet(sd, "mid_block.resnets.0", temb_list)
        self._pack_transformer(sd, "mid_block.attentions.0", kv_list, depth_at(cfg, nb - 1), heads_at(cfg, nb - 1))
        self._pack_resnet(sd, "mid_block.resnets.1", temb_list)
        up_attn = tuple(reversed(cfg["down_attn"]))
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                self._pack_resnet(sd, f"up_blocks.{i}.resnets.{j}", temb_list)
                if up_attn[i]:
                    self._pack_transformer(sd, f"up_blocks.{i}.attentions.{j}", kv_list, depth_at(cfg, nb - 1 - i), heads_at(cfg, nb - 1 - i))
            if i < nb - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                self._put(p + ".w", (pack_conv3x3_up2 if UPS_PHASES else pack_conv3x3)(sd[p + ".weight"]))
                if UPS_PHASES:       # plain 3x3 layout too: upsampling to an odd-sized skip cannot use the pre-summed phase weights
                    self._put(p + ".w3", pack_conv3x3(sd[p + ".weight"]))
                self._put(p + ".b", sd[p + ".bias"])
        self._put("conv_norm_out.g", sd["conv_norm_out.weight"])
        self._put("conv_norm_out.b", sd["conv_norm_out.bias"])
        self._put("conv_out.w", pack_conv3x3(sd["conv_out.weight"]))
        self._put("conv_out.b", sd["conv_out.bias"])
        # all ResnetBlock2D.time_emb_proj stacked into one [sum(Cout), temb] matrix -> one launch per step
        self.temb_off, off = {}, 0
        for p, wt, bt in temb_list:
            self.temb_off[p] = (off, wt.shape[0])
            off += wt.shape[0]
        self.temb_total = off
        self._put("temb_all.w", torch.cat([t[1] for t in temb_list], 0))
        self._put("temb_all.b", torch.cat([t[2] for t in temb_list], 0))
        # all cross-attention to_k|to_v stacked into one [sum(2C), ctx] matrix -> one GEMM per request
        self.kv_off, off = {}, 0
        for p, wk, wv in kv_list:
            self.kv_off[p] = (off, wk.shape[0])
            off += 2 * wk.shape[0]
        self.kv_total = off
        self._put("kv_all.w", torch.cat([torch.cat([wk, wv], 0) for _, wk, wv in kv_list], 0))

    def _put_ln_fold(self, name, W, b, gamma, beta, geglu=False):
        """Weights of LayerNorm(gamma, beta) -> Linear(W, b) as one contraction (ops.gemm_ln):
        ``name.w`` = gamma (*) W (fp16, GEGLU rows interleaved), ``name.g`` = its fp32 row sums, ``name.c`` = W beta + b (fp32).
        The fp32 per-column scale ``name.lnw`` / shift ``name.lnb`` stay around for the style-LoRA re-merge (lora.py)."""
        Wf, g32, b32 = W.float(), gamma.float(), beta.float()
        Wg = Wf * g32[None, :]
        c = Wf @ b32 + (b.float() if b is not None else 0.0)
        if geglu:
            Wg, c = pack_geglu(Wg, c)
        Wh = Wg.to(torch.float16)
        self._put(name + ".w", Wh)
        # g from the kernel that also refreshes it after a style re-merge (same summation order: weight 0 restores the bits)
        self._put(name + ".g", torch.zeros(Wh.shape[0]), torch.float32)
        ops.ln_fold_refresh(self.w[name + ".w"], self.w[name + ".g"])
        self._put(name + ".c", c, torch.float32)
        self._put(name + ".lnw", g32, torch.float32)
        self._put(name + ".lnb", b32, torch.float32)

    def _pack_transformer(self, sd, p, kv_list, depth=1, heads=8):
        self._put(p + ".norm.g", sd[p + ".norm.weight"])
        self._put(p + ".norm.b", sd[p + ".norm.bias"])
        self._put(p + ".proj_in.w", pack_conv1x1(sd[p + ".proj_in.weight"]))      # 1x1 conv (SD1.5) or Linear (SDXL)
        self._put(p + ".proj_in.b", sd[p + ".proj_in.bias"])
        self._put(p + ".proj_out.w", pack_conv1x1(sd[p + ".proj_out.weight"]))
        self._put(p + ".proj_out.b", sd[p + ".proj_out.bias"])
        for k in range(depth):
            t, q = f"{p}.transformer_blocks.{k}", f"{p}.{k}"
            for n in ("norm1", "norm2", "norm3"):
                self._put(f"{q}.{n}.g", sd[f"{t}.{n}.weight"])
                self._put(f"{q}.{n}.b", sd[f"{t}.{n}.bias"])
            wqkv = torch.cat([sd[f"{t}.attn1.to_{n}.weight"] for n in "qkv"], 0).float()
            wq2, wff, bff = sd[f"{t}.attn2.to_q.weight"].float(), sd[f"{t}.ff.net.0.proj.weight"], sd[f"{t}.ff.net.0.proj.bias"]
            if Q_PRESCALE:
                # softmax scale d^-0.5 and the exp2 conversion log2(e), multiplied into the (bias-free) to_q rows in fp32 before
                # the one fp16 rounding of the weight: the attention kernels then take q as it comes out of the projection
                # (no second fp16 rounding of a scaled copy, no per-element multiply); lora.LoraStyle scales its to_q deltas alike
                C = wq2.shape[0]
                self.qs[q] = qs = (C // heads) ** -0.5 * 1.4426950408889634
                wqkv[:C] *= qs
                wq2 = wq2 * qs
            if LN_FOLD:
                self._put_ln_fold(q + ".qkv", wqkv, None, sd[f"{t}.norm1.weight"], sd[f"{t}.norm1.bias"])
                self._put_ln_fold(q + ".q2", wq2, None, sd[f"{t}.norm2.weight"], sd[f"{t}.norm2.bias"])
                self._put_ln_fold(q + ".ff1", wff, bff, sd[f"{t}.norm3.weight"], sd[f"{t}.norm3.bias"], geglu=True)
            else:
                self._put(q + ".qkv.w", wqkv)
                self._put(q + ".q2.w", wq2)
                wp, bp = pack_geglu(wff, bff)
                self._put(q + ".ff1.w", wp)
                self._put(q + ".ff1.b", bp)
            self._put(q + ".o1.w", sd[f"{t}.attn1.to_out.0.weight"])
            self._put(q + ".o1.b", sd[f"{t}.attn1.to_out.0.bias"])
            self._put(q + ".o2.w", sd[f"{t}.attn2.to_out.0.weight"])
            self._put(q + ".o2.b", sd[f"{t}.attn2.to_out.0.bias"])
            kv_list.append((q, sd[f"{t}.attn2.to_k.weight"], sd[f"{t}.attn2.to_v.weight"]))
            self._put(q + ".ff2.w", pack_ff2_cols(sd[f"{t}.ff.net.2.weight"]))      # columns in the GEGLU output's stored order
            self._put(q + ".ff2.b", sd[f"{t}.ff.net.2.bias"])

    # ---- per request: cross-attention K/V of all 16 layers (depend on the prompt only) -------
    def encode_context(self, ehs, B):
        """ehs: fp16 [B*77, ctx] -> kv_all [B*77, kv_total]."""
        kv = self.buf.get("kv_all", B * TEXT_SEQ_LEN, self.kv_total)
        ops.gemm(ehs, self.w["kv_all.w"], kv, img_rows=TEXT_SEQ_LEN)
        return kv

    # ---- per step: time embedding MLP + all time_emb_proj (depend on t and guidance only) ----
    def encode_added(self, add_in, B):
        """SDXL: add_in fp16 [B, 2816] = [pooled text embeds | sinusoid(6 size/crop ids)] -> aug_emb [B, temb] (per request)."""
        w = self.w
        h = self.buf.get("add_h", B, self.temb_dim)
        ops.linear_smallm(add_in, w["add.linear_1.w"], h, B, self.temb_dim, self.added_dim, bias=w["add.linear_1.b"], silu_out=True)
        aug = self.buf.get("add_aug", B, self.temb_dim)
        ops.linear_smallm(h, w["add.linear_2.w"], aug, B, self.temb_dim, self.temb_dim, bias=w["add.linear_2.b"])
        return aug

    def time_embed(self, t, wemb, B, aug=None):
        w = self.w
        ch0 = self.cfg["block_out_channels"][0]
        e0 = self.buf.get("te0", B, ch0)
        ops.timestep_embedding(float(t), e0, B, ch0)
        if self.has_cond and wemb is not None:
            e1 = self.buf.get("te1", B, ch0)
            ops.linear_smallm(wemb, w["te.cond.w"], e1, B, ch0, wemb.shape[1], res=e0)
        else:
            e1 = e0
        h = self.buf.get("te_h", B, self.temb_dim)
        ops.linear_smallm(e1, w["te.linear_1.w"], h, B, self.temb_dim, ch0, bias=w["te.linear_1.b"], silu_out=True)
        # The time embedding only ever feeds the ResnetBlock2D.time_emb_proj layers, each behind a SiLU: the SiLU is applied ONCE, in
        # the epilogue that produces the embedding (after the SDXL `aug` residual), not per output feature of the stacked
        # 20160 x 1280 projection (whose launch was VALU-bound by 8 x 1280 SiLUs per output row: 99 us at batch 8)
        temb = self.buf.get("temb", B, self.temb_dim)
        ops.linear_smallm(h, w["te.linear_2.w"], temb, B, self.temb_dim, self.temb_dim, bias=w["te.linear_2.b"], res=aug, silu_out=True)
        ta = self.buf.get("temb_all", B, self.temb_total)
        self._temb_proj(temb, ta)
        return ta

    MAX_HOISTED_STEPS = 64

    def time_embed_all(self, ts, wemb, B, aug=None):
        """time_embed of every sampler step in one launch per layer: -> [len(ts) * B, temb_total], rows step-major (the rows of
        step i are [i*B, (i+1)*B)).  Bit-identical to time_embed per step (a row's summation order does not depend on M); the
        stacked 20160 x 1280 projection is streamed once per pass instead of once per step."""
        w = self.w
        S, ch0 = len(ts), self.cfg["block_out_channels"][0]
        M = S * B
        e0 = self.buf.get("te0_all", M, ch0)
        ops.timestep_embedding_steps([float(t) for t in ts], e0, B, ch0)
        if self.has_cond and wemb is not None:
            e1 = self.buf.get("te1_all", M, ch0)
            ops.linear_rows(wemb, w["te.cond.w"], e1, M, ch0, wemb.shape[1], x_rows=B, res=e0)
        else:
            e1 = e0
        h = self.buf.get("te_h_all", M, self.temb_dim)
        ops.linear_rows(e1, w["te.linear_1.w"], h, M, self.temb_dim, ch0, bias=w["te.linear_1.b"], silu_out=True)
        temb = self.buf.get("temb_rows", M, self.temb_dim)
        ops.linear_rows(h, w["te.linear_2.w"], temb, M, self.temb_dim, self.temb_dim, bias=w["te.linear_2.b"], res=aug, res_rows=B,
                        silu_out=True)
        ta = self.buf.get("temb_all_rows", M, self.temb_total)
        self._temb_proj(temb, ta)
        return ta

    def _temb_proj(self, temb, ta):
        """All time_emb_proj layers as ONE MFMA GEMM (N = 20160 for SD1.5), every row its own "image" (img_rows = 1: the K
        partition is that of a single row, whatever the number of steps and requests stacked here) -- the one-wave-per-feature
        kernel is VALU-bound from ~16 rows on (330 us for the 32 rows of a batch-8 4-step pass, against ~15 us)."""
        w = self.w
        ops.gemm(temb, w["temb_all.w"], ta, bias=w["temb_all.b"], img_rows=1)

    def _res(self, p, x, C1, Cout, B, H, W, ta, x2=None, C2=0, out_role="res_out", x_st=None, x2_st=None):
        off, n = self.temb_off[p]
        return self.resnet(p, x, C1, Cout, B, H, W, self.cfg["norm_eps"], x2=x2, C2=C2, rowadd=ta[:, off:off + n],
                           out_role=out_role, x_st=x_st, x2_st=x2_st)

    def transformer(self, p, x, C, B, H, W, kv_all, out_role, x_st=None, heads=8, depth=1):
        w = self.w
        HW, M, d = H * W, B * H * W, C // heads
        hn = self.buf.get("gn", M, C)
        self.norm(x, w[p + ".norm.g"], w[p + ".norm.b"], hn, B, HW, C, x_st=x_st, eps=1e-6, silu=False)
        h = self.buf.get("tf_h", M, C)
        ops.gemm(hn, w[p + ".proj_in.w"], h, bias=w[p + ".proj_in.b"], img_rows=HW)
        n = self.buf.get("tf_ln", M, C)
        qkv = self.buf.get("tf_qkv", M, 3 * C)
        a = self.buf.get("tf_attn", M, C)
        q2 = self.buf.get("tf_q2", M, C)
        ff = self.buf.get("tf_ff", M, 4 * C)
        for k in range(depth):
            q = f"{p}.{k}"
            if LN_FOLD:
                ops.gemm_ln(h, w[q + ".qkv.w"], w[q + ".qkv.g"], w[q + ".qkv.c"], qkv, img_rows=HW)
            else:
                ops.layernorm(h, w[q + ".norm1.g"], w[q + ".norm1.b"], n, M, C)
                ops.gemm(n, w[q + ".qkv.w"], qkv, img_rows=HW)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, B, heads, HW, HW, d, ldq=3 * C, ldk=3 * C,
                          ldv=3 * C, ldo=C, scale=0.0 if Q_PRESCALE else None)
            ops.gemm(a, w[q + ".o1.w"], h, bias=w[q + ".o1.b"], res=h, img_rows=HW)
            if LN_FOLD:
                ops.gemm_ln(h, w[q + ".q2.w"], w[q + ".q2.g"], w[q + ".q2.c"], q2, img_rows=HW)
            else:
                ops.layernorm(h, w[q + ".norm2.g"], w[q + ".norm2.b"], n, M, C)
                ops.gemm(n, w[q + ".q2.w"], q2, img_rows=HW)
            off, _ = self.kv_off[q]
            ops.attention(q2, kv_all[:, off:off + C], kv_all[:, off + C:off + 2 * C], a, B, heads, HW, TEXT_SEQ_LEN, d,
                          ldq=C, ldk=self.kv_total, ldv=self.kv_total, ldo=C, scale=0.0 if Q_PRESCALE else None)
            ops.gemm(a, w[q + ".o2.w"], h, bias=w[q + ".o2.b"], res=h, img_rows=HW)
            if LN_FOLD and ops.mlp_fused_applies(M, C, HW):
                # norm3 -> ff.net.0 -> GEGLU -> ff.net.2 -> + h as ONE kernel: the [M, 4C] intermediate stays on the CU.  A launch
                # parameter like the tile shape: chosen from the total row count, bit-identical to the two launches below
                ops.mlp_geglu(h, w[q + ".ff1.w"], w[q + ".ff1.g"], w[q + ".ff1.c"], w[q + ".ff2.w"], w[q + ".ff2.b"], h, img_rows=HW)
                continue
            if LN_FOLD:
                ops.gemm_ln(h, w[q + ".ff1.w"], w[q + ".ff1.g"], w[q + ".ff1.c"], ff, epilogue=1, img_rows=HW)
            else:
                ops.layernorm(h, w[q + ".norm3.g"], w[q + ".norm3.b"], n, M, C)
                ops.gemm(n, w[q + ".ff1.w"], ff, bias=w[q + ".ff1.b"], epilogue=1, img_rows=HW)
            ops.gemm(ff, w[q + ".ff2.w"], h, bias=w[q + ".ff2.b"], res=h, img_rows=HW)
        out = self.buf.get(out_role, M, C)
        out_st = self.stats(out_role, M, C, HW)
        ops.gemm(h, w[p + ".proj_out.w"], out, bias=w[p + ".proj_out.b"], res=x, stats=out_st, img_rows=HW)
        return out, out_st

    def forward(self, lat, t, kv_all, wemb, B, h, w_, eps_out, taps=None, aug=None, ta=None):
        """lat fp32 [B,4,h,w] -> eps_out fp32 [B,h,w,4] (pixel-major).  aug: SDXL additional embedding (encode_added).
        ta: this step's rows of time_embed_all (the sampler computes all steps ahead of the loop); None: computed here."""
        cfg, wt = self.cfg, self.w
        boc = cfg["block_out_channels"]
        nb = len(boc)
        if ta is None:
            ta = self.time_embed(t, wemb, B, aug)
        H, W = h, w_
        x = self.buf.get("skip0", B * H * W, boc[0])
        ops.conv3x3_c4(lat, wt["conv_in.w"], x, B, H, W, boc[0], bias=wt["conv_in.b"])
        skips = [(x, boc[0], None)]          # (tensor, channels, fused statistics or None)
        ch, ns, st = boc[0], 1, None

        def tap(name, t_, C, H_, W_):
            if taps is not None:
                taps[name] = t_.reshape(B, H_, W_, C).permute(0, 3, 1, 2).float().cpu()

        tap("conv_in", x, ch, H, W)
        sizes = [(H, W)]                      # spatial size per level: a stride-2, padding-1 conv gives ceil(n / 2)
        for i in range(nb):
            for j in range(cfg["layers_per_block"]):
                p = f"down_blocks.{i}.resnets.{j}"
                attn = cfg["down_attn"][i]
                x, st = self._res(p, x, ch, boc[i], B, H, W, ta, out_role="res_out" if attn else f"skip{ns}", x_st=st)
                ch = boc[i]
                tap(p, x, ch, H, W)
                if attn:
                    p = f"down_blocks.{i}.attentions.{j}"
                    x, st = self.transformer(p, x, ch, B, H, W, kv_all, f"skip{ns}", x_st=st, heads=heads_at(cfg, i),
                                             depth=depth_at(cfg, i))
                    tap(p, x, ch, H, W)
                skips.append((x, ch, st))
                ns += 1
            if i < nb - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                H2, W2 = (H + 1) // 2, (W + 1) // 2
                y = self.buf.get(f"skip{ns}", B * H2 * W2, ch)
                st = self.stats(f"skip{ns}", B * H2 * W2, ch, H2 * W2)
                ops.conv3x3(x, wt[p + ".w"], y, B, H, W, ch, ch, bias=wt[p + ".b"], stride=2, stats=st)
                H, W, x = H2, W2, y
                sizes.append((H, W))
                skips.append((x, ch, st))
                ns += 1
        x, st = self._res("mid_block.resnets.0", x, ch, ch, B, H, W, ta, x_st=st)
        x, st = self.transformer("mid_block.attentions.0", x, ch, B, H, W, kv_all, "tf_out", x_st=st,
                                 heads=heads_at(cfg, nb - 1), depth=depth_at(cfg, nb - 1))
        x, st = self._res("mid_block.resnets.1", x, ch, ch, B, H, W, ta, out_role="cur", x_st=st)
        tap("mid_block.resnets.1", x, ch, H, W)
        rboc = tuple(reversed(boc))
        up_attn = tuple(reversed(cfg["down_attn"]))
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                s, sc, s_st = skips.pop()
                p = f"up_blocks.{i}.resnets.{j}"
                x, st = self._res(p, x, ch, rboc[i], B, H, W, ta, x2=s, C2=sc, out_role="res_out" if up_attn[i] else "cur",
                                  x_st=st, x2_st=s_st)
                ch = rboc[i]
                tap(p, x, ch, H, W)
                if up_attn[i]:
                    p = f"up_blocks.{i}.attentions.{j}"
                    x, st = self.transformer(p, x, ch, B, H, W, kv_all, "cur", x_st=st, heads=heads_at(cfg, nb - 1 - i),
                                             depth=depth_at(cfg, nb - 1 - i))
                    tap(p, x, ch, H, W)
            if i < nb - 1:
                # Upsample2D with output_size = the next skip's size (UNet2DConditionModel passes upsample_size whenever the
                # sample is not a multiple of 2**num_upsamplers): 2H or 2H-1
                p = f"up_blocks.{i}.upsamplers.0.conv"
                Ho, Wo = sizes[nb - 2 - i]
                y = self.buf.get("ups", B * Ho * Wo, ch)
                st = self.stats("ups", B * Ho * Wo, ch, Ho * Wo)
                if (Ho, Wo) == (2 * H, 2 * W):
                    ops.conv3x3(x, wt[p + ".w"], y, B, H, W, ch, ch, bias=wt[p + ".b"], ups=2 if UPS_PHASES else 1, stats=st)
                else:                # odd target: the conv pads the CROPPED upsampled image with zeros -> loader-fused form
                    ops.conv3x3(x, wt[p + (".w3" if UPS_PHASES else ".w")], y, B, H, W, ch, ch, bias=wt[p + ".b"], ups=1,
                                stats=st, out_hw=(Ho, Wo))
                H, W, x = Ho, Wo, y
                tap(f"up_blocks.{i}.upsamplers.0", x, ch, H, W)
        hn = self.buf.get("gn", B * H * W, ch)
        self._norm_conv_out(x, "conv_norm_out", st, hn, eps_out, B, H, W, ch, cfg["out_channels"], cfg["norm_eps"], mode=0)
        return eps_out


def scale_vae_residual_stream(sd: dict, cfg: dict, s: float) -> dict:
    """Exact reparametrisation of the AutoencoderKL decoder that shrinks its residual stream by ``s`` (< 1).

    The SDXL VAE overflows fp16 in its up blocks, which is why the reference pipeline upcasts it to fp32
    (``force_upcast`` in the checkpoint's vae/config.json; StableDiffusionXLPipeline.upcast_vae, reached from
    backends/cuda_worker.py:330-352).  Every consumer of the decoder's residual stream is a GroupNorm -- invariant to
    the scale of its input (up to eps) -- or a linear op followed by one, so the stream can carry s*x instead of x:
    scale what WRITES into it (conv_in, each resnet's conv2, the attention's to_out) by s, scale the BIAS of the
    linear ops that map stream to stream (conv_shortcut, upsampler conv) by s, leave everything else alone; the final
    conv_norm_out removes the factor again.  Same function, activations 1/s smaller, no fp32 pass needed."""
    out = dict(sd)

    def mul(name):
        out[name] = (sd[name].float() * s).to(sd[name].dtype)

    for n in ("decoder.conv_in.weight", "decoder.conv_in.bias", "decoder.mid_block.attentions.0.to_out.0.weight",
              "decoder.mid_block.attentions.0.to_out.0.bias"):
        mul(n)
    resnets = ["decoder.mid_block.resnets.0", "decoder.mid_block.resnets.1"]
    nb = len(cfg["block_out_channels"])
    for i in range(nb):
        resnets += [f"decoder.up_blocks.{i}.resnets.{j}" for j in range(cfg["layers_per_block"] + 1)]
        if i < nb - 1:
            mul(f"decoder.up_blocks.{i}.upsamplers.0.conv.bias")
    for r in resnets:
        mul(r + ".conv2.weight")
        mul(r + ".conv2.bias")
        if r + ".conv_shortcut.bias" in sd:
            mul(r + ".conv_shortcut.bias")
    return out


# ==================================================================================================
class VAEDecoderHip(_Net):
    def __init__(self, sd: dict, cfg: dict | None = None, device="cuda"):
        super().__init__(device)
        self.cfg = cfg = vae_config(cfg)
        boc = cfg["block_out_channels"]
        if float(cfg.get("residual_scale", 1.0)) != 1.0:
            sd = scale_vae_residual_stream(sd, cfg, float(cfg["residual_scale"]))
        self._put("pq.w", sd["post_quant_conv.weight"].reshape(4, 4), torch.float32)
        self._put("pq.b", sd["post_quant_conv.bias"], torch.float32)
        self._put("conv_in.w", pack_conv3x3(sd["decoder.conv_in.weight"]))
        self._put("conv_in.b", sd["decoder.conv_in.bias"])
        self._pack_resnet(sd, "decoder.mid_block.resnets.0")
        self._pack_resnet(sd, "decoder.mid_block.resnets.1")
        a = "decoder.mid_block.attentions.0"
        self._put("attn.norm.g", sd[a + ".group_norm.weight"])
        self._put("attn.norm.b", sd[a + ".group_norm.bias"])
        # the wide-head flash kernel is instantiated for d = 512 (every SD1.5 / SDXL AutoencoderKL); other widths take the
        # GEMM -> softmax -> transpose -> GEMM form
        self.flash_attn = VAE_FLASH_ATTN and boc[-1] == 512
        if self.flash_attn:
            self._put("attn.qkv.w", torch.cat([sd[f"{a}.{n}.weight"].reshape(boc[-1], boc[-1]) for n in ("to_q", "to_k", "to_v")], 0))
            self._put("attn.qkv.b", torch.cat([sd[f"{a}.{n}.bias"] for n in ("to_q", "to_k", "to_v")], 0))
        else:
            for n in ("to_q", "to_k", "to_v"):
                self._put(f"attn.{n}.w", sd[f"{a}.{n}.weight"].reshape(boc[-1], boc[-1]))
                self._put(f"attn.{n}.b", sd[f"{a}.{n}.bias"])
        self._put("attn.o.w", sd[a + ".to_out.0.weight"].reshape(boc[-1], boc[-1]))
        self._put("attn.o.b", sd[a + ".to_out.0.bias"])
        nb = len(boc)
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                self._pack_resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}")
            if i < nb - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                self._put(p + ".w", (pack_conv3x3_up2 if UPS_PHASES else pack_conv3x3)(sd[p + ".weight"]))
                self._put(p + ".b", sd[p + ".bias"])
        self._put("norm_out.g", sd["decoder.conv_norm_out.weight"])
        self._put("norm_out.b", sd["decoder.conv_norm_out.bias"])
        self._put("conv_out.w", pack_conv3x3(sd["decoder.conv_out.weight"]))
        self._put("conv_out.b", sd["decoder.conv_out.bias"])

    def mid_attention(self, x, C, B, H, W, x_st=None):
        w = self.w
        S, M = H * W, B * H * W
        hn = self.buf.get("gn", M, C)
        self.norm(x, w["attn.norm.g"], w["attn.norm.b"], hn, B, S, C, x_st=x_st, eps=1e-6, silu=False)
        out = self.buf.get("res_out2", M, C)
        out_st = self.stats("res_out2", M, C, S)
        if self.flash_attn:
            # one q|k|v GEMM, then the wide-head flash kernel (csrc/attention.hip attn_wide_kernel): no S x S scores in HBM
            qkv = self.buf.get("attn_qkv", M, 3 * C)
            ops.gemm(hn, w["attn.qkv.w"], qkv, bias=w["attn.qkv.b"], img_rows=S)
            o = self.buf.get("attn_o", M, C)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, B, 1, S, S, C, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            ops.gemm(o, w["attn.o.w"], out, bias=w["attn.o.b"], res=x, stats=out_st, img_rows=S)
            return out, out_st
        # The S x S product runs on the MFMA GEMM (N and K multiples of 64): S is padded to Sp with zero key rows / zero
        # V^T columns (allocated zero, never written), the softmax normalises over the S real keys and zeroes the padding.
        Sp = -(-S // 64) * 64
        q, v = self.buf.get("attn_q", M, C), self.buf.get("attn_v", M, C)
        k = self.buf.get("attn_k", B * Sp, C, zero=True)
        ops.gemm(hn, w["attn.to_q.w"], q, bias=w["attn.to_q.b"], img_rows=S)
        ops.gemm(hn, w["attn.to_v.w"], v, bias=w["attn.to_v.b"], img_rows=S)
        if Sp == S:
            ops.gemm(hn, w["attn.to_k.w"], k, bias=w["attn.to_k.b"], img_rows=S)
        else:
            ops.gemm(hn, w["attn.to_k.w"], k, bias=w["attn.to_k.b"], M=S, N=C, K=C, lda=C, ldo=C, batch=B, strideA=S * C,
                     strideW=0, strideO=Sp * C)
        sc = self.buf.get("attn_scores", B * S, Sp)
        ops.gemm(q, k, sc, M=S, N=Sp, K=C, lda=C, ldo=Sp, batch=B, strideA=S * C, strideW=Sp * C, strideO=S * Sp,
                 out_scale=C ** -0.5)
        ops.softmax_rows(sc, B * S, S, Sp)
        vt = self.buf.get("attn_vt", B * C, Sp, zero=True)
        ops.transpose(v, vt, S, C, ldi=C, ldo=Sp, batch=B, stride_in=S * C, stride_out=C * Sp)
        o = self.buf.get("attn_o", M, C)
        ops.gemm(sc, vt, o, M=S, N=C, K=Sp, lda=Sp, ldo=C, batch=B, strideA=S * Sp, strideW=C * Sp, strideO=S * C)
        ops.gemm(o, w["attn.o.w"], out, bias=w["attn.o.b"], res=x, stats=out_st, img_rows=S)
        return out, out_st

    def decode(self, lat, B, h, w_, rgb_out, img_f32=None, taps=None, use_tiling=True):
        """AutoencoderKL.decode with vae.enable_tiling() as the reference sets it (backends/cuda_worker.py:91): plain
        unless a latent side exceeds sample_size / 8, then diffusers' overlapping-tile decode."""
        tmin = int(self.cfg.get("sample_size", 512)) // 8
        if use_tiling and (h > tmin or w_ > tmin):
            return self.decode_tiled(lat, B, h, w_, rgb_out, img_f32)
        return self.decode_plain(lat, B, h, w_, rgb_out, img_f32, taps)

    def decode_tiled(self, lat, B, h, w_, rgb_out, img_f32=None, overlap=0.25):
        sample = int(self.cfg.get("sample_size", 512))
        tl = sample // 8
        stride, extent = int(tl * (1 - overlap)), int(sample * overlap)
        limit = sample - extent
        H, W = 8 * h, 8 * w_
        ys, xs = list(range(0, h, stride)), list(range(0, w_, stride))
        tiles = {}
        for i, y0 in enumerate(ys):
            for j, x0 in enumerate(xs):
                th, tw = min(tl, h - y0), min(tl, w_ - x0)
                sub = self.buf.get("tile_lat", B, 4, th, tw, dtype=torch.float32)
                sub.copy_(lat[:, :, y0:y0 + th, x0:x0 + tw])                       # strided gather of the latent tile
                img = self.buf.get(f"tile_img_{i}_{j}", B, 8 * th, 8 * tw, 3, dtype=torch.float32)
                scratch = self.buf.get("tile_u8", B, 8 * th, 8 * tw, 3, dtype=torch.uint8)
                self.decode_plain(sub, B, th, tw, scratch, img_f32=img)
                tiles[(i, j)] = (img, 8 * th, 8 * tw)
        oy = 0
        for i in range(len(ys)):
            ox = 0
            for j in range(len(xs)):
                img, th, tw = tiles[(i, j)]
                if i > 0:
                    a, ah, aw = tiles[(i - 1, j)]
                    ops.vae_blend(a, ah, aw, img, th, tw, B, min(ah, th, extent), True)
                if j > 0:
                    a, ah, aw = tiles[(i, j - 1)]
                    ops.vae_blend(a, ah, aw, img, th, tw, B, min(aw, tw, extent), False)
                ch, cw = min(limit, th), min(limit, tw)
                ops.vae_place_tile(img, th, tw, rgb_out, img_f32, H, W, B, oy, ox, ch, cw)
                ox += cw
            oy += min(limit, tiles[(i, 0)][1])
        return rgb_out

    def decode_plain(self, lat, B, h, w_, rgb_out, img_f32=None, taps=None):
        """lat fp32 [B,4,h,w] (UNet space) -> rgb_out u8 [B,8h,8w,3]; optional fp32 NHWC image copy."""
        cfg, wt = self.cfg, self.w
        boc = cfg["block_out_channels"]
        H, W = h, w_
        ch = boc[-1]

        def tap(name, t_, C, H_, W_):
            if taps is not None:
                taps[name] = t_.reshape(B, H_, W_, C).permute(0, 3, 1, 2).float().cpu()

        x = self.buf.get("cur", B * H * W, ch)
        ops.conv3x3_c4(lat, wt["conv_in.w"], x, B, H, W, ch, bias=wt["conv_in.b"], pre_w=wt["pq.w"], pre_b=wt["pq.b"],
                       in_scale=1.0 / cfg["scaling_factor"])
        tap("decoder.conv_in", x, ch, H, W)
        x, st = self.resnet("decoder.mid_block.resnets.0", x, ch, ch, B, H, W, 1e-6)
        tap("decoder.mid_block.resnets.0", x, ch, H, W)
        x, st = self.mid_attention(x, ch, B, H, W, x_st=st)
        tap("decoder.mid_block.attentions.0", x, ch, H, W)
        x, st = self.resnet("decoder.mid_block.resnets.1", x, ch, ch, B, H, W, 1e-6, out_role="cur", x_st=st)
        rboc = tuple(reversed(boc))
        nb = len(boc)
        roles = ("res_out", "cur")
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}"
                x, st = self.resnet(p, x, ch, rboc[i], B, H, W, 1e-6, out_role=roles[j & 1], x_st=st)
                ch = rboc[i]
                tap(p, x, ch, H, W)
            if i < nb - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                y = self.buf.get("ups", B * 4 * H * W, ch)
                st = self.stats("ups", B * 4 * H * W, ch, 4 * H * W)
                ops.conv3x3(x, wt[p + ".w"], y, B, H, W, ch, ch, bias=wt[p + ".b"], ups=2 if UPS_PHASES else 1, stats=st)
                H, W, x = 2 * H, 2 * W, y
                tap(f"decoder.up_blocks.{i}.upsamplers.0", x, ch, H, W)
        hn = self.buf.get("gn", B * H * W, ch)
        self._norm_conv_out(x, "norm_out", st, hn, rgb_out, B, H, W, ch, cfg["out_channels"], 1e-6, mode=1, out_f32=img_f32)
        return rgb_out
