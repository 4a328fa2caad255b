"""Host-side weight packing: diffusers parameter layouts -> the kernel layouts of liblcmhip.so.

* 3x3 conv OIHW            -> [Cout][ky][kx][Cin]   (implicit-GEMM k order: tap major, channel minor)
* 1x1 conv / linear         -> [N][K] as is
* GEGLU proj (value|gate)   -> rows interleaved in blocks of 16 so value/gate of one output column land in
                               the same lane of adjacent MFMA tiles (csrc/igemm.hip epilogue)
* GEGLU output / ff.net.2   -> the [M][F] intermediate is stored in "operand order" (geglu_col_order): ff.net.2's columns are
                               permuted alike, so the fused FeedForward kernel (csrc/mlp_fused.hip) and the two-launch form
                               multiply the same values in the same MFMA k slots
* to_q|to_k|to_v            -> one [3C][C] matrix; cross-attention to_k|to_v -> [2C][768]
"""
from __future__ import annotations

import torch


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    return w.permute(0, 2, 3, 1).contiguous().reshape(w.shape[0], -1)


# taps of the 3x3 kernel that read the same low-resolution pixel after a nearest-2x upsample, per (phase, 2x2 tap):
# even outputs (phase 0) see rows {y-1: k0, y: k1+k2}; odd outputs (phase 1) see rows {y: k0+k1, y+1: k2}
_UP2_TAPS = (((0,), (1, 2)), ((0, 1), (2,)))


def pack_conv3x3_up2(w: torch.Tensor) -> torch.Tensor:
    """Upsample2D = F.interpolate(scale_factor=2, mode="nearest") -> conv3x3 as four 2x2 phase convolutions on the
    low-resolution input: -> [4 (py*2+px) * Cout, 2*2*Cin] with the coinciding 3x3 taps summed in fp32."""
    Cout, Cin = w.shape[:2]
    wf = w.float()
    out = torch.empty(4, Cout, 2, 2, Cin, dtype=torch.float32)
    for py in range(2):
        for px in range(2):
            for dy in range(2):
                for dx in range(2):
                    acc = torch.zeros(Cout, Cin)
                    for ky in _UP2_TAPS[py][dy]:
                        for kx in _UP2_TAPS[px][dx]:
                            acc += wf[:, :, ky, kx]
                    out[py * 2 + px, :, dy, dx, :] = acc
    return out.reshape(4 * Cout, 4 * Cin).to(w.dtype)


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], -1).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor | None):
    """w: [2F][K] with rows [0,F) = value, [F,2F) = gate (diffusers GEGLU: proj(x).chunk(2))."""
    F2, K = w.shape
    Fh = F2 // 2
    assert Fh % 16 == 0
    val = w[:Fh].reshape(Fh // 16, 16, K)
    gate = w[Fh:].reshape(Fh // 16, 16, K)
    wp = torch.stack([val, gate], dim=1).reshape(F2, K).contiguous()
    bp = None
    if b is not None:
        bp = torch.stack([b[:Fh].reshape(-1, 16), b[Fh:].reshape(-1, 16)], dim=1).reshape(F2).contiguous()
    return wp, bp


def geglu_col_order(F: int) -> torch.Tensor:
    """order[s] = value channel stored at column s of the GEGLU output (csrc/igemm_common.h geglu_store_col): channel
    16 P + 4 fq + j sits at 32 (P >> 1) + 8 fq + 4 (P & 1) + j -- 8 consecutive columns are what one lane of the producing MFMA
    tiles holds (the quads of value/gate pairs 2u and 2u + 1)."""
    assert F % 32 == 0
    s = torch.arange(F)
    u, r = s // 32, s % 32
    fq, par, j = r // 8, (r % 8) // 4, r % 4
    return 16 * (2 * u + par) + 4 * fq + j


def pack_ff2_cols(w: torch.Tensor) -> torch.Tensor:
    """ff.net.2 weight [C][F] (or any delta of it) with its columns in the stored order of the GEGLU output."""
    return w[:, geglu_col_order(w.shape[1]).to(w.device)].contiguous()
