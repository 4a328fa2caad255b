"""Per-kernel parity: every C-ABI entry point against the plain torch fp32 op it replaces.

Inputs are fp16-rounded, references are computed on the CPU in fp32 from the same rounded values
(the ops diffusers dispatches: F.linear / F.conv2d / F.group_norm / F.layer_norm / SDPA).
Tolerance: fp16 output rounding (2^-11 relative) plus accumulation-order noise.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from sdlcm_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


def to_nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def from_nhwc(y, B, H, W):
    return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)


def close(got, ref, rtol=4e-3, atol=None, what=""):
    got = got.float().cpu()
    ref = ref.float().cpu()
    scale = ref.abs().max().item() + 1e-6
    atol = 2e-3 * scale if atol is None else atol
    err = (got - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    assert not bad.any(), f"{what}: max err {err.max().item():.4g} (scale {scale:.3g}), {bad.sum().item()} bad of {bad.numel()}"


def pack3x3(w):  # OIHW -> [O][ky][kx][I]
    return w.permute(0, 2, 3, 1).contiguous().reshape(w.shape[0], -1)


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (1000, 640, 1280), (77, 1280, 768), (64, 1280, 2560),
                                   (1, 64, 64), (333, 128, 128), (4096, 2560, 320)])
def test_gemm_plain(M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV))
    close(out, F.linear(a.float(), w.float(), b.float()), what=f"gemm {M}x{N}x{K}")


def test_gemm_epilogues_and_split():
    M, N, K1, K2, rpb = 512, 320, 640, 320, 128
    a1, a2 = rnd(M, K1, seed=1), rnd(M, K2, seed=2)
    w = rnd(N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5)
    b, res, radd = rnd(N, seed=4), rnd(M, N, seed=5), rnd(M // rpb, N, seed=6)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a1.to(DEV), w.to(DEV), out, a2=a2.to(DEV), bias=b.to(DEV), res=res.to(DEV), rowadd=radd.to(DEV),
             rows_per_batch=rpb, out_scale=0.5)
    ref = (F.linear(torch.cat([a1, a2], 1).float(), w.float(), b.float())
           + radd.float().repeat_interleave(rpb, 0)) * 0.5 + res.float()
    close(out, ref, what="gemm split/epilogue")


def test_gemm_geglu():
    M, C = 300, 320
    a = rnd(M, C, seed=1)
    w = rnd(8 * C, C, seed=2, scale=C ** -0.5)
    b = rnd(8 * C, seed=3)
    from sdlcm_amd.packing import pack_geglu
    wp, bp = pack_geglu(w, b)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), wp.to(DEV), out, bias=bp.to(DEV), epilogue=1)
    h, g = F.linear(a.float(), w.float(), b.float()).chunk(2, dim=-1)
    from sdlcm_amd.packing import geglu_col_order
    close(out, (h * F.gelu(g))[:, geglu_col_order(4 * C)], what="geglu")       # stored in operand order (packing.geglu_col_order)


def test_gemm_batched_strided():
    Z, M, N, K = 3, 256, 192, 512
    a, w = rnd(Z, M, K, seed=1), rnd(Z, N, K, seed=2, scale=K ** -0.5)
    out = torch.empty(Z, M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, lda=K, ldo=N, batch=Z, strideA=M * K, strideW=N * K,
             strideO=M * N, out_scale=0.25)
    close(out, torch.einsum("zmk,znk->zmn", a.float(), w.float()) * 0.25, what="batched gemm")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [
    (1, 64, 64, 320, 320, 1, 0), (2, 16, 16, 640, 1280, 1, 0), (1, 32, 32, 320, 320, 2, 0),
    (1, 8, 8, 1280, 1280, 1, 1), (3, 1, 1, 128, 64, 1, 0), (1, 2, 2, 64, 128, 2, 0), (1, 5, 7, 128, 64, 1, 0),
    (2, 24, 40, 128, 128, 1, 1)])
def test_conv3x3(B, H, W, Cin, Cout, stride, ups):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    radd = rnd(B, Cout, seed=4)
    res = rnd(B, Cout, Ho, Wo, seed=5)
    ref = ref + radd.float()[:, :, None, None] + res.float()
    out = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
    ops.conv3x3(to_nhwc(x).to(DEV), pack3x3(w).to(DEV), out, B, H, W, Cin, Cout, bias=b.to(DEV), rowadd=radd.to(DEV),
                res=to_nhwc(res).to(DEV), stride=stride, ups=ups)
    close(from_nhwc(out, B, Ho, Wo), ref, what="conv3x3")


@pytest.mark.parametrize("B,H,W,C1,C2,Cout,ups,silu,splitk", [
    (2, 16, 16, 320, 0, 320, 0, True, False), (1, 64, 64, 320, 0, 320, 0, True, False),
    (1, 16, 16, 1280, 640, 1280, 0, True, True), (2, 8, 8, 1280, 1280, 1280, 0, True, True),
    (1, 24, 40, 128, 0, 128, 0, True, False), (1, 12, 20, 256, 0, 128, 1, False, False),
    (1, 1, 1, 1280, 0, 1280, 0, True, True), (3, 5, 7, 64, 64, 64, 0, True, False), (1, 32, 32, 640, 320, 640, 0, True, True),
    (1, 128, 128, 128, 0, 128, 0, True, False)])
def test_conv3x3_gn_fused(B, H, W, C1, C2, Cout, ups, silu, splitk):
    """GroupNorm(32)+SiLU -> conv3x3 over the skip concat, against F.group_norm / F.silu / F.conv2d."""
    C = C1 + C2
    x1 = rnd(B, C1, H, W, seed=1) * 1.5 + 0.3
    x2 = rnd(B, C2, H, W, seed=2) * 0.7 - 0.5 if C2 else None
    xc = torch.cat([x1, x2], 1) if C2 else x1
    gamma, beta = (1 + 0.1 * rnd(C, seed=3).float()).half(), rnd(C, seed=4, scale=0.1)
    w = rnd(Cout, C, 3, 3, seed=5, scale=(9 * C) ** -0.5)
    b, radd = rnd(Cout, seed=6), rnd(B, Cout, seed=7)
    h = F.group_norm(xc.float(), 32, gamma.float(), beta.float(), 1e-5)
    if silu:
        h = F.silu(h)
    if ups:
        h = F.interpolate(h, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(h, w.float(), b.float(), padding=1) + radd.float()[:, :, None, None]
    Ho, Wo = ref.shape[2:]
    res = rnd(B, Cout, Ho, Wo, seed=8)
    ref = ref + res.float()
    d1 = to_nhwc(x1).to(DEV)
    d2 = to_nhwc(x2).to(DEV) if C2 else None
    ws = torch.empty(ops.groupnorm_ws_bytes(B, H * W, C) // 4 + 16, dtype=torch.float32, device=DEV)
    scale = torch.empty(B, C, dtype=torch.float32, device=DEV)
    shift = torch.empty(B, C, dtype=torch.float32, device=DEV)
    ops.groupnorm_affine(d1, gamma.to(DEV), beta.to(DEV), scale, shift, B, H * W, C1, ws, x2=d2, C2=C2)
    out = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
    skws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws if splitk else None)
    try:
        ops.conv3x3_gn(d1, pack3x3(w).to(DEV), out, B, H, W, C1, Cout, x2=d2, C2=C2, gn_scale=scale, gn_shift=shift,
                       silu=silu, bias=b.to(DEV), rowadd=radd.to(DEV), res=to_nhwc(res).to(DEV), ups=ups)
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    close(from_nhwc(out, B, Ho, Wo), ref, rtol=6e-3, what="conv3x3_gn")


def _ref_gn(xc, B, HW, C, gamma, beta, eps, silu):
    ref = F.group_norm(xc.float().reshape(B, HW, C).permute(0, 2, 1), 32, gamma.float(), beta.float(), eps)
    if silu:
        ref = F.silu(ref)
    return ref.permute(0, 2, 1).reshape(B * HW, C)


@pytest.mark.parametrize("B,H,W,Cin,Cout,splitk,stride", [
    (2, 16, 16, 320, 320, False, 1), (1, 64, 64, 320, 640, False, 1), (1, 8, 8, 1280, 1280, True, 1),
    (1, 16, 16, 640, 1280, True, 1), (2, 24, 40, 128, 128, False, 1), (1, 32, 32, 320, 320, False, 2),
    (1, 4, 4, 1280, 1280, True, 1), (1, 128, 128, 128, 256, False, 1)])
def test_fused_groupnorm_stats_from_conv(B, H, W, Cin, Cout, splitk, stride):
    """conv3x3 writes the GroupNorm statistics of its output in the epilogue (or split-K reduce); the
    finalize+apply pair must reproduce F.group_norm of the stored fp16 tensor."""
    x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
    w = pack3x3(rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    HW = Ho * Wo
    out = torch.empty(B * HW, Cout, dtype=torch.float16, device=DEV)
    st = ops.Stats(torch.zeros(ops.stats_floats(B * HW, Cout, HW), dtype=torch.float32, device=DEV))
    skws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws if splitk else None)
    try:
        ops.conv3x3(x, w, out, B, H, W, Cin, Cout, bias=b, stride=stride, stats=st)
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    gamma, beta = (1 + 0.1 * rnd(Cout, seed=4).float()).half(), rnd(Cout, seed=5, scale=0.1)
    ref = _ref_gn(out.cpu(), B, HW, Cout, gamma, beta, 1e-5, True)
    y = torch.empty_like(out)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, Cout) // 4 + 16, dtype=torch.float32, device=DEV)
    if st.P > 0:
        ys = []
        try:
            for fused_bytes in (0, 1 << 40):            # two launches (finalize + apply) vs the single small-tensor launch
                ops.set_gn_fused_bytes(fused_bytes)
                y = torch.empty_like(out)
                ops.groupnorm_from_stats(out, gamma.to(DEV), beta.to(DEV), y, B, HW, Cout, st, ws)
                close(y, ref, what=f"gn from fused stats (P={st.P}, single-launch limit {fused_bytes})")
                ys.append(y)
        finally:
            ops.set_gn_fused_bytes(8 << 20)
        assert torch.equal(ys[0], ys[1])                # same arithmetic, same order
    else:
        assert HW % 32 != 0, "statistics should have been produced for this shape"


def test_fused_groupnorm_stats_concat_and_gemm():
    """Statistics from two producers (a gemm with residual and a conv) drive the GroupNorm of their channel concat,
    with a group straddling the concat boundary (C1=1280, C2=640, 60 channels per group)."""
    B, H, W, C1, C2 = 1, 16, 16, 1280, 640
    HW = H * W
    a = rnd(B * HW, 640, seed=1).to(DEV)
    w1 = rnd(C1, 640, seed=2, scale=640 ** -0.5).to(DEV)
    r1 = rnd(B * HW, C1, seed=3).to(DEV)
    x1 = torch.empty(B * HW, C1, dtype=torch.float16, device=DEV)
    st1 = ops.Stats(torch.zeros(ops.stats_floats(B * HW, C1, HW), dtype=torch.float32, device=DEV))
    ops.gemm(a, w1, x1, res=r1, stats=st1, img_rows=HW)
    xin = to_nhwc(rnd(B, 320, H, W, seed=4)).to(DEV)
    w2 = pack3x3(rnd(C2, 320, 3, 3, seed=5, scale=(9 * 320) ** -0.5)).to(DEV)
    x2 = torch.empty(B * HW, C2, dtype=torch.float16, device=DEV)
    st2 = ops.Stats(torch.zeros(ops.stats_floats(B * HW, C2, HW), dtype=torch.float32, device=DEV))
    ops.conv3x3(xin, w2, x2, B, H, W, 320, C2, stats=st2)
    torch.cuda.synchronize()
    assert st1.P > 0 and st2.P > 0
    C = C1 + C2
    gamma, beta = (1 + 0.1 * rnd(C, seed=6).float()).half(), rnd(C, seed=7, scale=0.1)
    ref = _ref_gn(torch.cat([x1.cpu(), x2.cpu()], 1), B, HW, C, gamma, beta, 1e-5, True)
    y = torch.empty(B * HW, C, dtype=torch.float16, device=DEV)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C) // 4 + 16, dtype=torch.float32, device=DEV)
    try:
        for fused_bytes in (0, 1 << 40):
            ops.set_gn_fused_bytes(fused_bytes)
            ops.groupnorm_from_stats(x1, gamma.to(DEV), beta.to(DEV), y, B, HW, C1, st1, ws, x2=x2, C2=C2, st2=st2)
            close(y, ref, what=f"gn from fused stats over a concat (single-launch limit {fused_bytes})")
    finally:
        ops.set_gn_fused_bytes(8 << 20)


@pytest.mark.parametrize("B,H,W,Cin,Cout,ups,splitk", [(2, 12, 20, 128, 192, 0, False), (1, 9, 7, 64, 64, 0, False),
                                                      (1, 6, 10, 192, 128, 1, False), (1, 64, 64, 320, 320, 0, True),
                                                      (1, 8, 8, 1280, 1280, 0, True), (2, 16, 16, 640, 1280, 0, True),
                                                      (1, 1, 1, 1280, 1280, 0, True)])
def test_conv_halo_pipelined_variant_is_bit_identical(B, H, W, Cin, Cout, ups, splitk):
    """The pipelined (3-stage weight ring, double-buffered halo) and the single-buffer halo kernels walk K in
    the same order: outputs must be bit-identical."""
    x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
    w = pack3x3(rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws if splitk else None)
    outs = []
    try:
        for thr in (0, 1 << 30):
            ops.set_halo_pipe_threshold(thr)
            o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
            ops.conv3x3(x, w, o, B, H, W, Cin, Cout, bias=b, ups=ups)
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        ops.set_halo_pipe_threshold(768)
        ops.set_workspace(None)
    assert torch.equal(outs[0], outs[1])
    # plan variant 3: a whole kernel row of taps per K-step -- same K order, bit-identical again
    from sdlcm_amd import lib as _l
    for bm, bn in ((128, 64), (64, 64), (128, 128), (64, 128)):
        if Cout % bn or (min(Ho, Wo) <= 8 and bm == 128):
            continue
        ops.set_workspace(ws if splitk else None)
        try:
            pair = []
            for var in (2, 3):
                ops.plan_clear()
                ops.plan_set(2, B * Ho * Wo, Cout, 9 * Cin, Wo << 1, bm, bn, 2 if (splitk and Cin >= 128) else 1, var)
                o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
                ops.conv3x3(x, w, o, B, H, W, Cin, Cout, bias=b, ups=ups)
                pair.append(o)
            torch.cuda.synchronize()
        finally:
            ops.plan_reset()
            ops.set_workspace(None)
        assert torch.equal(pair[0], pair[1]), f"row-step variant differs at tile {bm}x{bn}"
    xin = from_nhwc(x.cpu().float(), B, H, W)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.cpu().float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.cpu().float(), padding=1)
    close(from_nhwc(outs[1], B, Ho, Wo), ref, what="conv halo pipelined")


@pytest.mark.parametrize("B,H,W,Cin,Cout,splitk", [(2, 12, 20, 128, 192, False), (1, 9, 7, 64, 64, False), (1, 6, 10, 192, 128, False),
                                                  (1, 16, 16, 1280, 1280, True), (2, 32, 32, 640, 640, True),
                                                  (1, 1, 1, 128, 64, False), (1, 64, 48, 128, 128, False)])
def test_conv_upsample_phase_decomposition(B, H, W, Cin, Cout, splitk):
    """Upsample2D (F.interpolate nearest 2x -> conv3x3) as four 2x2 phase convolutions on the low-resolution input
    (ups=2, packing.pack_conv3x3_up2): same result as the reference op order, pipelined == single-buffer bitwise,
    fused GroupNorm statistics of the output intact."""
    from sdlcm_amd.packing import pack_conv3x3_up2
    x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
    w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    wp = pack_conv3x3_up2(w4).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    Ho, Wo = 2 * H, 2 * W
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws if splitk else None)
    outs, sts = [], []
    try:
        for thr in (0, 1 << 30):
            ops.set_halo_pipe_threshold(thr)
            o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
            st = ops.Stats(torch.zeros(ops.stats_floats(B * Ho * Wo, Cout, Ho * Wo), dtype=torch.float32, device=DEV))
            ops.conv3x3(x, wp, o, B, H, W, Cin, Cout, bias=b, ups=2, stats=st)
            outs.append(o)
            sts.append(st)
        torch.cuda.synchronize()
    finally:
        ops.set_halo_pipe_threshold(768)
        ops.set_workspace(None)
    assert torch.equal(outs[0], outs[1])
    if H >= 6 and W >= 6:      # plan variant 3 (a row of the 2x2 taps per K-step) against variant 2, bitwise
        pair = []
        try:
            for var in (2, 3):
                ops.plan_clear()
                ops.plan_set(2, B * Ho * Wo, Cout, 4 * Cin, Wo << 1, 64, 64, 1, var)
                o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
                ops.conv3x3(x, wp, o, B, H, W, Cin, Cout, bias=b, ups=2)
                pair.append(o)
            torch.cuda.synchronize()
        finally:
            ops.plan_reset()
        assert torch.equal(pair[0], pair[1])
        close(pair[1], outs[0], what="row-step phase conv vs default plan")
    xin = F.interpolate(from_nhwc(x.cpu().float(), B, H, W), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w4.float(), b.cpu().float(), padding=1)
    close(from_nhwc(outs[1], B, Ho, Wo), ref, what="phase-decomposed upsample conv")
    if Cout % 32 == 0 and sts[1].P > 0:
        gamma, beta = (1 + 0.1 * rnd(Cout, seed=6).float()).half(), rnd(Cout, seed=7, scale=0.1)
        y = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
        gws = torch.empty(ops.groupnorm_ws_bytes(B, Ho * Wo, Cout) // 4 + 16, dtype=torch.float32, device=DEV)
        ops.groupnorm_from_stats(outs[1], gamma.to(DEV), beta.to(DEV), y, B, Ho * Wo, Cout, sts[1], gws)
        close(y, _ref_gn(outs[1].cpu(), B, Ho * Wo, Cout, gamma, beta, 1e-5, True), what="gn from phase-conv statistics")


@pytest.mark.parametrize("variant", [-1, 1, 2, 4])
def test_160_wide_tiles_gemm_and_conv(variant):
    """The BN=160 tile (N = 320 / 640 / 960: 5 n-fragments per wave, fewer L2->LDS bytes per FLOP than 64-wide) is a
    plan-table choice: forced here for GEMM (with residual, split-K, fused statistics) and the halo conv (plain,
    pipelined, phase-decomposed upsample) against the torch ops."""
    from sdlcm_amd.packing import pack_conv3x3_up2
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    ops.set_kernel_variant(variant)
    try:
        for (M, N, K, bm, sp) in [(4096, 320, 320, 128, 1), (1000, 640, 1280, 128, 1), (256, 960, 1280, 64, 4), (8192, 320, 1280, 64, 1)]:
            ops.plan_clear()
            ops.plan_set(0, M, N, K, 1, bm, 160, sp)
            a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
            out = torch.empty(M, N, dtype=torch.float16, device=DEV)
            hw = 256 if M % 256 == 0 else 0
            st = ops.Stats(torch.zeros(ops.stats_floats(M, N, hw), dtype=torch.float32, device=DEV))
            ops.gemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV), res=r.to(DEV), stats=st if hw else None, img_rows=hw)
            close(out, F.linear(a.float(), w.float(), b.float()) + r.float(), what=f"gemm 160-wide {M}x{N}x{K}")
            if hw and N % 32 == 0 and st.P > 0:
                Bimg = M // hw
                gamma, beta = (1 + 0.1 * rnd(N, seed=6).float()).half(), rnd(N, seed=7, scale=0.1)
                y = torch.empty(M, N, dtype=torch.float16, device=DEV)
                gws = torch.empty(ops.groupnorm_ws_bytes(Bimg, hw, N) // 4 + 16, dtype=torch.float32, device=DEV)
                ops.groupnorm_from_stats(out, gamma.to(DEV), beta.to(DEV), y, Bimg, hw, N, st, gws)
                close(y, _ref_gn(out.cpu(), Bimg, hw, N, gamma, beta, 1e-5, True), what="gn from 160-wide gemm statistics")
        for (B, H, W, Cin, Cout, bm, sp, ups, thr) in [(2, 16, 16, 128, 320, 128, 1, 0, 0), (1, 12, 20, 192, 320, 64, 1, 0, 1 << 30),
                                                       (1, 8, 8, 640, 640, 64, 5, 0, 1 << 30), (1, 16, 16, 128, 320, 128, 1, 2, 0),
                                                       (2, 8, 8, 256, 320, 64, 2, 2, 1 << 30)]:
            ops.plan_clear()
            ops.set_halo_pipe_threshold(thr)
            Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
            ops.plan_set(2, B * Ho * Wo, Cout, (4 if ups else 9) * Cin, Wo << 1, bm, 160, sp)
            x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
            w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
            wk = (pack_conv3x3_up2(w4) if ups else pack3x3(w4)).to(DEV)
            b = rnd(Cout, seed=3).to(DEV)
            o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
            ops.conv3x3(x, wk, o, B, H, W, Cin, Cout, bias=b, ups=ups)
            xin = from_nhwc(x.cpu().float(), B, H, W)
            if ups:
                xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
            close(from_nhwc(o, B, Ho, Wo), F.conv2d(xin, w4.float(), b.cpu().float(), padding=1), what=f"conv 160-wide {H}x{W} {Cin}->{Cout} ups={ups}")
    finally:
        ops.plan_reset()
        ops.set_kernel_variant(-1)
        ops.set_halo_pipe_threshold(768)
        ops.set_workspace(None)


def test_conv_halo_matches_row_gather_igemm():
    """The two 3x3 implementations agree to fp32 summation-order noise on a plain conv (border + m-tail)."""
    B, H, W, Cin, Cout = 2, 20, 28, 192, 128
    x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
    w = pack3x3(rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)).to(DEV)
    outs = []
    for impl in (0, 1):
        ops.set_conv_impl(impl)
        o = torch.empty(B * H * W, Cout, dtype=torch.float16, device=DEV)
        ops.conv3x3(x, w, o, B, H, W, Cin, Cout)
        outs.append(o.float().cpu())
    ops.set_conv_impl(1)
    assert (outs[0] - outs[1]).abs().max() < 4e-3


@pytest.mark.parametrize("B,H,Cin,Cout", [(1, 8, 1280, 1280), (1, 16, 640, 1280), (1, 32, 640, 640), (1, 64, 320, 320),
                                         (2, 8, 2560, 1280)])
def test_conv3x3_splitk(B, H, Cin, Cout):
    """Deep-K / small-M layers run split-K once a workspace is registered; result must match the unsplit one."""
    x = rnd(B, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b, radd, res = rnd(Cout, seed=3), rnd(B, Cout, seed=4), rnd(B, Cout, H, H, seed=5)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + radd.float()[:, :, None, None] + res.float()
    args = (to_nhwc(x).to(DEV), pack3x3(w).to(DEV))
    kw = dict(bias=b.to(DEV), rowadd=radd.to(DEV), res=to_nhwc(res).to(DEV))
    o1 = torch.empty(B * H * H, Cout, dtype=torch.float16, device=DEV)
    o2 = torch.empty_like(o1)
    ops.set_workspace(None)
    ops.conv3x3(*args, o1, B, H, H, Cin, Cout, **kw)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    try:
        ops.conv3x3(*args, o2, B, H, H, Cin, Cout, **kw)
        ops.conv3x3(*args, o1, B, H, H, Cin, Cout, **kw)     # determinism of the slab reduce
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    close(from_nhwc(o2, B, H, H), ref, what="conv3x3 split-K")
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("variant", [0, 1, 3, 4])
def test_contraction_kernel_variants(variant):
    """The LDS-DMA pipeline variants must reproduce the register-staged kernel bit for bit
    (same MFMA contraction order), on conv (borders, upsample, stride 2, m-tail) and split-source gemm."""
    cases = [(2, 12, 20, 128, 192, 1, 0), (1, 9, 7, 64, 64, 2, 0), (1, 6, 10, 192, 128, 1, 1), (1, 64, 64, 320, 320, 1, 0),
             (1, 8, 8, 1280, 1280, 1, 0)]
    ws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_conv_impl(0)
    try:
        for (B, H, W, Cin, Cout, stride, ups) in cases:
            x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
            w = pack3x3(rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)).to(DEV)
            b = rnd(Cout, seed=3).to(DEV)
            Ho, Wo = ((2 * H, 2 * W) if ups else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
            outs = []
            for v in (2, variant):
                ops.set_kernel_variant(v)
                ops.set_workspace(ws)
                o = torch.empty(B * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
                ops.conv3x3(x, w, o, B, H, W, Cin, Cout, bias=b, stride=stride, ups=ups)
                outs.append(o)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[1]), f"variant {variant} differs on conv {(B, H, W, Cin, Cout, stride, ups)}"
        M, N, K1, K2 = 300, 320, 640, 320
        a1, a2 = rnd(M, K1, seed=1).to(DEV), rnd(M, K2, seed=2).to(DEV)
        w = rnd(N, K1 + K2, seed=3, scale=0.03).to(DEV)
        outs = []
        for v in (2, variant):
            ops.set_kernel_variant(v)
            o = torch.empty(M, N, dtype=torch.float16, device=DEV)
            ops.gemm(a1, w, o, a2=a2)
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
    finally:
        ops.set_kernel_variant(-1)     # library default (auto)
        ops.set_conv_impl(1)
        ops.set_workspace(None)


@pytest.mark.parametrize("M,N,K,epi", [(32768, 320, 320, 0), (8192, 2560, 320, 1), (16384, 1280, 640, 0), (20000, 960, 320, 0),
                                       (32768, 320, 1280, 0)])
def test_gemm_persistent_over_n_is_bit_identical(M, N, K, epi):
    """Walking several n-tiles per workgroup keeps each tile's K order: results must not change at all."""
    a, w, b = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5).to(DEV), rnd(N, seed=3).to(DEV)
    res = rnd(M, N, seed=4).to(DEV) if epi == 0 else None
    outs = []
    try:
        for on in (0, 1):
            ops.set_persist_n(on)
            o = torch.empty(M, N // 2 if epi == 1 else N, dtype=torch.float16, device=DEV)
            st = ops.Stats(torch.zeros(ops.stats_floats(M, N, 4096 if M % 4096 == 0 else 0), dtype=torch.float32, device=DEV)) if epi == 0 else None
            ops.gemm(a, w, o, bias=b, res=res, epilogue=epi, stats=st, img_rows=4096 if M % 4096 == 0 else 0)
            outs.append((o, st.buf.clone() if st is not None else None, st.P if st is not None else 0))
        torch.cuda.synchronize()
    finally:
        ops.set_persist_n(0)
    assert torch.equal(outs[0][0], outs[1][0])
    if epi == 0:
        assert outs[0][2] == outs[1][2] and torch.equal(outs[0][1], outs[1][1])
        ref = F.linear(a.float().cpu(), w.float().cpu(), b.float().cpu()) + res.float().cpu()
        close(outs[1][0], ref, what="persistent gemm")


def test_gemm_splitk():
    M, N, K = 64, 1280, 5120
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ws = torch.empty(8 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    try:
        ops.gemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV))
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    close(out, F.linear(a.float(), w.float(), b.float()), what="gemm split-K")


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("B,H,W,Cout", [(2, 24, 16, 320), (3, 5, 7, 128), (1, 9, 13, 512)])
def test_conv_c4(pre, B, H, W, Cout):
    """conv_in from the fp32 latent on the matrix pipe with the input split hi + lo (csrc/misc.hip): against F.conv2d in fp64 on
    the same fp16 weights the only error left is the fp16 rounding of the output (pixel counts that are no multiple of the
    16-pixel tile included)."""
    lat = torch.randn(B, 4, H, W, generator=torch.Generator().manual_seed(1))
    w = rnd(Cout, 4, 3, 3, seed=2, scale=1 / 6)
    b = rnd(Cout, seed=3)
    pw = torch.randn(4, 4, generator=torch.Generator().manual_seed(4)) * 0.5
    pb = torch.randn(4, generator=torch.Generator().manual_seed(5)) * 0.1
    z = lat
    if pre:
        z = F.conv2d(lat / 0.18215, pw[:, :, None, None], pb)
    ref = F.conv2d(z.double(), w.double(), b.double(), padding=1)
    out = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv3x3_c4(lat.to(DEV), pack3x3(w).to(DEV), out, B, H, W, Cout, bias=b.to(DEV),
                   pre_w=pw.to(DEV) if pre else None, pre_b=pb.to(DEV) if pre else None,
                   in_scale=1 / 0.18215 if pre else 1.0)
    got = from_nhwc(out, B, H, W).cpu().double()
    close(got.float(), ref.float(), what="conv_c4")
    err = (got - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -11 + 2e-5).all(), f"conv_c4: beyond the output rounding: {err.max().item():.3e}"


@pytest.mark.parametrize("Cin,Cout,mode,shape", [(320, 4, 0, (2, 20, 12)), (128, 3, 1, (2, 20, 12)), (256, 3, 1, (2, 20, 12)),
                                                 (128, 3, 1, (1, 9, 72)), (128, 4, 0, (3, 5, 33)), (64, 3, 1, (1, 8, 40))])
def test_conv_smalln(Cin, Cout, mode, shape):
    """conv_out (<= 4 output channels).  Cin % 64 == 0 takes the MFMA kernel (8 x 16 pixel patches: the sizes give ragged
    patches in both directions), other Cin the VALU kernels."""
    B, H, W = shape
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    of = torch.empty(B * H * W, Cout, dtype=torch.float32, device=DEV)
    if mode == 0:
        ops.conv3x3_smalln(to_nhwc(x).to(DEV), pack3x3(w).to(DEV), of, B, H, W, Cin, Cout, bias=b.to(DEV), mode=0)
        close(from_nhwc(of, B, H, W), ref, rtol=1e-4, atol=1e-4, what="conv_smalln f32")
    else:
        o8 = torch.empty(B * H * W, Cout, dtype=torch.uint8, device=DEV)
        ops.conv3x3_smalln(to_nhwc(x).to(DEV), pack3x3(w).to(DEV), o8, B, H, W, Cin, Cout, bias=b.to(DEV), mode=1,
                           out_f32=of)
        close(from_nhwc(of, B, H, W), ref, rtol=1e-4, atol=1e-4, what="conv_smalln f32 copy")
        # u8 must be the reference post-process of the kernel's own float output (ties/half-even included)
        exp = (np.clip(of.cpu().numpy() / 2 + 0.5, 0, 1) * 255).round().astype(np.uint8)
        got = o8.cpu().numpy()
        assert np.abs(got.astype(int) - exp.astype(int)).max() <= 1
        assert (got != exp).mean() < 1e-3


@pytest.mark.parametrize("C,Cout,mode,shape", [(128, 3, 1, (2, 24, 40)), (320, 4, 0, (2, 16, 16)), (128, 3, 1, (1, 13, 21)), (64, 3, 0, (3, 8, 16)),
                                               (128, 3, 1, (1, 20, 88)), (128, 3, 0, (2, 9, 72))])
def test_conv_out_with_fused_groupnorm_is_bit_identical(C, Cout, mode, shape):
    """conv_norm_out -> SiLU -> conv_out with the GroupNorm applied inside the conv's staging pass (lcm_conv3x3_smalln_gn on the
    raw tensor + scale / shift tables) against the two-launch form (GroupNorm-apply to memory, then the conv): same statistics,
    same fp16 rounding of the normalised value -> identical bits, RGB8 and fp32 modes."""
    B, H, W = shape
    M, HW, Cin0 = B * H * W, H * W, 64
    x0 = rnd(M, Cin0, seed=1).to(DEV)
    w0 = (rnd(C, 9 * Cin0, seed=2) * (9 * Cin0) ** -0.5).to(DEV)
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    st = ops.Stats(torch.zeros(ops.stats_floats(M, C, HW), dtype=torch.float32, device=DEV))
    ops.conv3x3(x0, w0, x, B, H, W, Cin0, C, stats=st)
    assert st.P > 0
    gamma, beta = (1 + 0.2 * rnd(C, seed=3)).to(DEV), (0.1 * rnd(C, seed=4)).to(DEV)
    wo = pack3x3(rnd(Cout, C, 3, 3, seed=5, scale=(9 * C) ** -0.5)).to(DEV)
    bo = rnd(Cout, seed=6, scale=0.1).to(DEV)
    ws = torch.empty(max(ops.groupnorm_ws_bytes(B, HW, C) // 4, 1024), dtype=torch.float32, device=DEV)
    hn = torch.empty(M, C, dtype=torch.float16, device=DEV)
    ops.groupnorm_from_stats(x, gamma, beta, hn, B, HW, C, st, ws, eps=1e-6, silu=True)
    dt = torch.uint8 if mode == 1 else torch.float32
    a, b = torch.empty(M, Cout, dtype=dt, device=DEV), torch.full((M, Cout), 7, dtype=dt, device=DEV)
    fa, fb = torch.empty(M, Cout, dtype=torch.float32, device=DEV), torch.full((M, Cout), 7.0, dtype=torch.float32, device=DEV)
    ops.conv3x3_smalln(hn, wo, a, B, H, W, C, Cout, bias=bo, mode=mode, out_f32=fa)
    sc, sh = ops.groupnorm_tables_from_stats(gamma, beta, B, HW, C, st, ws, eps=1e-6)
    ops.conv3x3_smalln(x, wo, b, B, H, W, C, Cout, bias=bo, mode=mode, out_f32=fb, gn_scale=sc, gn_shift=sh, silu=True)
    torch.cuda.synchronize()
    assert torch.equal(fa, fb) and torch.equal(a, b)
    ref = F.conv2d(from_nhwc(hn, B, H, W).float().cpu(), torch.from_numpy(np.ascontiguousarray(wo.cpu().numpy().reshape(Cout, 3, 3, C).transpose(0, 3, 1, 2))).float(),
                   bo.float().cpu(), padding=1)
    close(from_nhwc(fb, B, H, W), ref, rtol=1e-4, atol=1e-4, what="conv_out fused gn")


@pytest.mark.parametrize("B,HW,C1,C2,silu,eps", [(2, 4096, 320, 0, True, 1e-5), (1, 256, 1280, 640, True, 1e-5),
                                                (2, 64, 1280, 1280, True, 1e-5), (1, 1024, 640, 320, True, 1e-5),
                                                (1, 1, 1280, 0, True, 1e-5), (1, 5000, 128, 0, True, 1e-6),
                                                (3, 100, 512, 0, False, 1e-6)])
def test_groupnorm(B, HW, C1, C2, silu, eps):
    C = C1 + C2
    x1 = rnd(B * HW, C1, seed=1) * 2 + 0.5
    x2 = (rnd(B * HW, C2, seed=2) * 0.5 - 1.0) if C2 else None
    gamma, beta = (1 + 0.1 * rnd(C, seed=3).float()).half(), rnd(C, seed=4, scale=0.1)
    xc = torch.cat([x1, x2], 1) if C2 else x1
    ref = F.group_norm(xc.float().reshape(B, HW, C).permute(0, 2, 1), 32, gamma.float(), beta.float(), eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * HW, C)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C) // 4, dtype=torch.float32, device=DEV)
    out = torch.empty(B * HW, C, dtype=torch.float16, device=DEV)
    ops.groupnorm(x1.to(DEV), gamma.to(DEV), beta.to(DEV), out, B, HW, C1, ws, x2=x2.to(DEV) if C2 else None, C2=C2,
                  eps=eps, silu=silu)
    close(out, ref, what="groupnorm")


@pytest.mark.parametrize("M,C", [(4096, 320), (77, 640), (5, 1280)])
def test_layernorm(M, C):
    x = rnd(M, C, seed=1) * 3 + 1
    g, b = (1 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), out, M, C)
    close(out, F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5), what="layernorm")


@pytest.mark.parametrize("B,heads,Sq,Sk,d", [(1, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160),
                                            (1, 8, 64, 64, 160), (2, 8, 4096, 77, 40), (1, 8, 1024, 77, 80),
                                            (3, 8, 256, 77, 160), (1, 8, 16, 16, 160), (1, 8, 1, 1, 160),
                                            (1, 8, 200, 130, 80), (1, 2, 100, 77, 64)])
def test_attention(B, heads, Sq, Sk, d):
    C = heads * d
    q, k, v = rnd(B * Sq, C, seed=1), rnd(B * Sk, C, seed=2), rnd(B * Sk, C, seed=3)
    qh = q.float().reshape(B, Sq, heads, d).transpose(1, 2)
    kh = k.float().reshape(B, Sk, heads, d).transpose(1, 2)
    vh = v.float().reshape(B, Sk, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * Sq, C)
    out = torch.empty(B * Sq, C, dtype=torch.float16, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), out, B, heads, Sq, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C)
    close(out, ref, rtol=6e-3, what=f"attention B{B} S{Sq}x{Sk} d{d}")


@pytest.mark.parametrize("B,heads,S,d", [(1, 1, 4096, 512), (2, 1, 1024, 512), (1, 1, 3185, 512), (3, 1, 77, 512), (1, 1, 1, 512),
                                         (2, 2, 1000, 512), (2, 1, 64, 512), (1, 1, 33, 512)])
def test_attention_wide_heads(B, heads, S, d):
    """AutoencoderKL mid-block attention shape (one head, d = 512, S = h*w incl. ragged S from odd latent sizes): the
    wide-head flash kernel against torch SDPA in fp32, q|k|v as column slices of one buffer the way the VAE calls it, with a
    dominant late key so the running-max rescale path runs.  A query row's result must not depend on the batch."""
    C = heads * d
    qkv = rnd(B * S, 3 * C, seed=1)
    if S > 40:
        qkv[S - 30, C:2 * C] = qkv[5, :C] * 3          # key S-30 dominates query 5 of image 0
    qh, kh, vh = (qkv[:, i * C:(i + 1) * C].float().reshape(B, S, heads, d).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * S, C)
    t = qkv.to(DEV)
    out = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
    ops.attention(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], out, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
    close(out, ref, rtol=6e-3, what=f"wide attention B{B} S{S} d{d}")
    if B > 1:
        solo = torch.empty(S, C, dtype=torch.float16, device=DEV)
        t1 = t[S:2 * S].contiguous()
        ops.attention(t1[:, :C], t1[:, C:2 * C], t1[:, 2 * C:], solo, 1, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
        assert torch.equal(solo, out[S:2 * S])


def test_attention_peaked_softmax():
    """Forces the online-softmax rescale path: one key dominates late in the sequence."""
    B, heads, S, d = 1, 8, 512, 40
    C = heads * d
    q, k, v = rnd(S, C, seed=1), rnd(S, C, seed=2), rnd(S, C, seed=3)
    k[300] = q[7] * 6   # spike
    k[470] = q[100] * 8
    qh, kh, vh = (t.float().reshape(B, S, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(S, C)
    out = torch.empty(S, C, dtype=torch.float16, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), out, B, heads, S, S, d, ldq=C, ldk=C, ldv=C, ldo=C)
    close(out, ref, rtol=6e-3, what="attention peaked")


@pytest.mark.parametrize("heads,Sq,Sk,d", [(8, 1024, 1024, 40), (8, 256, 77, 80), (4, 64, 64, 160), (10, 300, 300, 64)])
def test_attention_prescaled_q(heads, Sq, Sk, d):
    """scale <= 0: q arrives with d^-0.5 * log2(e) already in it (model.Q_PRESCALE folds it into the to_q weights) -- the
    streaming, the register-staged and the d = 160 kernels against torch SDPA in fp32 on the unscaled q."""
    B, C = 2, heads * d
    q, k, v = rnd(B * Sq, C, seed=1), rnd(B * Sk, C, seed=2), rnd(B * Sk, C, seed=3)
    qh, kh, vh = (t.float().reshape(B, S, heads, d).transpose(1, 2) for t, S in ((q, Sq), (k, Sk), (v, Sk)))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * Sq, C)
    qs = (q.float() * (d ** -0.5 * 1.4426950408889634)).to(torch.float16)
    out = torch.empty(B * Sq, C, dtype=torch.float16, device=DEV)
    ops.attention(qs.to(DEV), k.to(DEV), v.to(DEV), out, B, heads, Sq, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C, scale=0.0)
    close(out, ref, rtol=6e-3, what=f"prescaled attention S{Sq}x{Sk} d{d}")


@pytest.mark.parametrize("B,heads,S,d", [(1, 8, 1088, 40), (2, 8, 1030, 80), (1, 4, 1024, 64), (1, 2, 4096, 40), (1, 8, 2049, 40)])
def test_attention_key_split(B, heads, S, d):
    """1024..4096 keys: the keys of a query block are split over two wave groups and merged (odd tile counts: group 1 idles one
    iteration; ragged last tile in group 1; a dominant key in either half).  Against torch SDPA in fp32, and against the
    unsplit kernel (lcm_set_attention_ksplit(0)) -- same accuracy, different summation order."""
    C = heads * d
    qkv = rnd(B * S, 3 * C, seed=S)
    qkv[S - 30, C:2 * C] = qkv[5, :C] * 3           # late key dominates query 5 (second group)
    qkv[17, C:2 * C] = qkv[S - 9, :C] * 3           # early key dominates a late query (first group)
    qh, kh, vh = (qkv[:, i * C:(i + 1) * C].float().reshape(B, S, heads, d).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * S, C)
    t = qkv.to(DEV)
    outs = []
    try:
        for on in (1, 0):
            ops.set_attention_ksplit(on)
            o = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
            ops.attention(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], o, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        ops.set_attention_ksplit(1)
    close(outs[0], ref, rtol=6e-3, what=f"key-split attention S{S} d{d}")
    close(outs[1], ref, rtol=6e-3, what=f"unsplit attention S{S} d{d}")
    assert not torch.equal(outs[0], outs[1]) or S < 1024


def test_attention_strided_qkv():
    """Fused QKV buffer: q/k/v are column slices of one [B*S, 3C] tensor."""
    B, heads, S, d = 2, 8, 256, 40
    C = heads * d
    qkv = rnd(B * S, 3 * C, seed=1)
    qh, kh, vh = (qkv[:, i * C:(i + 1) * C].float().reshape(B, S, heads, d).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * S, C)
    t = qkv.to(DEV)
    out = torch.empty(B * S, C, dtype=torch.float16, device=DEV)
    ops.attention(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], out, B, heads, S, S, d, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
    close(out, ref, rtol=6e-3, what="attention strided")


def test_softmax_and_transpose():
    x = rnd(300, 4096, seed=1) * 3
    t = x.to(DEV).clone()
    ops.softmax_rows(t, 300, 4096, 4096)
    close(t, torch.softmax(x.float(), -1), rtol=4e-3, atol=1e-5, what="softmax")
    a = rnd(2, 100, 200, seed=2)
    o = torch.empty(2, 200, 100, dtype=torch.float16, device=DEV)
    ops.transpose(a.to(DEV), o, 100, 200, ldi=200, ldo=100, batch=2, stride_in=100 * 200, stride_out=200 * 100)
    assert torch.equal(o.cpu(), a.transpose(1, 2))


@pytest.mark.parametrize("M,N,K,si,so", [(1, 1280, 320, False, True), (8, 20160, 1280, True, False), (3, 320, 256, False, False)])
def test_linear_smallm(M, N, K, si, so):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.linear_smallm(x.to(DEV), w.to(DEV), out, M, N, K, bias=b.to(DEV), res=r.to(DEV), silu_in=si, silu_out=so)
    xin = F.silu(x.float()) if si else x.float()
    ref = F.linear(xin, w.float(), b.float()) + r.float()
    close(out, F.silu(ref) if so else ref, what="linear_smallm")


def test_timestep_embedding_and_step_and_pool():
    from oracle.unet import timestep_sinusoid
    from oracle.scheduler import LCMSchedulerOracle
    from oracle import glue
    out = torch.empty(3, 320, dtype=torch.float16, device=DEV)
    for t in (999, 259, 19):
        ops.timestep_embedding(t, out, 3, 320)
        close(out, timestep_sinusoid(torch.tensor([t] * 3), 320), rtol=1e-3, atol=1.5e-3, what=f"temb t={t}")
    s = LCMSchedulerOracle()
    s.set_timesteps(4)
    B, h, w = 2, 12, 20
    g = torch.Generator().manual_seed(3)
    lat, eps, noise = (torch.randn(B, 4, h, w, generator=g) for _ in range(3))
    for i in range(4):
        ref, _ = s.step(eps, i, lat, noise)
        coef = s.coefficients(i)
        d = lat.to(DEV).clone()
        ops.scheduler_step(eps.permute(0, 2, 3, 1).contiguous().to(DEV), d, noise.to(DEV), coef[:6], coef[6], B, h, w)
        close(d, ref, rtol=1e-5, atol=1e-5, what=f"scheduler step {i}")
    # classifier-free guidance variant
    eu = torch.randn(B, 4, h, w, generator=g)
    ref, _ = s.step(eu + 7.5 * (eps - eu), 1, lat, noise)
    d = lat.to(DEV).clone()
    coef = s.coefficients(1)
    ops.scheduler_step(eps.permute(0, 2, 3, 1).contiguous().to(DEV), d, noise.to(DEV), coef[:6], coef[6], B, h, w,
                       eps_uncond=eu.permute(0, 2, 3, 1).contiguous().to(DEV), guidance=7.5)
    close(d, ref, rtol=1e-5, atol=2e-5, what="scheduler step cfg")
    for hh, ww in ((64, 64), (96, 64), (8, 8), (12, 20)):
        lt = torch.randn(1, 4, hh, ww, generator=g)
        o = torch.empty(1, 4, 8, 8, dtype=torch.float16, device=DEV)
        ops.latents_pool8(lt.to(DEV), o, 1, hh, ww)
        assert o.cpu().numpy().tobytes() == glue.latents_blob(lt.numpy()) or \
            np.abs(o.cpu().float().numpy() - np.frombuffer(glue.latents_blob(lt.numpy()), np.float16).reshape(1, 4, 8, 8).astype(np.float32)).max() < 2e-3


# ----------------------------------------------------------------------------------------------
# Determinism (include/lcm_hip.h): results are a function of the per-image problem only.
# ----------------------------------------------------------------------------------------------
def _gn_of(out, st, B, HW, C):
    gamma, beta = (1 + 0.1 * rnd(C, seed=4).float()).half().to(DEV), rnd(C, seed=5, scale=0.1).to(DEV)
    y = torch.empty_like(out)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C) // 4 + 16, dtype=torch.float32, device=DEV)
    ops.groupnorm_from_stats(out, gamma, beta, y, B, HW, C, st, ws)
    return y


@pytest.mark.parametrize("B,H,W,Cin,Cout,ups", [(1, 32, 32, 320, 640, 0), (2, 16, 16, 640, 640, 0), (1, 24, 24, 320, 320, 0),
                                               (1, 16, 16, 640, 640, 2), (1, 12, 12, 128, 128, 0), (1, 64, 64, 128, 128, 0)])
def test_conv_results_do_not_depend_on_tile_or_variant(B, H, W, Cin, Cout, ups):
    """Every (tile, pipeline variant) the autotuner may pick for a 3x3 convolution gives the same output bits, the same
    statistics slabs (count, layout, values) and hence the same GroupNorm downstream: plans are launch parameters."""
    from sdlcm_amd.packing import pack_conv3x3_up2
    x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
    w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    w = (pack_conv3x3_up2(w4) if ups == 2 else pack3x3(w4)).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    M, K = B * Ho * Wo, (4 if ups == 2 else 9) * Cin
    skws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)
    seen = []
    try:
        for bm in (128, 64):
            for bn in (160, 128, 64):
                if Cout % bn or (bm == 128 and not (W % 16 == 0 or W > 16)):
                    continue
                for var in (1, 2, 3) if bn <= 128 else (1, 2):
                    ops.plan_clear()
                    sp = ops.canonical_splits(2, M // B, Cout, K, Wo << 1, 1 if ups == 2 else 0)
                    ops.plan_set(2, M, Cout, K, Wo << 1, bm, bn, ops.canonical_splits(2, M, Cout, K, Wo << 1, 1 if ups == 2 else 0), var)
                    assert ops.canonical_splits(2, M // B, Cout, K, Wo << 1, 1 if ups == 2 else 0) == sp
                    o = torch.empty(M, Cout, dtype=torch.float16, device=DEV)
                    st = ops.Stats(torch.zeros(ops.stats_floats(M, Cout, Ho * Wo), dtype=torch.float32, device=DEV))
                    ops.conv3x3(x, w, o, B, H, W, Cin, Cout, bias=b, ups=ups, stats=st)
                    assert st.P > 0
                    seen.append(((bm, bn, var), o, st.P, st.buf[:B * st.P * Cout * 2].clone(), _gn_of(o, st, B, Ho * Wo, Cout)))
        torch.cuda.synchronize()
    finally:
        ops.plan_reset()
        ops.set_workspace(None)
    assert len(seen) >= 3
    for tag, o, P, sb, y in seen[1:]:
        assert P == seen[0][2], (tag, P, seen[0][2])
        assert torch.equal(o, seen[0][1]), f"output differs under plan {tag}"
        assert torch.equal(sb, seen[0][3]), f"statistics slabs differ under plan {tag}"
        assert torch.equal(y, seen[0][4]), f"GroupNorm differs under plan {tag}"


@pytest.mark.parametrize("M,N,K,hw", [(4096, 320, 320, 4096), (2048, 640, 1280, 1024), (512, 1280, 1280, 256), (1152, 320, 640, 576)])
def test_gemm_results_do_not_depend_on_tile_or_variant(M, N, K, hw):
    a, w, b = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5).to(DEV), rnd(N, seed=3).to(DEV)
    res = rnd(M, N, seed=4).to(DEV)
    skws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)
    seen = []
    try:
        for bm in (128, 64):
            for bn in (160, 128, 64):
                if N % bn:
                    continue
                for var in (-1, 0, 1, 2, 4):
                    ops.plan_clear()
                    ops.plan_set(0, M, N, K, 1, bm, bn, ops.canonical_splits(0, M, N, K), var)
                    o = torch.empty(M, N, dtype=torch.float16, device=DEV)
                    st = ops.Stats(torch.zeros(ops.stats_floats(M, N, hw), dtype=torch.float32, device=DEV))
                    ops.gemm(a, w, o, bias=b, res=res, stats=st, img_rows=hw)
                    assert st.P == hw // 32 or ops.canonical_splits(0, hw, N, K) > 1
                    seen.append(((bm, bn, var), o, st.P, st.buf[:(M // hw) * st.P * N * 2].clone(), _gn_of(o, st, M // hw, hw, N)))
        torch.cuda.synchronize()
    finally:
        ops.plan_reset()
        ops.set_workspace(None)
    for tag, o, P, sb, y in seen[1:]:
        assert P == seen[0][2]
        assert torch.equal(o, seen[0][1]), f"output differs under plan {tag}"
        assert torch.equal(sb, seen[0][3]), f"statistics slabs differ under plan {tag}"
        assert torch.equal(y, seen[0][4]), f"GroupNorm differs under plan {tag}"
    close(seen[0][1], F.linear(a.float().cpu(), w.float().cpu(), b.float().cpu()) + res.float().cpu(), what="gemm under plans")


@pytest.mark.parametrize("H,W,Cin,Cout,stride,ups", [(8, 8, 1280, 1280, 1, 0), (16, 16, 640, 1280, 1, 0), (32, 32, 320, 640, 1, 0),
                                                   (16, 16, 1280, 1280, 1, 2), (32, 32, 320, 320, 2, 0), (12, 20, 128, 128, 1, 0)])
def test_contractions_are_batch_invariant(H, W, Cin, Cout, stride, ups):
    """Image i of a batch of 3 gets exactly the bits it gets alone: the K partition (split-K included), the reduce slabs
    and the statistics slabs are keyed on the per-image shape.  Covers halo conv (split and unsplit), the phase-decomposed
    upsample conv, the stride-2 row-gather conv, and a GEMM with fused statistics on the conv output."""
    from sdlcm_amd.packing import pack_conv3x3_up2
    B = 3
    xs = rnd(B, Cin, H, W, seed=1)
    w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    w = (pack_conv3x3_up2(w4) if ups == 2 else pack3x3(w4)).to(DEV)
    wg = rnd(Cout, Cout, seed=6, scale=Cout ** -0.5).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    Ho, Wo = (2 * H, 2 * W) if ups else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    HW = Ho * Wo
    skws = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)

    def run(x4):
        n = x4.shape[0]
        x = to_nhwc(x4).to(DEV)
        o = torch.empty(n * HW, Cout, dtype=torch.float16, device=DEV)
        st = ops.Stats(torch.zeros(ops.stats_floats(n * HW, Cout, HW), dtype=torch.float32, device=DEV))
        ops.conv3x3(x, w, o, n, H, W, Cin, Cout, bias=b, stride=stride, ups=ups, stats=st)
        y = _gn_of(o, st, n, HW, Cout) if st.P > 0 else o
        o2 = torch.empty_like(o)
        st2 = ops.Stats(torch.zeros(ops.stats_floats(n * HW, Cout, HW), dtype=torch.float32, device=DEV))
        ops.gemm(y, wg, o2, res=o, stats=st2, img_rows=HW)
        y2 = _gn_of(o2, st2, n, HW, Cout) if st2.P > 0 else o2
        torch.cuda.synchronize()
        return o.reshape(n, HW, Cout), st.P, y2.reshape(n, HW, Cout)
    try:
        ob, Pb, yb = run(xs)
        for i in range(B):
            o1, P1, y1 = run(xs[i:i + 1])
            assert P1 == Pb
            assert torch.equal(o1[0], ob[i]), f"conv output of image {i} depends on the batch"
            assert torch.equal(y1[0], yb[i]), f"conv -> GroupNorm -> gemm -> GroupNorm of image {i} depends on the batch"
    finally:
        ops.set_workspace(None)


def test_split_workspace_too_small_fails_loudly():
    """A registered workspace that cannot hold the canonical partition is an error, never a silently different split."""
    from sdlcm_amd.lib import LcmHipError
    M, N, K = 64, 1280, 5120
    assert ops.canonical_splits(0, M, N, K) > 1
    a, w = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5).to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    tiny = torch.empty(1024, dtype=torch.float32, device=DEV)
    ops.set_workspace(tiny)
    try:
        with pytest.raises(LcmHipError, match="workspace too small"):
            ops.gemm(a, w, out)
    finally:
        ops.set_workspace(None)


# ----------------------------------------------------------------------------------------------
# Request sizes that are multiples of 8 but not of 64 (backends/rknnlcm.py:380-381; the UI ships 640x360).
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,oh,ow", [(1, 7, 12, 128, 128, 13, 24), (2, 13, 9, 64, 128, 25, 17), (1, 12, 23, 128, 64, 23, 45),
                                                (1, 7, 12, 128, 128, 13, 23), (1, 25, 33, 320, 320, 49, 65), (1, 7, 9, 1280, 1280, 13, 17)])
def test_conv_upsample_to_odd_skip_size(B, H, W, Cin, Cout, oh, ow):
    """Upsample2D with output_size = an odd-sized skip (2H-1 / 2W-1): F.interpolate(size=..., mode="nearest") -> conv3x3.
    The conv pads the cropped upsampled image with zeros, so the border outputs see fewer taps than the pre-summed phase
    weights hold: the loader-fused form (ups=1, plain 3x3 weights) serves it, the phase form must refuse.  Fused
    statistics of the output included (split-K reduce slabs of an odd-sized image in the 1280-channel case)."""
    from sdlcm_amd.lib import LcmHipError
    x4 = rnd(B, Cin, H, W, seed=1)
    w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    ref = F.conv2d(F.interpolate(x4.float(), size=(oh, ow), mode="nearest"), w4.float(), b.float(), padding=1)
    wk = pack3x3(w4).to(DEV)
    o = torch.empty(B * oh * ow, Cout, dtype=torch.float16, device=DEV)
    st = ops.Stats(torch.zeros(ops.stats_floats(B * oh * ow, Cout, oh * ow), dtype=torch.float32, device=DEV))
    skws = torch.empty(16 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)
    try:
        with pytest.raises(LcmHipError):
            ops.conv3x3(to_nhwc(x4).to(DEV), wk, o, B, H, W, Cin, Cout, bias=b.to(DEV), ups=2, out_hw=(oh, ow))
        ops.conv3x3(to_nhwc(x4).to(DEV), wk, o, B, H, W, Cin, Cout, bias=b.to(DEV), ups=1, stats=st, out_hw=(oh, ow))
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    close(from_nhwc(o, B, oh, ow), ref, what="upsample conv to odd size")
    assert st.P > 0
    gamma, beta = (1 + 0.1 * rnd(Cout, seed=4).float()).half(), rnd(Cout, seed=5, scale=0.1)
    y = _gn_of(o, st, B, oh * ow, Cout)
    close(y, _ref_gn(o.cpu(), B, oh * ow, Cout, gamma, beta, 1e-5, True), what="gn from the statistics of a cropped upsample conv")


@pytest.mark.parametrize("rows,n,ld", [(37, 100, 128), (5, 3185, 3200), (64, 64, 64), (9, 7, 64)])
def test_softmax_ragged_rows_zero_padding(rows, n, ld):
    x = (rnd(rows, ld, seed=1) * 3).to(DEV)
    ref = torch.softmax(x[:, :n].float().cpu(), dim=1)
    ops.softmax_rows(x, rows, n, ld)
    torch.cuda.synchronize()
    got = x.float().cpu()
    assert (got[:, n:] == 0).all()
    assert (got[:, :n] - ref).abs().max() < 2e-3


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(1, 16, 16, 1280, 640, 1280), (2, 32, 32, 640, 320, 640), (1, 64, 64, 128, 0, 128),
                                             (1, 24, 24, 320, 320, 320)])
def test_gn_fused_conv_is_bit_identical_to_apply_then_conv(B, H, W, C1, C2, Cout):
    """GroupNorm+SiLU applied while the conv stages its halo (chosen from the TOTAL tensor size, so from the batch) must give
    the bits of the separate apply pass followed by the plain conv: same affine, same fp16 rounding of the normalised
    value, same K partition (split-K included: the 16x16 and 32x32 cases split)."""
    C = C1 + C2
    x1 = to_nhwc(rnd(B, C1, H, W, seed=1) * 1.5 + 0.3).to(DEV)
    x2 = to_nhwc(rnd(B, C2, H, W, seed=2) * 0.7 - 0.5).to(DEV) if C2 else None
    gamma, beta = (1 + 0.1 * rnd(C, seed=3).float()).half().to(DEV), rnd(C, seed=4, scale=0.1).to(DEV)
    w = pack3x3(rnd(Cout, C, 3, 3, seed=5, scale=(9 * C) ** -0.5)).to(DEV)
    b, radd = rnd(Cout, seed=6).to(DEV), rnd(B, Cout, seed=7).to(DEV)
    HW = H * W
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C) // 4 + 16, dtype=torch.float32, device=DEV)
    scale = torch.empty(B, C, dtype=torch.float32, device=DEV)
    shift = torch.empty(B, C, dtype=torch.float32, device=DEV)
    skws = torch.empty(32 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)
    try:
        ops.groupnorm_affine(x1, gamma, beta, scale, shift, B, HW, C1, ws, x2=x2, C2=C2)
        o_fused = torch.empty(B * HW, Cout, dtype=torch.float16, device=DEV)
        st_f = ops.Stats(torch.zeros(ops.stats_floats(B * HW, Cout, HW), dtype=torch.float32, device=DEV))
        ops.conv3x3_gn(x1, w, o_fused, B, H, W, C1, Cout, x2=x2, C2=C2, gn_scale=scale, gn_shift=shift, silu=True, bias=b,
                       rowadd=radd, stats=st_f)
        hn = torch.empty(B * HW, C, dtype=torch.float16, device=DEV)
        ops.groupnorm(x1, gamma, beta, hn, B, HW, C1, ws, x2=x2, C2=C2)
        o_plain = torch.empty_like(o_fused)
        st_p = ops.Stats(torch.zeros(ops.stats_floats(B * HW, Cout, HW), dtype=torch.float32, device=DEV))
        ops.conv3x3(hn, w, o_plain, B, H, W, C, Cout, bias=b, rowadd=radd, stats=st_p)
        torch.cuda.synchronize()
    finally:
        ops.set_workspace(None)
    assert torch.equal(o_fused, o_plain)
    assert st_f.P == st_p.P and torch.equal(st_f.buf[:B * st_f.P * Cout * 2], st_p.buf[:B * st_p.P * Cout * 2])


@pytest.mark.parametrize("M,C,N,geglu,bias", [(4096, 320, 960, False, False), (1024, 640, 640, False, False), (256, 1280, 10240, True, True),
                                             (77, 320, 2560, True, True), (3185, 320, 320, False, True)])
def test_gemm_with_folded_layernorm(M, C, N, geglu, bias):
    """LayerNorm -> Linear (-> GEGLU) as one contraction (lcm_gemm_ln_f16) against F.layer_norm + F.linear (+ x * gelu(gate)),
    on rows with a large common offset (mean / std ~ 3: the cancellation the fold has to survive); every tile / variant the
    plan table may choose gives the same bits."""
    from sdlcm_amd.packing import pack_geglu
    x = (rnd(M, C, seed=1).float() * 0.7 + 2.0 + rnd(M, 1, seed=8).float()).half()
    gamma, beta = (1 + 0.2 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.2)
    W = rnd(N, C, seed=4, scale=C ** -0.5)
    b = rnd(N, seed=5, scale=0.3) if bias else None
    ref = F.linear(F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5), W.float(), b.float() if bias else None)
    Wg = W.float() * gamma.float()[None, :]
    c = W.float() @ beta.float() + (b.float() if bias else 0.0)
    if geglu:
        from sdlcm_amd.packing import geglu_col_order
        val, gate = ref.chunk(2, dim=1)
        ref = (val * F.gelu(gate))[:, geglu_col_order(N // 2)]             # the GEGLU output is stored in operand order
        Wg, c = pack_geglu(Wg, c)
    Wh = Wg.half()
    g = Wh.float().sum(1)
    outs = []
    try:
        for bm, bn, var in ((128, 128, -1), (64, 64, 4), (128, 64, 2), (64, 128, 0), (128, 160, 1), (64, 64, 1), (64, 128, 2), (128, 128, 0),
                            (64, 64, 0), (128, 64, 3)):
            if N % bn or (bn == 160 and geglu) or (bm == 128 and M < 128):
                continue
            ops.plan_clear()
            ops.plan_set(0, M, N, C, 1, bm, bn, 1, var)
            o = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=DEV)
            ops.gemm_ln(x.to(DEV), Wh.to(DEV), g.to(DEV), c.to(DEV), o, epilogue=1 if geglu else 0)
            outs.append(((bm, bn, var), o))
        torch.cuda.synchronize()
    finally:
        ops.plan_reset()
    close(outs[0][1], ref, rtol=6e-3, what="gemm with folded LayerNorm")
    for tag, o in outs[1:]:
        assert torch.equal(o, outs[0][1]), f"folded-LayerNorm gemm differs under plan {tag}"
    # refresh kernel: g from the live fp16 weight, c = c_base + alpha * c_delta
    g2 = torch.zeros(N, dtype=torch.float32, device=DEV)
    c2 = torch.zeros(N, dtype=torch.float32, device=DEV)
    dc = rnd(N, seed=9).float().to(DEV)
    ops.ln_fold_refresh(Wh.to(DEV), g2, c.to(DEV), dc, 0.5, c2)
    torch.cuda.synchronize()
    assert (g2.cpu() - g).abs().max() < 1e-4 * max(1.0, g.abs().max().item())
    assert torch.allclose(c2.cpu(), c + 0.5 * dc.cpu(), atol=1e-6)


@pytest.mark.parametrize("kind,B,H,W,Cin,Cout,stride,ups", [("conv", 2, 16, 16, 1280, 1280, 1, 0), ("conv", 1, 32, 32, 640, 640, 1, 0),
                                                          ("conv", 2, 16, 16, 1280, 1280, 1, 2), ("conv", 1, 12, 20, 512, 128, 1, 0),
                                                          ("conv", 1, 32, 32, 320, 640, 2, 0), ("gemm", 2, 16, 16, 5120, 1280, 1, 0),
                                                          ("gemm", 1, 32, 32, 2560, 640, 1, 0)])
def test_segmented_accumulation_is_bit_identical_to_split(kind, B, H, W, Cin, Cout, stride, ups):
    """A layer whose canonical K partition has several parts runs either split over workgroups (fp32 slabs + the reduce
    launch) or segmented in one workgroup (parts accumulated separately, added in part order in registers): outputs,
    statistics slabs and the GroupNorm downstream must be the same bits, with bias / row-add / residual epilogues."""
    from sdlcm_amd.packing import pack_conv3x3_up2
    Ho, Wo = (2 * H, 2 * W) if ups else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    HW = Ho * Wo
    b, radd = rnd(Cout, seed=3).to(DEV), rnd(B, Cout, seed=7).to(DEV)
    res = rnd(B * HW, Cout, seed=8).to(DEV)
    if kind == "conv":
        x = to_nhwc(rnd(B, Cin, H, W, seed=1)).to(DEV)
        w4 = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
        w = (pack_conv3x3_up2(w4) if ups == 2 else pack3x3(w4)).to(DEV)
        K = (4 if ups == 2 else 9) * Cin
        nparts = ops.canonical_splits(1 if stride == 2 else 2, HW, Cout, K, (Wo << 1) if stride == 1 else 1, 1 if ups == 2 else 0)
    else:
        x = rnd(B * HW, Cin, seed=1).to(DEV)
        w = rnd(Cout, Cin, seed=2, scale=Cin ** -0.5).to(DEV)
        nparts = ops.canonical_splits(0, HW, Cout, Cin)
    skws = torch.empty(32 << 20, dtype=torch.float32, device=DEV)
    ops.set_workspace(skws)
    got = []
    try:
        assert nparts > 1, "pick a shape whose canonical partition has parts"
        for mode in (2, 1):
            ops.set_seg_mode(mode)
            o = torch.empty(B * HW, Cout, dtype=torch.float16, device=DEV)
            st = ops.Stats(torch.zeros(ops.stats_floats(B * HW, Cout, HW), dtype=torch.float32, device=DEV))
            if kind == "conv":
                ops.conv3x3(x, w, o, B, H, W, Cin, Cout, bias=b, rowadd=radd, res=res, stride=stride, ups=ups, stats=st)
            else:
                ops.gemm(x, w, o, bias=b, rowadd=radd, rows_per_batch=HW, res=res, stats=st, img_rows=HW)
            assert st.P > 0
            got.append((o, st.P, st.buf[:B * st.P * Cout * 2].clone(), _gn_of(o, st, B, HW, Cout)))
        torch.cuda.synchronize()
    finally:
        ops.set_seg_mode(0)
        ops.set_workspace(None)
    assert got[0][1] == got[1][1]
    assert torch.equal(got[0][0], got[1][0]), "segmented output differs from split + reduce"
    assert torch.equal(got[0][2], got[1][2]), "statistics slabs differ"
    assert torch.equal(got[0][3], got[1][3])


@pytest.mark.parametrize("B,heads,Sq,Sk,d,causal", [(1, 8, 4096, 4096, 40, False), (2, 8, 1024, 1024, 80, False), (1, 8, 3185, 3185, 40, False),
                                                  (1, 20, 1024, 1024, 64, False), (1, 8, 1000, 1000, 80, False),
                                                  (1, 8, 256, 77, 160, False), (2, 12, 77, 77, 64, True)])
def test_attention_workgroup_shape_is_bit_neutral(B, heads, Sq, Sk, d, causal):
    """Workgroup shapes of one attention kernel (register-staged kernel: 64 / 128 query rows; streaming kernel of the long
    non-causal sequences: 128 / 256): a query row's online softmax walks the same K/V tiles in the same order either way, so
    the choice (made from the grid size, i.e. from the batch) never changes a bit."""
    C = heads * d
    q, k, v = (rnd(B * S, C, seed=i + 1).to(DEV) for i, S in enumerate((Sq, Sk, Sk)))
    outs = []
    streaming = (not causal) and Sk >= 128 and d in (40, 64, 80)
    try:
        for waves in ((4, 8, 0) if streaming else (4, 2, 0)):
            ops.set_attention_waves(waves)
            o = torch.empty(B * Sq, C, dtype=torch.float16, device=DEV)
            ops.attention(q, k, v, o, B, heads, Sq, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C, causal=causal)
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        ops.set_attention_waves(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    qf, kf, vf = (t.float().cpu().reshape(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal).transpose(1, 2).reshape(B * Sq, C)
    close(outs[0], ref, what="attention")


@pytest.mark.parametrize("H,W,Cin,Cout", [(25, 45, 64, 128), (49, 65, 128, 64), (13, 13, 320, 320)])
def test_statistics_buffer_exact_size_and_guard(H, W, Cin, Cout):
    """Odd image sizes (partial slabs at the border -- the case that overflowed the buffer in round 2): with a statistics buffer
    of EXACTLY slabs x channels x 2 floats the launch succeeds and never writes past it (guard words behind it stay intact);
    one float less is refused with LCM_EINVAL before anything runs."""
    from sdlcm_amd.lib import LcmHipError
    x = rnd(H * W, Cin, seed=1).to(DEV)
    w = (rnd(Cout, 9 * Cin, seed=2) * (9 * Cin) ** -0.5).to(DEV)
    out = torch.empty(H * W, Cout, dtype=torch.float16, device=DEV)
    ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    try:
        big = ops.Stats(torch.zeros(ops.stats_floats(H * W, Cout, H * W), dtype=torch.float32, device=DEV))
        ops.conv3x3(x, w, out, 1, H, W, Cin, Cout, stats=big)
        assert big.P > 0
        n = big.P * Cout * 2
        guard = 4096
        raw = torch.full((n + guard,), 12345.0, dtype=torch.float32, device=DEV)
        exact = ops.Stats(raw[:n])
        ops.conv3x3(x, w, out, 1, H, W, Cin, Cout, stats=exact)
        torch.cuda.synchronize()
        assert exact.P == big.P
        assert torch.equal(raw[:n], big.buf[:n])
        assert bool((raw[n:] == 12345.0).all()), "statistics written past the buffer"
        small = ops.Stats(raw[:n - 1])
        with pytest.raises(LcmHipError, match="statistics buffer too small"):
            ops.conv3x3(x, w, out, 1, H, W, Cin, Cout, stats=small)
    finally:
        ops.set_workspace(None)


@pytest.mark.gpu
def test_lane_streams_are_never_recycled_handles():
    """torch.cuda.Stream() hands out 32 pooled handles per device round-robin, so long-lived owners collide after ~16
    lanes.  Lanes take streams from lcm_stream_create instead: distinct while held, reused only after release; and the library
    refuses a second workspace owner on a live handle."""
    import torch
    from sdlcm_amd import ops
    from sdlcm_amd.lib import LcmHipError
    held = [ops.acquire_stream("cuda:0") for _ in range(40)]
    assert len({s.cuda_stream for s in held}) == 40
    pooled = {torch.cuda.Stream(device="cuda:0").cuda_stream for _ in range(64)}
    assert len(pooled) <= 32 and not (pooled & {s.cuda_stream for s in held})
    a, b = torch.empty(1 << 18, device="cuda:0"), torch.empty(1 << 18, device="cuda:0")
    ops.set_stream_workspace(held[0], a)
    with pytest.raises(LcmHipError, match="another owner"):
        ops.set_stream_workspace(held[0], b)
    ops.set_stream_workspace(held[0], b, forget=True)           # not b's entry: stays
    with pytest.raises(LcmHipError, match="another owner"):
        ops.set_stream_workspace(held[0], b)
    ops.set_stream_workspace(held[0], a, forget=True)
    ops.set_stream_workspace(held[0], b)
    ops.set_stream_workspace(held[0], b, forget=True)
    # work runs on such a stream like on any other
    with torch.cuda.stream(held[1]):
        x = torch.ones(1024, device="cuda:0") * 3
    held[1].synchronize()
    assert float(x.sum()) == 3072.0
    handles = {s.cuda_stream for s in held}
    for s in held:
        ops.release_stream(s)
    again = [ops.acquire_stream("cuda:0") for _ in range(40)]
    assert {s.cuda_stream for s in again} == handles
    for s in again:
        ops.release_stream(s)


@pytest.mark.parametrize("M,img", [(32768, 4096), (24576 + 77, 0), (36864, 9216)])
def test_fused_mlp_is_bit_identical_to_two_launches(M, img):
    """norm3 -> ff.net.0 -> GEGLU -> ff.net.2 -> + h as ONE kernel (lcm_mlp_geglu_f16, csrc/mlp_fused.hip) against the two launches
    it replaces (lcm_gemm_ln_f16 with the GEGLU epilogue, lcm_gemm_f16 with bias + residual): the same bits -- the kernel is a
    launch parameter chosen from the row count, like a tile shape -- and both against the torch fp32 reference of
    h + Linear(GEGLU(Linear(LayerNorm(h)))).  Ragged last workgroup (M % 128 != 0), in-place output, rows with a common offset."""
    from sdlcm_amd.packing import pack_ff2_cols, pack_geglu
    C, Fh = 320, 1280
    x = (rnd(M, C, seed=1).float() * 0.7 + 0.5 + rnd(M, 1, seed=8).float()).half()
    gamma, beta = (1 + 0.2 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.2)
    W1, b1 = rnd(2 * Fh, C, seed=4, scale=C ** -0.5), rnd(2 * Fh, seed=5, scale=0.3)
    W2, b2 = rnd(C, Fh, seed=6, scale=Fh ** -0.5), rnd(C, seed=7, scale=0.3)
    Wg = W1.float() * gamma.float()[None, :]
    c = W1.float() @ beta.float() + b1.float()
    Wg, c = pack_geglu(Wg, c)
    W1h = Wg.half().to(DEV)
    g = torch.zeros(2 * Fh, dtype=torch.float32, device=DEV)
    ops.ln_fold_refresh(W1h, g)
    c, W2p, b2d = c.to(DEV), pack_ff2_cols(W2).to(DEV), b2.to(DEV)
    # two launches
    h2 = x.to(DEV).clone()
    ff = torch.empty(M, Fh, dtype=torch.float16, device=DEV)
    ops.gemm_ln(h2, W1h, g, c, ff, epilogue=1, img_rows=img)
    ops.gemm(ff, W2p, h2, bias=b2d, res=h2, img_rows=img)
    # one launch, in place
    h1 = x.to(DEV).clone()
    guard = torch.full((4096,), 7.0, dtype=torch.float16, device=DEV)
    assert ops.mlp_fused_applies(M, C, img)
    ops.mlp_geglu(h1, W1h, g, c, W2p, b2d, h1, img_rows=img)
    torch.cuda.synchronize()
    assert torch.equal(h1, h2), f"fused FeedForward differs from the two launches: max |d| = {(h1.float() - h2.float()).abs().max().item():.3g}"
    assert bool((guard == 7.0).all())
    # against torch fp32 on a sample of rows (first / last workgroups, the ragged tail)
    rows = torch.cat([torch.arange(0, 256), torch.arange(M // 2, M // 2 + 128), torch.arange(M - 200, M)])
    xr = x[rows].float()
    v, gt = F.linear(F.layer_norm(xr, (C,), gamma.float(), beta.float(), 1e-5), W1.float(), b1.float()).chunk(2, dim=1)
    ref = xr + F.linear((v * F.gelu(gt)).half().float(), W2.float(), b2.float())
    close(h1[rows.to(DEV)], ref, rtol=6e-3, what="fused FeedForward")


def test_fused_mlp_refuses_what_it_cannot_reproduce():
    """Layers whose canonical K partition of ff.net.2 has parts (small images), and widths other than 320, take the two launches:
    the selector says no and the C ABI refuses with a message instead of computing something else."""
    from sdlcm_amd.lib import LcmHipError
    assert not ops.mlp_fused_applies(32768, 640, 1024)
    assert not ops.mlp_fused_applies(1024, 320, 1024)              # too few rows
    parts = ops.canonical_splits(0, 256, 320, 1280, 1, 0)
    assert parts > 1 and not ops.mlp_fused_applies(32768, 320, 256)
    x = torch.zeros(32768, 320, dtype=torch.float16, device=DEV)
    w1, w2 = torch.zeros(2560, 320, dtype=torch.float16, device=DEV), torch.zeros(320, 1280, dtype=torch.float16, device=DEV)
    z = torch.zeros(2560, dtype=torch.float32, device=DEV)
    with pytest.raises(LcmHipError, match="canonical K partition"):
        ops.mlp_geglu(x, w1, z, z, w2, None, x, img_rows=256)


@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [(2, 64, 64, 128, 128, "gn"), (1, 40, 24, 128, 128, "plain"), (2, 33, 20, 64, 320, "plain"),
                                                (1, 16, 16, 320, 640, "gn"), (2, 12, 12, 128, 64, "ups"), (1, 8, 40, 64, 320, "plain160"),
                                                (2, 16, 24, 128, 320, "plain160"), (2, 16, 16, 1280, 1280, "seg"), (2, 32, 32, 640, 640, "seg")])
def test_staged_epilogue_is_bit_identical(B, H, W, Cin, Cout, mode):
    """The halo conv's plain launches move the residual / result tile through an LDS image of the tile in whole rows
    (tile_epilogue_staged) instead of 8-byte pieces per lane: data movement only -- output and fused GroupNorm statistics must be
    the same bits, with and without residual, on ragged images (tiles hanging over the border), 64 / 128 / 160-wide tiles, the
    GroupNorm-fused staging and the phase-decomposed upsample."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B * H * W, Cin, generator=g).half().to(DEV)
    ups = 2 if mode == "ups" else 0
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    from sdlcm_amd.packing import pack_conv3x3, pack_conv3x3_up2
    w = (pack_conv3x3_up2(wt.half()) if ups else pack_conv3x3(wt.half())).to(DEV)
    bias = torch.randn(Cout, generator=g).half().to(DEV)
    res = torch.randn(B * Ho * Wo, Cout, generator=g).half().to(DEV)
    sc = (torch.rand(B, Cin, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(B, Cin, generator=g) * 0.1).to(DEV)
    outs = {}
    ops.set_halo_pipe_threshold(0)          # small grids too take the single-buffer kernel (the one that has the staged form)
    if mode == "plain160":                  # the 160-wide tile (20 chunks per pixel row: the rotation is not a power of two)
        ops.plan_set(2, B * H * W, Cout, 9 * Cin, W << 1, 128 if H * W >= 256 and W >= 16 else 64, 160, 1, 1)
        mode = "plain"
    if mode == "seg":                       # segmented accumulation (the batched form of a layer whose K partition has parts): pipe kernel
        assert ops.canonical_splits(2, H * W, Cout, 9 * Cin, W << 1, 0) > 1
        ops.set_seg_mode(1)
        mode = "plain"
    try:
        for staged in (0, 1):
            ops.set_staged_epilogue(staged)
            for with_res in (False, True):
                o = torch.full((B * Ho * Wo + 64, Cout), 7.0, dtype=torch.float16, device=DEV)      # guard rows behind the output
                st = ops.Stats(torch.zeros(ops.stats_floats(B * Ho * Wo, Cout, Ho * Wo), dtype=torch.float32, device=DEV))
                kw = dict(bias=bias, stats=st, res=res if with_res else None)
                if mode == "gn":
                    ops.conv3x3_gn(x, w, o[:B * Ho * Wo], B, H, W, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True, **kw)
                else:
                    ops.conv3x3(x, w, o[:B * Ho * Wo], B, H, W, Cin, Cout, ups=ups, **kw)
                torch.cuda.synchronize()
                assert bool((o[B * Ho * Wo:] == 7.0).all()), "wrote past the output"
                outs[(staged, with_res)] = (o[:B * Ho * Wo].clone(), st.buf.clone(), st.P)
    finally:
        ops.set_staged_epilogue(1)
        ops.set_halo_pipe_threshold(768)
        ops.set_seg_mode(0)
        ops.plan_reset()
    for with_res in (False, True):
        a, b = outs[(0, with_res)], outs[(1, with_res)]
        assert a[2] == b[2]
        assert torch.equal(a[0], b[0]), f"staged epilogue changes the output (residual {with_res}): max |d| {(a[0].float() - b[0].float()).abs().max().item():.3g}"
        assert torch.equal(a[1], b[1]), "staged epilogue changes the fused statistics"
    assert not torch.equal(outs[(1, False)][0], outs[(1, True)][0])


@pytest.mark.parametrize("M,N,K,plan", [(4096, 320, 320, (128, 160, 2)), (1000, 640, 640, (64, 160, 3)), (300, 320, 1280, (64, 64, 4)),
                                        (4096, 1280, 1280, (128, 128, 2)), (520, 320, 320, (128, 64, 2))])
def test_staged_epilogue_of_plain_gemms_is_bit_identical(M, N, K, plan):
    """Plain GEMMs with a residual (attn.to_out, ff.net.2, proj_out) take the staged epilogue too (igemm2_kernel<..., EPI = 1>):
    same output and same fused statistics as the per-lane form, ragged last tile, in-place residual (out aliases res)."""
    a = rnd(M, K, seed=1).to(DEV)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(DEV)
    b = rnd(N, seed=3).to(DEV)
    r = rnd(M, N, seed=4).to(DEV)
    outs = []
    try:
        ops.plan_set(0, M, N, K, 1, plan[0], plan[1], 1, plan[2])
        for staged in (0, 3):                 # bit 1 = the GEMM side of the switch (off by default: measured neutral)
            ops.set_staged_epilogue(staged)
            h = torch.full((M + 16, N), 7.0, dtype=torch.float16, device=DEV)
            h[:M].copy_(r)
            st = ops.Stats(torch.zeros(ops.stats_floats(M, N, M), dtype=torch.float32, device=DEV)) if M % 32 == 0 else None
            ops.gemm(a, w, h[:M], bias=b, res=h[:M], stats=st, img_rows=M)          # in place, as the transformer blocks call it
            torch.cuda.synchronize()
            assert bool((h[M:] == 7.0).all())
            outs.append((h[:M].clone(), st.buf.clone() if st is not None else None, st.P if st is not None else 0))
    finally:
        ops.set_staged_epilogue(1)
        ops.plan_reset()
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])
    close(outs[1][0], F.linear(a.float().cpu(), w.float().cpu(), b.float().cpu()) + r.float().cpu(), what="gemm + residual (staged)")
