"""Thin torch-tensor front end of the C ABI: pointers + sizes in, kernels enqueued on the current
torch HIP stream.  torch is plumbing here (device memory, streams); all arithmetic is in liblcmhip.so.
Activations are pixel-major fp16 ``[B*H*W, C]`` tensors."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Launch hooks.  Both are THREAD-LOCAL: the lanes of a pipeline run on threads of their own (backends/hip_worker.py), and a
# process-global hook let the eager launches of another lane (its prompt encoder runs outside the plan lock) land in the
# tuning lane's record list -- the autotuner then replayed them on live buffers of a request in flight.
#   recording():  every contraction / attention launch of THIS thread appends (plan key, metadata, replay closure)
#   profiling():  every such launch appends the record of its algorithmic work (bench.py's roofline leg pairs them, in
#                 launch order, with the library's event brackets: lcm_profile_begin / lcm_profile_end)
import contextlib
import threading

_hooks = threading.local()


def _record_list():
    return getattr(_hooks, "record", None)


def _profile_list():
    return getattr(_hooks, "profile", None)


@contextlib.contextmanager
def recording():
    """``with ops.recording() as recs:`` -- launches issued by this thread inside the block are appended to ``recs``."""
    prev, recs = _record_list(), []
    _hooks.record = recs
    try:
        yield recs
    finally:
        _hooks.record = prev


@contextlib.contextmanager
def profiling():
    prev, recs = _profile_list(), []
    _hooks.profile = recs
    try:
        yield recs
    finally:
        _hooks.profile = prev


class _Timed:
    """Launch-order record of the algorithmic work of each MFMA launch; durations come from the library's own
    event brackets (lcm_profile_begin / lcm_profile_end), matched 1:1 by launch order."""

    def __init__(self, kind, tile, flops, bytes_):
        self.rec = None
        if _profile_list() is not None:
            self.rec = dict(kind=kind, flops=float(flops), bytes=float(bytes_))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.rec is not None:
            pl = _profile_list()
            if pl is not None:
                pl.append(self.rec)
        return False


def plan_set(kind, M, N, K, aux, bm, bn, splits, variant=-1):
    _lib.check(_lib.load().lcm_plan_set(kind, M, N, K, aux, bm, bn, splits, variant), "lcm_plan_set")


def plan_clear():
    _lib.check(_lib.load().lcm_plan_clear(), "lcm_plan_clear")


def plan_reset():
    """Back to the table the library is loaded with (shipped plans + LCM_TUNE_CACHE)."""
    plan_clear()
    _lib._install_plans(_lib.load())


def canonical_splits(kind, m_img, N, K, aux=1, ph=0):
    """The K partition the library runs a contraction of this PER-IMAGE shape with (include/lcm_hip.h, Determinism)."""
    n = _lib.load().lcm_canonical_splits(int(kind), int(m_img), int(N), int(K), int(aux), int(ph))
    if n < 0:
        _lib.check(n, "lcm_canonical_splits")
    return n


def tile_config(M, N, batch=1):
    c = _lib.load().lcm_gemm_tile_config(int(M), int(N), int(batch))
    return f"{c // 1000}x{c % 1000}"


class Stats:
    """Producer-written GroupNorm partial statistics of one tensor: fp32 [B*P][C][2] + the slab count P."""

    __slots__ = ("buf", "P")

    def __init__(self, buf):
        self.buf = buf
        self.P = 0


def stats_floats(M, N, hw=None):
    """fp32 elements that always suffice for the fused statistics of an [M, N] output made of M // hw images of hw rows
    (hw None: one image): lcm_stats_bytes of the library (slab counts are per image: the slab structure never depends on
    the batch)."""
    return int(_lib.load().lcm_stats_bytes(int(M), int(N), int(hw) if hw else 0)) // 4


def _stats_args(stats):
    """(pointer, bytes behind it, slab-count out parameter): the library refuses a launch whose slabs would not fit."""
    if stats is None:
        return None, 0, None
    sp = C.c_int(0)
    return _p(stats.buf), stats.buf.numel() * stats.buf.element_size(), sp


def gemm(a, w, out, *, bias=None, res=None, rowadd=None, rows_per_batch=0, a2=None, epilogue=0, out_scale=1.0,
         M=None, N=None, K=None, lda=None, ldo=None, batch=1, strideA=0, strideW=0, strideO=0, stats=None, img_rows=0):
    """out[m][n] = out_scale * sum_k [a|a2][m][k] w[n][k] + bias + rowadd + res  (see include/lcm_hip.h).
    img_rows: rows per image when the M rows stack independent requests (keys the K partition: results are per-request
    bit-identical at any batch size).  stats: optional ``Stats`` to receive the fused GroupNorm statistics of ``out``
    (stats.P == 0 afterwards: not produced)."""
    L = _lib.load()
    sbuf, sbytes, sp = _stats_args(stats)
    M = a.shape[0] if M is None else M
    K1 = a.shape[-1] if a2 is not None else 0
    K = (a.shape[-1] + (a2.shape[-1] if a2 is not None else 0)) if K is None else K
    N = w.shape[0] if N is None else N
    lda = a.stride(-2) if lda is None else lda
    ldo = out.stride(-2) if ldo is None else ldo
    RECORD = _record_list()
    if RECORD is not None:
        kw = dict(bias=bias, res=res, rowadd=rowadd, rows_per_batch=rows_per_batch, a2=a2, epilogue=epilogue,
                  out_scale=out_scale, M=M, N=N, K=K, lda=lda, ldo=ldo, batch=batch, strideA=strideA, strideW=strideW,
                  strideO=strideO, stats=stats, img_rows=img_rows)
        RECORD.append(((0, M, N, K, batch), dict(halo=False, geglu=(epilogue == 1), m_img=(img_rows if img_rows and M % img_rows == 0 else M),
                                                 splittable=(epilogue == 0 and batch == 1 and not (strideA or strideW or strideO))),
                       lambda: gemm(a, w, out, **kw)))
    with _Timed("gemm", tile_config(M, N, batch) if _profile_list() is not None else "", 2.0 * M * N * K * batch,
                2.0 * batch * (M * K + N * K + M * N)):
        rc = L.lcm_gemm_f16(_p(a), lda, _p(a2), a2.stride(0) if a2 is not None else 0, K1, _p(w), _p(bias), _p(rowadd),
                            rowadd.stride(0) if rowadd is not None else 0, rows_per_batch,
                            _p(res), res.stride(0) if res is not None else 0, _p(out), ldo,
                            M, N, K, epilogue, float(out_scale), batch, strideA, strideW, strideO, int(img_rows),
                            sbuf, sbytes, C.byref(sp) if sp is not None else None, _stream())
    if stats is not None:
        stats.P = sp.value
    _lib.check(rc, "lcm_gemm_f16")
    return out


def gemm_ln(a, w, ln_g, ln_c, out, *, eps=1e-5, epilogue=0, img_rows=0):
    """out = LayerNorm(a) W^T + b with the LayerNorm folded into the contraction (include/lcm_hip.h, lcm_gemm_ln_f16):
    w = gamma (*) W fp16 [N, K], ln_g / ln_c fp32 [N]."""
    L = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    RECORD = _record_list()
    if RECORD is not None:
        RECORD.append(((0, M, N, K, 1), dict(halo=False, geglu=(epilogue == 1), m_img=(img_rows if img_rows and M % img_rows == 0 else M),
                                             splittable=False),
                       lambda: gemm_ln(a, w, ln_g, ln_c, out, eps=eps, epilogue=epilogue, img_rows=img_rows)))
    with _Timed("gemm", "", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)):
        rc = L.lcm_gemm_ln_f16(_p(a), a.stride(0), _p(w), _p(ln_g), _p(ln_c), float(eps), _p(out), out.stride(0), M, N, K,
                               int(epilogue), int(img_rows), _stream())
    _lib.check(rc, "lcm_gemm_ln_f16")
    return out


import os as _os

_MLP_FUSED_MIN_ROWS = int(_os.environ.get("LCM_MLP_FUSED_MIN_ROWS", "24576") or 0)      # 0 switches the fused kernel off


def mlp_fused_applies(M, C, img_rows):
    """Whether the FeedForward of a transformer block runs as the one fused kernel (csrc/mlp_fused.hip): C = 320 (its register
    budget), enough rows to give most CUs a 128-row workgroup (24 576 = 192 workgroups: batches of 6 and more at 512x512),
    and a layer whose canonical K partition of ff.net.2 has one part (the fused kernel accumulates one).  Bit-identical to the
    two-launch form, so -- like a tile shape -- it may follow the batch."""
    if _MLP_FUSED_MIN_ROWS <= 0 or C != 320 or M < _MLP_FUSED_MIN_ROWS:
        return False
    m_img = img_rows if img_rows and M % img_rows == 0 else M
    return canonical_splits(0, m_img, C, 4 * C, 1, 0) == 1


def mlp_geglu(x, w1, ln_g, ln_c, w2, b2, out, *, eps=1e-5, img_rows=0):
    """out = x + Linear_2(GEGLU(Linear_1(LayerNorm(x)))) in one launch (include/lcm_hip.h, lcm_mlp_geglu_f16); out may be x."""
    L = _lib.load()
    M, Cc = x.shape
    F = w2.shape[1]
    RECORD = _record_list()
    if RECORD is not None:          # nothing to tune (no tile / variant choice): recorded for the replay legs only
        RECORD.append((None, None, lambda: mlp_geglu(x, w1, ln_g, ln_c, w2, b2, out, eps=eps, img_rows=img_rows)))
    with _Timed("mlp", "fused", 2.0 * M * (2 * F) * Cc + 2.0 * M * Cc * F, 2.0 * (2 * M * Cc + 2 * F * Cc + Cc * F)):
        rc = L.lcm_mlp_geglu_f16(_p(x), x.stride(0), _p(w1), _p(ln_g), _p(ln_c), float(eps), _p(w2), _p(b2), _p(out),
                                 out.stride(0), M, Cc, int(img_rows), _stream())
    _lib.check(rc, "lcm_mlp_geglu_f16")
    return out


def ln_fold_refresh(w, g_out, c_base=None, c_delta=None, alpha=0.0, c_out=None):
    _lib.check(_lib.load().lcm_ln_fold_refresh(_p(w), w.shape[0], w.shape[1], _p(c_base), _p(c_delta), float(alpha), _p(g_out),
                                               _p(c_out), _stream()), "lcm_ln_fold_refresh")


def conv3x3(x, w, out, B, H, W, Cin, Cout, *, bias=None, rowadd=None, res=None, stride=1, ups=0, stats=None, out_hw=None):
    """out_hw: (Ho, Wo) of an upsampling conv whose target is the odd-sized skip tensor (2H-1 / 2W-1): Upsample2D called
    with output_size; F.interpolate(size=..., mode="nearest") is the 2x result cropped by one row / column, and the conv
    then pads THAT with zeros -- so only the loader-fused form (ups=1, plain 3x3 weights) can serve it."""
    L = _lib.load()
    sbuf, sbytes, sp = _stats_args(stats)
    Ho, Wo = ((2 * H, 2 * W) if ups else ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W))
    flags = ups
    if out_hw is not None and tuple(out_hw) != (Ho, Wo):
        if ups != 1 or out_hw[0] not in (2 * H, 2 * H - 1) or out_hw[1] not in (2 * W, 2 * W - 1):
            raise _lib.LcmHipError(f"conv3x3: output size {tuple(out_hw)} from {H}x{W} needs ups=1 (plain 3x3 weights) and 2n / 2n-1 "
                                   f"targets, got ups={ups}")
        flags = ups | (4 if out_hw[0] == 2 * H - 1 else 0) | (8 if out_hw[1] == 2 * W - 1 else 0)
        Ho, Wo = out_hw
    Mo = B * Ho * Wo
    taps = 4 if ups == 2 else 9                 # ups=2: phase-packed weights (packing.pack_conv3x3_up2), 4 taps per output
    RECORD = _record_list()
    if RECORD is not None:
        kw = dict(bias=bias, rowadd=rowadd, res=res, stride=stride, ups=ups, stats=stats, out_hw=out_hw)
        key = (1, Mo, Cout, 9 * Cin, 1) if stride == 2 else (2, Mo, Cout, taps * Cin, (Wo << 1))
        RECORD.append((key, dict(halo=stride == 1, W=(W if ups == 2 else Wo), phases=4 if ups == 2 else 1, m_img=Mo // B, splittable=True),
                       lambda: conv3x3(x, w, out, B, H, W, Cin, Cout, **kw)))
    with _Timed("conv3x3", tile_config(Mo, Cout) if _profile_list() is not None else "", 2.0 * Mo * Cout * taps * Cin,
                2.0 * (B * H * W * Cin + 9 * Cin * Cout + Mo * Cout)):
        rc = L.lcm_conv3x3_f16(_p(x), _p(w), _p(bias), _p(rowadd), rowadd.stride(0) if rowadd is not None else 0,
                               _p(res), _p(out), B, H, W, Cin, Cout, stride, flags, sbuf, sbytes,
                               C.byref(sp) if sp is not None else None, _stream())
    if stats is not None:
        stats.P = sp.value
    _lib.check(rc, "lcm_conv3x3_f16")
    return out


def conv3x3_gn(x, w, out, B, H, W, C1, Cout, *, x2=None, C2=0, gn_scale=None, gn_shift=None, silu=True, bias=None,
               rowadd=None, res=None, ups=0, stats=None):
    """Fused [GroupNorm-apply (+SiLU)] -> conv3x3 (stride 1) over the channel concat [x | x2]."""
    L = _lib.load()
    sbuf, sbytes, sp = _stats_args(stats)
    Cin = C1 + (C2 if x2 is not None else 0)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    Mo = B * Ho * Wo
    RECORD = _record_list()
    if RECORD is not None:
        kw = dict(x2=x2, C2=C2, gn_scale=gn_scale, gn_shift=gn_shift, silu=silu, bias=bias, rowadd=rowadd, res=res, ups=ups,
                  stats=stats)
        RECORD.append(((2, Mo, Cout, 9 * Cin, (Wo << 1) | (1 if gn_scale is not None else 0)),
                       dict(halo=True, W=Wo, m_img=Mo // B, splittable=True), lambda: conv3x3_gn(x, w, out, B, H, W, C1, Cout, **kw)))
    with _Timed("conv3x3", "halo", 2.0 * Mo * Cout * 9 * Cin, 2.0 * (B * H * W * Cin + 9 * Cin * Cout + Mo * Cout)):
        rc = L.lcm_conv3x3_gn_f16(_p(x), C1, _p(x2), C2 if x2 is not None else 0, _p(gn_scale), _p(gn_shift),
                                  1 if silu else 0, _p(w), _p(bias), _p(rowadd),
                                  rowadd.stride(0) if rowadd is not None else 0, _p(res), _p(out), B, H, W, Cout, ups,
                                  sbuf, sbytes, C.byref(sp) if sp is not None else None, _stream())
    if stats is not None:
        stats.P = sp.value
    _lib.check(rc, "lcm_conv3x3_gn_f16")
    return out


def groupnorm_affine(x, gamma, beta, scale, shift, B, HW, C1, ws, *, x2=None, C2=0, groups=32, eps=1e-5):
    L = _lib.load()
    rc = L.lcm_groupnorm_affine_f16(_p(x), C1, _p(x2), C2 if x2 is not None else 0, _p(gamma), _p(beta), _p(scale),
                                    _p(shift), B, HW, groups, float(eps), _p(ws), _stream())
    _lib.check(rc, "lcm_groupnorm_affine_f16")


def set_halo_pipe_threshold(wgs):
    _lib.check(_lib.load().lcm_set_halo_pipe_threshold(int(wgs)), "lcm_set_halo_pipe_threshold")


def set_halo_prefetch(on):
    _lib.check(_lib.load().lcm_set_halo_prefetch(1 if on else 0), "lcm_set_halo_prefetch")


def set_persist_n(on):
    _lib.check(_lib.load().lcm_set_persist_n(1 if on else 0), "lcm_set_persist_n")


def set_conv_impl(impl):
    _lib.check(_lib.load().lcm_set_conv_impl(int(impl)), "lcm_set_conv_impl")


def conv3x3_c4(lat_f32, w, out, B, H, W, Cout, *, bias=None, pre_w=None, pre_b=None, in_scale=1.0):
    L = _lib.load()
    rc = L.lcm_conv3x3_c4_f32in(_p(lat_f32), _p(pre_w), _p(pre_b), float(in_scale), _p(w), _p(bias), _p(out),
                                B, H, W, Cout, _stream())
    _lib.check(rc, "lcm_conv3x3_c4_f32in")
    return out


def conv3x3_smalln(x, w, out, B, H, W, Cin, Cout, *, bias=None, mode=0, out_f32=None, gn_scale=None, gn_shift=None, silu=True):
    """gn_scale / gn_shift (groupnorm_tables_from_stats): x is the raw tensor, GroupNorm-apply (+SiLU) fused into the staging."""
    L = _lib.load()
    if gn_scale is not None:
        rc = L.lcm_conv3x3_smalln_gn(_p(x), _p(gn_scale), _p(gn_shift), 1 if silu else 0, _p(w), _p(bias), _p(out), _p(out_f32),
                                     B, H, W, Cin, Cout, mode, _stream())
    else:
        rc = L.lcm_conv3x3_smalln(_p(x), _p(w), _p(bias), _p(out), _p(out_f32), B, H, W, Cin, Cout, mode, _stream())
    _lib.check(rc, "lcm_conv3x3_smalln")
    return out


def groupnorm_ws_bytes(B, HW, C, groups=32):
    return int(_lib.load().lcm_groupnorm_ws_bytes(B, HW, C, groups))


def groupnorm(x, gamma, beta, out, B, HW, C1, ws, *, x2=None, C2=0, groups=32, eps=1e-5, silu=True):
    L = _lib.load()
    rc = L.lcm_groupnorm_f16(_p(x), C1, _p(x2), C2, _p(gamma), _p(beta), _p(out), B, HW, groups, float(eps),
                             1 if silu else 0, _p(ws), _stream())
    _lib.check(rc, "lcm_groupnorm_f16")
    return out


def groupnorm_from_stats(x, gamma, beta, out, B, HW, C1, st1, ws, *, x2=None, C2=0, st2=None, groups=32, eps=1e-5, silu=True):
    """GroupNorm(+SiLU) of [x | x2] using producer-written statistics (``Stats`` objects with P > 0)."""
    L = _lib.load()
    rc = L.lcm_groupnorm_from_stats_f16(_p(x), C1, _p(x2), C2 if x2 is not None else 0, _p(st1.buf), st1.P,
                                        _p(st2.buf) if st2 is not None else None, st2.P if st2 is not None else 0,
                                        _p(gamma), _p(beta), _p(out), B, HW, groups, float(eps), 1 if silu else 0, _p(ws), _stream())
    _lib.check(rc, "lcm_groupnorm_from_stats_f16")
    return out


def groupnorm_tables_from_stats(gamma, beta, B, HW, C1, st1, ws, *, C2=0, st2=None, groups=32, eps=1e-5):
    """Finalize only: -> (scale, shift) fp32 [B, C1+C2] views of ``ws`` for ``conv3x3_gn`` (the apply pass is skipped)."""
    L = _lib.load()
    C = C1 + (C2 if st2 is not None else 0)
    rc = L.lcm_groupnorm_from_stats_f16(None, C1, None, C2 if st2 is not None else 0, _p(st1.buf), st1.P,
                                        _p(st2.buf) if st2 is not None else None, st2.P if st2 is not None else 0,
                                        _p(gamma), _p(beta), None, B, HW, groups, float(eps), 0, _p(ws), _stream())
    _lib.check(rc, "lcm_groupnorm_from_stats_f16")
    return ws[:B * C].view(B, C), ws[B * C:2 * B * C].view(B, C)


def layernorm(x, gamma, beta, out, M, C, eps=1e-5):
    L = _lib.load()
    _lib.check(L.lcm_layernorm_f16(_p(x), _p(gamma), _p(beta), _p(out), M, C, float(eps), _stream()), "lcm_layernorm_f16")
    return out


def embed_tokens(ids, tok_emb, pos_emb, out, B, S, D):
    L = _lib.load()
    _lib.check(L.lcm_embed_tokens_f16(_p(ids), _p(tok_emb), _p(pos_emb), _p(out), B, S, D, tok_emb.shape[0], _stream()),
               "lcm_embed_tokens_f16")
    return out


def attention(q, k, v, out, B, heads, Sq, Sk, d, *, ldq, ldk, ldv, ldo, scale=None, causal=False):
    """scale None: d^-0.5.  scale 0: q already carries scale * log2(e) (folded into the projection that made it, model.Q_PRESCALE)."""
    L = _lib.load()
    scale = d ** -0.5 if scale is None else scale
    RECORD = _record_list()
    if RECORD is not None:
        RECORD.append((None, None, lambda: attention(q, k, v, out, B, heads, Sq, Sk, d, ldq=ldq, ldk=ldk, ldv=ldv, ldo=ldo,
                                                     scale=scale, causal=causal)))
    with _Timed("attention", f"d{d}", 4.0 * B * heads * Sq * Sk * d, 2.0 * B * heads * d * (2 * Sq + 2 * Sk)):
        rc = L.lcm_attention_f16(_p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, B, heads, Sq, Sk, d, float(scale),
                                 1 if causal else 0, _stream())
    _lib.check(rc, "lcm_attention_f16")
    return out


def softmax_rows(x, rows, n, ld):
    L = _lib.load()
    _lib.check(L.lcm_softmax_rows_f16(_p(x), rows, n, ld, _stream()), "lcm_softmax_rows_f16")
    return x


def transpose(x, out, R, Cc, *, ldi, ldo, batch=1, stride_in=0, stride_out=0):
    L = _lib.load()
    _lib.check(L.lcm_transpose_f16(_p(x), ldi, _p(out), ldo, R, Cc, batch, stride_in, stride_out, _stream()), "lcm_transpose_f16")
    return out


def linear_smallm(x, w, out, M, N, K, *, bias=None, res=None, silu_in=False, silu_out=False, ldx=None, ldo=None):
    L = _lib.load()
    rc = L.lcm_linear_smallm_f16(_p(x), x.stride(0) if ldx is None else ldx, _p(w), _p(bias), _p(res),
                                 res.stride(0) if res is not None else 0, _p(out), out.stride(0) if ldo is None else ldo,
                                 M, N, K, int(silu_in), int(silu_out), _stream())
    _lib.check(rc, "lcm_linear_smallm_f16")
    return out


def linear_rows(x, w, out, M, N, K, *, x_rows=None, bias=None, res=None, res_rows=None, silu_in=False, silu_out=False):
    """linear_smallm for any M; row m reads x[m % x_rows] and res[m % res_rows] (lcm_linear_rows_f16)."""
    L = _lib.load()
    rc = L.lcm_linear_rows_f16(_p(x), x.stride(0), M if x_rows is None else x_rows, _p(w), _p(bias), _p(res),
                               res.stride(0) if res is not None else 0, M if res_rows is None else res_rows, _p(out), out.stride(0),
                               M, N, K, int(silu_in), int(silu_out), _stream())
    _lib.check(rc, "lcm_linear_rows_f16")
    return out


def timestep_embedding_steps(ts, out, B, dim):
    """out [len(ts) * B, dim], rows step-major."""
    L = _lib.load()
    arr = (C.c_float * len(ts))(*[float(t) for t in ts])
    _lib.check(L.lcm_timestep_embedding_steps(arr, len(ts), _p(out), B, dim, _stream()), "lcm_timestep_embedding_steps")
    return out


def timestep_embedding(t, out, B, dim):
    L = _lib.load()
    _lib.check(L.lcm_timestep_embedding(float(t), _p(out), B, dim, _stream()), "lcm_timestep_embedding")
    return out


def scheduler_step(eps, lat, noise, coef6, last, B, h, w, *, eps_uncond=None, guidance=1.0):
    L = _lib.load()
    arr = (C.c_float * 6)(*[float(c) for c in coef6])
    rc = L.lcm_scheduler_step(_p(eps), _p(eps_uncond), float(guidance), _p(lat), _p(noise), arr, int(bool(last)),
                              B, h, w, _stream())
    _lib.check(rc, "lcm_scheduler_step")
    return lat


def latents_pool8(lat, out, B, h, w):
    L = _lib.load()
    _lib.check(L.lcm_latents_pool8(_p(lat), _p(out), B, h, w, _stream()), "lcm_latents_pool8")
    return out


def set_workspace(t):
    """Register an fp32 scratch tensor for deterministic split-K on the current device (None disables)."""
    L = _lib.load()
    _lib.check(L.lcm_set_workspace(_p(t), 0 if t is None else t.numel() * t.element_size()), "lcm_set_workspace")


def set_stream_workspace(stream, t, forget=False):
    """A split-K workspace of its own for the launches of ``stream`` (a torch.cuda.Stream).  The entry is owned by the tensor
    that registered it: registering another tensor over a live entry raises; ``forget=True`` removes the entry only if it still
    holds ``t``; ``t=None`` removes it whoever owns it."""
    L = _lib.load()
    nbytes = 0 if (t is None or forget) else t.numel() * t.element_size()
    _lib.check(L.lcm_set_stream_workspace(C.c_void_p(stream.cuda_stream), _p(t), nbytes), "lcm_set_stream_workspace")


_OWN_STREAMS = {}            # device index -> idle streams created through lcm_stream_create


def acquire_stream(device):
    """A stream no other owner in this process holds: torch.cuda.Stream() hands handles out of a 32-entry round-robin pool per
    device, so after ~16 lanes (engine reloads, an SD1.5 beside an SDXL engine, a long test session) two live pipelines would
    be keyed on ONE handle -- and the library keys a lane's split-K workspace on it.  Streams come from lcm_stream_create,
    wrapped as torch.cuda.ExternalStream, and go back to an idle list on release (never destroyed: the allocator may still
    hold blocks tagged with them)."""
    import torch
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    idle = _OWN_STREAMS.setdefault(idx, [])
    if idle:
        return idle.pop()
    L = _lib.load()
    out = C.c_void_p()
    with torch.cuda.device(idx):
        _lib.check(L.lcm_stream_create(C.byref(out)), "lcm_stream_create")
    return torch.cuda.ExternalStream(out.value, device=torch.device("cuda", idx))


def release_stream(stream):
    if stream is not None:
        _OWN_STREAMS.setdefault(stream.device.index, []).append(stream)


def set_tuning(target_wgs=0, max_splits=0, min_wgs=0):
    _lib.check(_lib.load().lcm_set_tuning(int(target_wgs), int(max_splits), int(min_wgs)), "lcm_set_tuning")


def set_split_policy(max_rows_per_image=4096, max_parts=8):
    _lib.check(_lib.load().lcm_set_split_policy(int(max_rows_per_image), int(max_parts)), "lcm_set_split_policy")


def set_attention_waves(waves):
    _lib.check(_lib.load().lcm_set_attention_waves(int(waves)), "lcm_set_attention_waves")


def set_attention_impl(impl):
    """1 (default): long non-causal sequences (Sk >= 128, d in 40 / 64 / 80) on the streaming kernel; 0: everything on the
    register-staged kernel (A/B switch)."""
    _lib.check(_lib.load().lcm_set_attention_impl(int(impl)), "lcm_set_attention_impl")


def set_attention_ksplit(on):
    _lib.check(_lib.load().lcm_set_attention_ksplit(int(on)), "lcm_set_attention_ksplit")


def set_seg_mode(mode):
    """0 auto, 1 always segmented accumulation, 2 always split + reduce (bit-identical; include/lcm_hip.h)."""
    _lib.check(_lib.load().lcm_set_seg_mode(int(mode)), "lcm_set_seg_mode")


def set_gn_fused_bytes(n):
    _lib.check(_lib.load().lcm_set_gn_fused_bytes(int(n)), "lcm_set_gn_fused_bytes")


def set_kernel_variant(v):
    _lib.check(_lib.load().lcm_set_kernel_variant(int(v)), "lcm_set_kernel_variant")


def profile_begin(max_launches=1 << 17):
    _lib.check(_lib.load().lcm_profile_begin(int(max_launches)), "lcm_profile_begin")


def profile_end():
    """-> list of (kernel instantiation, milliseconds) in launch order."""
    buf = C.create_string_buffer(1 << 24)
    n = _lib.load().lcm_profile_end(buf, len(buf))
    if n < 0:
        _lib.check(n, "lcm_profile_end")
    out = []
    for line in buf.value.decode().splitlines():
        name, ms = line.rsplit("\t", 1)
        out.append((name, float(ms)))
    return out


def axpy(base, delta, alpha, out):
    """out = base + alpha * delta (fp16, fp32 math); out may be the live weight tensor."""
    _lib.check(_lib.load().lcm_axpy_f16(_p(base), _p(delta), float(alpha), _p(out), base.numel(), _stream()), "lcm_axpy_f16")
    return out


def vae_blend(a, ah, aw, b, bh, bw, B, extent, vertical):
    _lib.check(_lib.load().lcm_vae_blend_f32(_p(a), ah, aw, _p(b), bh, bw, B, extent, 1 if vertical else 0, _stream()), "lcm_vae_blend_f32")


def vae_place_tile(tile, th, tw, out_u8, out_f32, H, W, B, oy, ox, ch, cw):
    _lib.check(_lib.load().lcm_vae_place_tile(_p(tile), th, tw, _p(out_u8), _p(out_f32), H, W, B, oy, ox, ch, cw, _stream()),
               "lcm_vae_place_tile")


def set_staged_epilogue(on):
    """A/B switch: 1 (default) = the halo conv's plain launches move the residual / result tile through LDS in whole rows."""
    _lib.check(_lib.load().lcm_set_staged_epilogue(int(on)), "lcm_set_staged_epilogue")


def debug_spin(usec):
    _lib.check(_lib.load().lcm_debug_spin(int(usec), _stream()), "lcm_debug_spin")


class Graph:
    """hipGraph captured from the kernels enqueued on the current torch stream."""

    def __init__(self):
        self._exec = C.c_void_p()

    def __enter__(self):
        _lib.check(_lib.load().lcm_graph_begin(_stream()), "lcm_graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = _lib.load().lcm_graph_end(_stream(), C.byref(self._exec))
        if et is None:
            _lib.check(rc, "lcm_graph_end")
        return False

    def launch(self):
        _lib.check(_lib.load().lcm_graph_launch(self._exec, _stream()), "lcm_graph_launch")

    def close(self):
        if self._exec:
            _lib.load().lcm_graph_destroy(self._exec)
            self._exec = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
