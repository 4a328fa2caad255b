"""ctypes binding of liblcmhip.so (the C ABI declared in include/lcm_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, a
``LcmHipError`` is raised (the worker pool turns it into a failed future / HTTP 500,
backends/worker_pool.py:333-336 in the reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcmhip.so")


class LcmHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into liblcmhip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise LcmHipError("building liblcmhip.so failed")
    return LIB_PATH


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

_SIGS = {
    "lcm_gemm_f16": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i,
                     _i64, _i64, _i64, _i, _vp, _i64, C.POINTER(_i), _vp],
    "lcm_gemm_ln_f16": [_vp, _i, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_mlp_geglu_f16": [_vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lcm_ln_fold_refresh": [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp],
    "lcm_conv3x3_f16": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, C.POINTER(_i), _vp],
    "lcm_groupnorm_from_stats_f16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "lcm_conv3x3_c4_f32in": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lcm_conv3x3_smalln": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_conv3x3_smalln_gn": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_groupnorm_f16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "lcm_layernorm_f16": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "lcm_attention_f16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "lcm_embed_tokens_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lcm_softmax_rows_f16": [_vp, _i, _i, _i, _vp],
    "lcm_transpose_f16": [_vp, _i, _vp, _i, _i, _i, _i, _i64, _i64, _vp],
    "lcm_linear_smallm_f16": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_timestep_embedding": [_f, _vp, _i, _i, _vp],
    "lcm_linear_rows_f16": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_timestep_embedding_steps": [_vp, _i, _vp, _i, _i, _vp],
    "lcm_scheduler_step": [_vp, _vp, _f, _vp, _vp, C.POINTER(C.c_float), _i, _i, _i, _i, _vp],
    "lcm_latents_pool8": [_vp, _vp, _i, _i, _i, _vp],
    "lcm_png_encode_rgb8": [_vp, _i, _i, C.c_longlong, _i, _vp, C.c_longlong, C.POINTER(C.c_longlong)],
    "lcm_stream_create": [C.POINTER(_vp)],
    "lcm_stream_destroy": [_vp],
    "lcm_graph_begin": [_vp],
    "lcm_graph_end": [_vp, C.POINTER(_vp)],
    "lcm_graph_launch": [_vp, _vp],
    "lcm_graph_destroy": [_vp],
    "lcm_gemm_tile_config": [_i, _i, _i],
    "lcm_debug_spin": [_i, _vp],
    "lcm_debug_grid_barrier": [_i, _i, _vp, _vp],
    "lcm_axpy_f16": [_vp, _vp, _f, _vp, _i64, _vp],
    "lcm_vae_blend_f32": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "lcm_vae_place_tile": [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lcm_profile_begin": [_i],
    "lcm_profile_end": [C.c_char_p, _i64],
    "lcm_set_workspace": [_vp, _i64],
    "lcm_set_stream_workspace": [_vp, _vp, _i64],
    "lcm_set_tuning": [_i, _i, _i],
    "lcm_set_split_policy": [_i, _i],
    "lcm_set_seg_mode": [_i],
    "lcm_set_attention_waves": [_i],
    "lcm_set_attention_impl": [_i],
    "lcm_set_attention_ksplit": [_i],
    "lcm_set_kernel_variant": [_i],
    "lcm_set_conv_impl": [_i],
    "lcm_set_persist_n": [_i],
    "lcm_set_halo_pipe_threshold": [_i],
    "lcm_set_halo_prefetch": [_i],
    "lcm_set_staged_epilogue": [_i],
    "lcm_set_gn_fused_bytes": [_i64],
    "lcm_plan_set": [_i] * 9,
    "lcm_plan_clear": [],
    "lcm_canonical_splits": [_i] * 6,
    "lcm_conv3x3_gn_f16": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i64,
                           C.POINTER(_i), _vp],
    "lcm_groupnorm_affine_f16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp],
    "lcm_device_info": [_i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(C.c_uint64)],
}
EXPORTS = tuple(sorted(list(_SIGS) + ["lcm_last_error", "lcm_version", "lcm_groupnorm_ws_bytes", "lcm_stats_bytes", "lcm_png_bound"]))

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcmHipError(f"{LIB_PATH} not built: run __graft_entry__.build() (make -C csrc); "
                          "there is no CPU fallback for the HIP path")
    # liblcmhip.so needs libamdhip64.so.7; PyTorch-ROCm bundles its own copy.  Whichever is mapped first serves BOTH, so
    # torch must come first: with the system runtime loaded ahead of torch's, device memory, streams and the kernels end
    # up on mismatched runtimes ("set_workspace: no device" when build() and smoke() ran in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = sig
        fn.restype = _i
    lib.lcm_last_error.restype = C.c_char_p
    lib.lcm_last_error.argtypes = []
    lib.lcm_version.restype = _i
    lib.lcm_groupnorm_ws_bytes.restype = _i64
    lib.lcm_groupnorm_ws_bytes.argtypes = [_i, _i, _i, _i]
    lib.lcm_stats_bytes.restype = _i64
    lib.lcm_png_bound.restype = C.c_longlong
    lib.lcm_png_bound.argtypes = [_i, _i, _i]
    lib.lcm_stats_bytes.argtypes = [_i, _i, _i]
    _install_plans(lib)
    _lib = lib
    return lib


PACKAGED_PLANS = os.path.join(_HERE, "tuned_plans_gfx950.json")


def read_plans(path) -> dict:
    import json
    try:
        with open(path) as f:
            return {tuple(int(v) for v in k.split(",")): tuple(val) for k, val in json.load(f).items()}
    except Exception:
        return {}


def known_plans() -> dict:
    """The launch-plan table known before any tuning launch: the one shipped with the package (tools/make_plans.py: the
    standard SD1.5 / SDXL request shapes, tuned offline on an MI355X) overlaid by the user's LCM_TUNE_CACHE file.
    LCM_TUNED_PLANS=0 ignores the shipped table."""
    plans = {}
    if os.environ.get("LCM_TUNED_PLANS", "1") != "0" and os.path.exists(PACKAGED_PLANS):
        plans.update(read_plans(PACKAGED_PLANS))
    p = os.environ.get("LCM_TUNE_CACHE", "")
    if p and os.path.exists(p):
        plans.update(read_plans(p))
    return plans


def _install_plans(lib):
    """The WHOLE table goes into the library when it is loaded, not shape by shape as plans are built: a launch reads
    its K partition from the entry of its per-image shape, which must not depend on which requests this process
    happened to serve first (include/lcm_hip.h, Determinism)."""
    for (kind, M, N, K, aux), val in known_plans().items():
        bm, bn, sp, v = (int(x) for x in val[:4])
        lib.lcm_plan_set(kind, M, N, K, aux, bm, bn, sp, v)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().lcm_last_error().decode("utf-8", "replace")
        raise LcmHipError(f"{what or 'lcm call'} failed (rc={rc}): {msg}")
