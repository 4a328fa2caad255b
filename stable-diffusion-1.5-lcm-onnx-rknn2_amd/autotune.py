"""Per-shape launch autotuner for the MFMA contraction kernels (gemm / conv3x3).

The sampler touches ~80 distinct (M, N, K) contraction shapes per (batch, size) plan.  Tile shape and pipeline
variant are pure launch parameters -- they never change a bit of the result (the K partition alone fixes the fp32
summation order, and the fused statistics are slab-canonical: csrc/igemm.hip, "What decides the numbers") -- and the
best choice depends on how many workgroups a shape yields on 256 CUs.  At plan-build time every distinct shape
without an entry in the shipped table is replayed on its real buffers under each candidate configuration, timed with
HIP events on the launch stream, and the winner is written to the library's plan table (``lcm_plan_set``).

The split-K factor is NOT a launch parameter: it is the canonical partition of the per-image shape
(``ops.canonical_splits``) and is only ever tuned offline (``tools/make_plans.py``, ``tune_splits=True``), into the
table that ships with the package -- never by timing in a serving process, so results are reproducible across
processes, boxes and batch sizes.

Disable with LCM_AUTOTUNE=0 (the built-in heuristics of csrc/igemm.hip then apply).
"""
from __future__ import annotations

import os

import torch

from . import ops


def _canonical_splits(key, meta, M=None):
    kind, M0, N, K, aux = key
    ph = 1 if meta.get("phases", 1) == 4 else 0
    return ops.canonical_splits(kind, M0 if M is None else M, N, K, aux if kind == 2 else 1, ph)


def _candidates(key, meta, ws_bytes, cold=False, tune_splits=False):
    kind, M, N, K, aux = key
    nk = K // 64
    tiles = []
    for bm in (128, 64):
        for bn in (160, 128, 64):
            if N % bn:
                continue
            if bn == 160 and (kind == 1 or meta.get("geglu")):     # 160-wide tile: plain GEMM and halo conv only
                continue
            if meta["halo"]:
                tw = 16 if (meta["W"] % 16 == 0 or meta["W"] > 16) else 8
                if tw == 8 and bm == 128:
                    continue
            elif bm == 128 and M < 128:
                continue
            tiles.append((bm, bn))
    # the plan entry's own split field is the canonical partition of a request whose rows-per-image equal this M
    entry_splits = _canonical_splits(key, meta)
    out = []
    for bm, bn in tiles:
        ntile = -(-M // bm) * (N // bn) * (aux if kind == 0 else 1)
        taps = 4 if meta.get("phases", 1) == 4 else 9         # phase-decomposed upsample conv: K = 4 * Cin
        units = (K // (64 * taps)) if meta["halo"] else nk    # halo conv splits over 64-channel chunks
        splits = [entry_splits]
        if tune_splits and meta.get("splittable") and meta.get("m_img", M) == M and M <= 4096:
            # offline, single-image shapes only, inside the library's split policy (csrc/igemm.hip lcm_split_policy:
            # <= 4096 rows per image, <= 8 parts)
            splits = [1]
            for s in (2, 3, 4, 5, 6, 8):
                if s <= units and ntile * s <= 4096 and s * M * N * 4 <= ws_bytes and (meta["halo"] or s <= nk // 2):
                    splits.append(s)
        for s in splits:
            # (GEMM: 0 = register-staged double buffer, LDS-DMA ring depth 1/2/4; halo conv: 1 = single-buffer, 2 =
            # pipelined weight ring, 3 = pipelined with a whole kernel row of taps per K-step -- 64/128-wide tiles only).
            # The depth is a fair candidate only when the timing runs cold (autotune(cold=True)): replaying one launch
            # warm keeps its weights cache-resident, which hides exactly the latency the deeper variants exist for.
            variants = ((-1, 0, 1, 2, 4) if kind == 0 else ((1, 2, 3) if bn <= 128 else (1, 2)) if meta["halo"] else (-1,)) if cold else (-1,)
            for v in variants:
                out.append((bm, bn, s, v))
    return out


def _time(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


_FLUSH = {}


def _time_cold(fn, reps):
    """Per-launch time with every cache level flushed before each launch: at batch 1-2 a layer's weights always come
    from HBM in situ (1.7 GB of UNet weights per step against 256 MB of MALL), while a back-to-back replay of one
    layer keeps them cache-resident and favours the wrong tile / split-K / depth."""
    dev = torch.cuda.current_device()
    buf = _FLUSH.get(dev)
    if buf is None:
        buf = _FLUSH[dev] = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    fn()
    total = 0.0
    evs = []
    for _ in range(reps):
        buf.fill_(1)                              # 768 MB of writes: evicts L2 and MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    evs[-1][1].synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts[:max(1, len(ts) // 2)]) / max(1, len(ts) // 2)      # mean of the faster half (event jitter is one-sided)


def release_flush_buffers():
    _FLUSH.clear()


def _cache_path():
    return os.environ.get("LCM_TUNE_CACHE", "")


def _load_cache():
    from . import lib as _lib
    return _lib.known_plans()


def _save_cache(cache):
    import json
    p = _cache_path()
    if p:
        with open(p, "w") as f:
            json.dump({",".join(str(v) for v in k): list(val) for k, val in cache.items()}, f)


def autotune(records, ws_bytes, reps=None, verbose=False, cold=False, tune_splits=False):
    """records: list of (key, meta, replay) from ops.RECORD.  Returns {key: (bm, bn, splits, variant, ms)}.
    LCM_TUNE_CACHE=<file> persists the winners (no tuning launches next time).  tune_splits: offline table generation
    only (tools/make_plans.py)."""
    reps = reps or int(os.environ.get("LCM_AUTOTUNE_REPS", "6"))
    seen, chosen = {}, {}
    cache = _load_cache()
    dirty = False
    for key, meta, fn in records:
        seen.setdefault(key, (meta, fn))
    # entries of the shipped / cached table first, single-image shapes before stacked ones: a stacked shape's K partition
    # is read from the entry of its per-image shape
    order = sorted(seen.items(), key=lambda kv: (kv[0] not in cache, kv[1][0].get("m_img", kv[0][1]) != kv[0][1]))
    for key, (meta, fn) in order:
        if key in cache:
            bm, bn, s, v = (int(x) for x in cache[key][:4])
            ops.plan_set(key[0], key[1], key[2], key[3], key[4], bm, bn, s, v)
            chosen[key] = (bm, bn, s, v, float(cache[key][4]) if len(cache[key]) > 4 else 0.0)
            continue
        best = None
        for (bm, bn, s, v) in _candidates(key, meta, ws_bytes, cold, tune_splits):
            ops.plan_set(key[0], key[1], key[2], key[3], key[4], bm, bn, s, v)
            if cold:
                ms = _time_cold(fn, reps)
            else:
                # hold the stream briefly so the timed launches run back to back (host enqueue is slower than tiny kernels)
                ops.debug_spin(150)
                ms = _time(fn, reps)
            if best is None or ms < best[4]:
                best = (bm, bn, s, v, ms)
        ops.plan_set(key[0], key[1], key[2], key[3], key[4], *best[:4])
        chosen[key] = best
        cache[key] = best
        dirty = True
        if verbose:
            print(f"[autotune] kind{key[0]} M{key[1]} N{key[2]} K{key[3]} aux{key[4]} -> {best[0]}x{best[1]} "
                  f"splits {best[2]} variant {best[3]} {best[4] * 1e3:.1f}us")
    if dirty:
        _save_cache(cache)
    if cold:
        release_flush_buffers()
    return chosen
