#!/usr/bin/env python3
"""What one dependency costs inside a launch against a kernel boundary, on this box (a cooperative kernel for
the mid-block's 11-launch chain, kept only if >= 3x faster).  Measures
  (a) a grid-wide seam inside ONE launch of 256 / 128 / 64 co-resident workgroups (lcm_debug_grid_barrier: release + monotonic
      counter + bounded relaxed poll + acquire + re-read of another workgroup's record), per seam;
  (b) the boundary between two trivial dependent kernels on one stream, eager and inside a captured hipGraph;
  (c) the mid-block's transformer chain as it ships (batch 1, 8x8 level, C = 1280: 11 launches), per launch and in total.
The cooperative kernel would replace 10 boundaries by 10 seams and keep the bodies (at best 0.85x, the guide's phase-in-launch
row): it wins 3x only if (a) << (b), which the numbers below answer."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa: E402,F401
from sdlcm_amd import lib, ops  # noqa: E402
import ctypes as C  # noqa: E402

DEV = "cuda:0"
L = lib.load()


def ev_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.debug_spin(50)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def seams():
    state = torch.zeros(16 + 256 * 128, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wgs in (256, 128, 64):
        res = {}
        for n in (0, 10, 110):
            res[n] = ev_time(lambda: lib.check(L.lcm_debug_grid_barrier(wgs, n, C.c_void_p(state.data_ptr()), st), "grid_barrier"))[0]
        torch.cuda.synchronize()
        timeout = int(state[4:8].view(torch.int32).item())
        print(f"(a) {wgs:3d} workgroups: launch alone {res[0]:6.1f} us; per seam {(res[110] - res[10]) / 100:5.2f} us "
              f"(10 seams {res[10] - res[0]:6.1f} us); timeout word {timeout}", flush=True)


def boundaries():
    x = torch.zeros(256 * 256, device=DEV)
    def chain(n):
        for _ in range(n):
            x.add_(1.0)
    for n in (1, 11, 111):
        med, mn = ev_time(lambda: chain(n))
        print(f"(b) eager chain of {n:3d} trivial dependent kernels: {med:7.1f} us", flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(3)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            chain(110)
    torch.cuda.synchronize()
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e6)
    t.sort()
    print(f"(b) hipGraph of 110 trivial dependent kernels: {t[len(t) // 2]:7.1f} us per replay = {t[len(t) // 2] / 110:5.2f} us per kernel + boundary", flush=True)


def midblock():
    from sdlcm_amd import weights
    from sdlcm_amd.pipeline import LcmHipPipeline
    pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device=DEV)
    P = pipe.plan(1, 64, 64, 4)
    with torch.cuda.stream(P.lane.stream):
        pipe.tune(P)
        pipe._enqueue(P, 1.0)
        P.lane.stream.synchronize()
        with ops.profiling() as recs:
            ops.profile_begin()
            pipe._enqueue(P, 1.0)
            P.lane.stream.synchronize()
            times = ops.profile_end()
    print(f"(c) one eager batch-1 pass: {len(times)} bracketed MFMA launches, {sum(ms for _, ms in times) * 1e3:.0f} us inside brackets", flush=True)
    # the mid-block transformer = the GEMM / attention launches whose M is 64 rows (8x8) and N, K in {1280, 3840, 10240, 5120}
    small = [(n, ms) for (n, ms), r in zip(times, recs) if r["kind"] in ("gemm", "attention") and r["flops"] < 2.2e9]
    print(f"(c) launches under 2.2 GFLOP (the 16x16 / 8x8 levels' transformer GEMMs and attention): {len(small)}, "
          f"mean {sum(ms for _, ms in small) / max(1, len(small)) * 1e3:.1f} us bracketed", flush=True)
    pipe.close()


if __name__ == "__main__":
    seams()
    boundaries()
    midblock()
