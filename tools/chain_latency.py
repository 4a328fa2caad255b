#!/usr/bin/env python3
"""Per-kernel cost inside a hipGraph chain of dependent launches (what a batch-1 pass is made of): N launches of one
op captured in a graph, replayed, wall / N.  Rotates over R distinct weight sets larger than the caches so weights
stream from HBM as in situ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
st = torch.cuda.Stream()
N = 240


def chain(name, make_fn, flops=0.0):
    with torch.cuda.stream(st):
        fns = make_fn()
        for f in fns[:4]:
            f()
        st.synchronize()
        g = ops.Graph()
        with g:
            for i in range(N):
                fns[i % len(fns)]()
        for _ in range(3):
            g.launch()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            g.launch()
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 5 / N * 1e3
        print(f"{name:58s} {us:7.2f} us/launch" + (f"  {flops / us / 1e6:7.1f} TFLOP/s" if flops else ""), flush=True)
        g.close()


ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)
chain("spin(0) trivial kernel", lambda: [lambda: ops.debug_spin(0)])


def gemm_case(M, Nn, K, bm, bn, sp, var, nsets=None, res=False, bias=True):
    def mk():
        nset = nsets or max(2, int(400e6 / (Nn * K * 2)))
        nset = min(nset, 240)
        Ws = [rnd(Nn, K) for _ in range(nset)]
        a = [rnd(M, K), rnd(M, K)]
        o = [torch.empty(M, Nn, device=DEV, dtype=torch.float16) for _ in range(2)]
        b = rnd(Nn) if bias else None
        r = rnd(M, Nn) if res else None
        ops.plan_clear()
        ops.plan_set(0, M, Nn, K, 1, bm, bn, sp, var)
        return [(lambda i=i: ops.gemm(a[i % 2], Ws[i], o[i % 2], bias=b, res=r)) for i in range(nset)]
    return mk


for (M, Nn, K, bm, bn, sp, var) in [(4096, 320, 64, 64, 64, 1, 4), (4096, 320, 320, 64, 64, 1, 4), (4096, 320, 320, 64, 64, 1, 2),
                                    (4096, 320, 320, 64, 64, 1, 1), (4096, 320, 320, 128, 64, 1, 2), (4096, 320, 320, 64, 160, 1, 2),
                                    (4096, 320, 1280, 64, 64, 1, 4), (4096, 2560, 320, 128, 128, 1, 1),
                                    (1024, 640, 640, 64, 64, 1, 4), (1024, 640, 640, 64, 64, 2, 4), (256, 1280, 1280, 64, 64, 1, 4),
                                    (256, 1280, 1280, 64, 64, 4, 4), (64, 1280, 1280, 64, 64, 8, 4)]:
    chain(f"gemm M{M} N{Nn} K{K} tile {bm}x{bn} splits {sp} depth {var}", gemm_case(M, Nn, K, bm, bn, sp, var), 2.0 * M * Nn * K)
chain("gemm M4096 N320 K320 64x64 d4 +res, warm single weight", gemm_case(4096, 320, 320, 64, 64, 1, 4, nsets=1, res=True), 2.0 * 4096 * 320 * 320)
chain("gemm M4096 N320 K320 64x64 d4 no bias, warm", gemm_case(4096, 320, 320, 64, 64, 1, 4, nsets=1, bias=False), 2.0 * 4096 * 320 * 320)


def ln_case(M, C):
    def mk():
        x, g, b, o = rnd(M, C), rnd(C), rnd(C), torch.empty(M, C, device=DEV, dtype=torch.float16)
        return [lambda: ops.layernorm(x, g, b, o, M, C)]
    return mk


chain("layernorm 4096x320", ln_case(4096, 320))
chain("layernorm 256x1280", ln_case(256, 1280))
