#!/usr/bin/env python3
"""conv3x3 inside a dependent hipGraph chain with rotating (cold) weights: tile / split-K / pipeline sweep at batch 1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
st = torch.cuda.Stream()
N = 120
ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)


def chain(fns):
    with torch.cuda.stream(st):
        for f in fns[:3]:
            f()
        st.synchronize()
        g = ops.Graph()
        with g:
            for i in range(N):
                fns[i % len(fns)]()
        for _ in range(2):
            g.launch()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            g.launch()
        e1.record(st)
        e1.synchronize()
        g.close()
        return e0.elapsed_time(e1) / 4 / N * 1e3


for (B, H, Cin, Cout) in [(1, 64, 320, 320), (1, 32, 640, 640), (1, 16, 1280, 1280), (1, 8, 1280, 1280), (1, 32, 1280, 640)]:
    M = B * H * H
    nset = min(120, max(2, int(400e6 / (Cout * 9 * Cin * 2))))
    Ws = [rnd(Cout, 9 * Cin) for _ in range(nset)]
    x = [rnd(M, Cin), rnd(M, Cin)]
    o = [torch.empty(M, Cout, device=DEV, dtype=torch.float16) for _ in range(2)]
    bias = rnd(Cout)
    fl = 2.0 * M * Cout * 9 * Cin
    res = []
    for bm in (128, 64):
        if H * H < bm or (H <= 8 and bm == 128):
            continue
        for bn in (160, 128, 64):
            if Cout % bn:
                continue
            for sp in (1, 2, 3, 4, 5, 8, 10, 20):
                if sp > Cin // 64:
                    continue
                for var in (1, 2, 3):
                    ops.plan_clear()
                    ops.plan_set(2, M, Cout, 9 * Cin, H << 1, bm, bn, sp, var)
                    fns = [(lambda i=i: ops.conv3x3(x[i % 2], Ws[i], o[i % 2], B, H, H, Cin, Cout, bias=bias)) for i in range(nset)]
                    res.append((chain(fns), bm, bn, sp, var))
    res.sort()
    print(f"conv B{B} {H}x{H} {Cin}->{Cout} ({fl / 1e9:.1f} GF, weights {Cout * 9 * Cin * 2 / 1e6:.1f} MB): " +
          "  ".join(f"{t:.1f}us[{bm}x{bn} s{sp} {v}]" for t, bm, bn, sp, v in res[:6]) +
          f"  ... worst {res[-1][0]:.1f}us", flush=True)
