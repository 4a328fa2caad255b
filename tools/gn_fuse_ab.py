#!/usr/bin/env python3
"""GroupNorm(+SiLU) inside the conv's halo staging against its own pass + a plain conv, on the AutoencoderKL's large layers at
batch 8 / 1 with the shipped launch plans (interleaved rounds, HIP events): is the 64 MB fusion threshold still right?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
ws = torch.empty(256 << 20, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)


def timed(fn, iters=6):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, Cin, Cout, res) in ((8, 512, 128, 128, False), (8, 512, 128, 128, True), (8, 256, 256, 256, False), (8, 256, 256, 256, True),
                               (8, 128, 512, 512, False), (1, 512, 128, 128, False), (1, 256, 256, 256, False)):
    M = B * H * H
    x, w = rnd(M, Cin), rnd(Cout, 9 * Cin) * (9 * Cin) ** -0.5
    xin, wprev = rnd(M, 64), rnd(Cin, 9 * 64) * (9 * 64) ** -0.5
    st = ops.Stats(torch.zeros(ops.stats_floats(M, Cin, H * H), dtype=torch.float32, device=DEV))
    ops.conv3x3(xin, wprev, x, B, H, H, 64, Cin, stats=st)          # producer with fused statistics
    gamma, beta = rnd(Cin), rnd(Cin)
    hn, o1, o2 = torch.empty_like(x), torch.empty(M, Cout, device=DEV, dtype=torch.float16), torch.empty(M, Cout, device=DEV, dtype=torch.float16)
    r = rnd(M, Cout) if res else None
    st_o = ops.Stats(torch.zeros(ops.stats_floats(M, Cout, H * H), dtype=torch.float32, device=DEV))
    gws = torch.empty(ops.groupnorm_ws_bytes(B, H * H, Cin) // 4 + 16, dtype=torch.float32, device=DEV)

    def fused():
        sc, sh = ops.groupnorm_tables_from_stats(gamma, beta, B, H * H, Cin, st, gws)
        ops.conv3x3_gn(x, w, o1, B, H, H, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True, res=r, stats=st_o)

    def separate():
        ops.groupnorm_from_stats(x, gamma, beta, hn, B, H * H, Cin, st, gws)
        ops.conv3x3(hn, w, o2, B, H, H, Cin, Cout, res=r, stats=st_o)

    def apply_only():
        ops.groupnorm_from_stats(x, gamma, beta, hn, B, H * H, Cin, st, gws)

    tf, tsep, ta = [], [], []
    for _ in range(4):
        tf.append(timed(fused)); tsep.append(timed(separate)); ta.append(timed(apply_only))
    print(f"B{B} {H}x{H} {Cin}->{Cout}{' +res' if res else ''}: fused {min(tf):7.1f} us | finalize+apply+plain conv {min(tsep):7.1f} us "
          f"(finalize+apply alone {min(ta):6.1f}) | equal bits: {torch.equal(o1, o2)}", flush=True)
