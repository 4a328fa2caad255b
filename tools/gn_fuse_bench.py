#!/usr/bin/env python3
"""Is GroupNorm-apply(+SiLU) cheaper inside the conv's halo staging (XFORM) than as its own pass?  Per shape:
best plain conv + (finalize + apply) vs best XFORM conv + finalize, cold-cache timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops
from sdlcm_amd.autotune import _time_cold

DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
shapes = [(8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 64, 640, 320), (8, 512, 128, 128), (8, 256, 256, 256),
          (8, 128, 512, 512), (1, 64, 320, 320), (1, 32, 640, 640), (1, 512, 128, 128)]
ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)
for (B, H, Cin, Cout) in shapes:
    M = B * H * H
    x, w = rnd(M, Cin), rnd(Cout, 9 * Cin) * (9 * Cin) ** -0.5
    wprev = rnd(Cin, 9 * 64) * (9 * 64) ** -0.5
    xin = rnd(M, 64)
    st = ops.Stats(torch.zeros(ops.stats_floats(M, Cin), dtype=torch.float32, device=DEV))
    ops.conv3x3(xin, wprev, x, B, H, H, 64, Cin, stats=st)          # producer with fused statistics
    gamma, beta = rnd(Cin), rnd(Cin)
    hn = torch.empty_like(x)
    o = torch.empty(M, Cout, device=DEV, dtype=torch.float16)
    gws = torch.empty(ops.groupnorm_ws_bytes(B, H * H, Cin) // 4 + 16, dtype=torch.float32, device=DEV)
    scale = torch.empty(B, Cin, dtype=torch.float32, device=DEV)
    shift = torch.empty(B, Cin, dtype=torch.float32, device=DEV)
    ops.groupnorm_affine(x, gamma, beta, scale, shift, B, H * H, Cin, gws)
    t_gn = _time_cold(lambda: ops.groupnorm_from_stats(x, gamma, beta, hn, B, H * H, Cin, st, gws), 8)
    best = {}
    for xf in (0, 1):
        for bm in (128, 64):
            for bn in (160, 128, 64):
                if Cout % bn:
                    continue
                for sp in ((1,) if M >= 32768 else (1, 2, 4)):
                    if sp > Cin // 64:
                        continue
                    ops.plan_clear()
                    ops.plan_set(2, M, Cout, 9 * Cin, (H << 1) | xf, bm, bn, sp, 1 if M >= 32768 else -1)
                    if xf:
                        fn = lambda: ops.conv3x3_gn(x, w, o, B, H, H, Cin, Cout, gn_scale=scale, gn_shift=shift, silu=True)
                    else:
                        fn = lambda: ops.conv3x3(hn, w, o, B, H, H, Cin, Cout)
                    t = _time_cold(fn, 6)
                    if xf not in best or t < best[xf][0]:
                        best[xf] = (t, bm, bn, sp)
    fl = 2.0 * M * Cout * 9 * Cin
    print(f"B{B} {H}x{H} {Cin}->{Cout}: finalize+apply {t_gn * 1e3:7.1f}us | plain {best[0][0] * 1e3:7.1f}us {best[0][1:]} {fl / best[0][0] / 1e9:5.0f}TF"
          f" | xform {best[1][0] * 1e3:7.1f}us {best[1][1:]} {fl / best[1][0] / 1e9:5.0f}TF | separate {1e3 * (t_gn + best[0][0]):7.1f} fused {1e3 * (best[1][0]) + 5:7.1f}", flush=True)
