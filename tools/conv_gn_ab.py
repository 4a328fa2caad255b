#!/usr/bin/env python3
"""A/B of the GroupNorm-fused halo conv with / without the next-chunk halo prefetch (GPU box): bit-equality, then interleaved
timing rounds in one process on the AutoencoderKL shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"


def case(B, H, Cin, Cout, rounds=4, iters=6):
    x = torch.randn(B * H * H, Cin, device=DEV, dtype=torch.float16)
    w = torch.randn(Cout, 9 * Cin, device=DEV, dtype=torch.float16) * (9 * Cin) ** -0.5
    sc = torch.rand(B, Cin, device=DEV) + 0.5
    sh = torch.randn(B, Cin, device=DEV) * 0.1
    outs, res = {}, {}
    for r in range(rounds):
        for name, on in (("nopf", 0), ("pf", 1), ("plain", -1)):
            o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
            if on >= 0:
                ops.set_halo_prefetch(on)
                fn = lambda: ops.conv3x3_gn(x, w, o, B, H, H, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True)
            else:
                fn = lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout)
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / iters * 1e3)
            outs[name] = o
    ops.set_halo_prefetch(1)
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"conv_gn B{B} {H}x{H} {Cin}->{Cout}: " + "  ".join(f"{n} {min(t):8.1f}us ({fl / min(t) / 1e6:5.0f} TF)" for n, t in res.items()) +
          f"  pf==nopf: {torch.equal(outs['pf'], outs['nopf'])}", flush=True)


if __name__ == "__main__":
    for shp in ((8, 512, 128, 128), (8, 256, 256, 256), (1, 512, 128, 128), (1, 256, 256, 256), (8, 512, 256, 128), (2, 512, 128, 128)):
        case(*shp)
