#!/usr/bin/env python3
"""Halo-conv kernel variants (plan variant 1 single-buffer / 2 ring of 3 / 3 row of taps / 4 double buffer) on batched shapes,
one process, interleaved; checks bit-equality across variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"


def case(B, H, Cin, Cout, tiles=((128, 128),), variants=(1, 2, 4), ups=0):
    Hi = H // 2 if ups else H
    x = torch.randn(B * Hi * Hi, Cin, device=DEV, dtype=torch.float16)
    taps = 4 if ups == 2 else 9
    w = torch.randn((4 if ups == 2 else 1) * Cout, taps * Cin, device=DEV, dtype=torch.float16) * (9 * Cin) ** -0.5
    M = B * H * H
    ref, out = None, []
    for bm, bn in tiles:
        if Cout % bn:
            continue
        for v in variants:
            ops.plan_clear()
            ops.plan_set(2, M, Cout, taps * Cin, (H << 1), bm, bn, 1, v)
            o = torch.empty(M, Cout, device=DEV, dtype=torch.float16)
            fn = lambda: ops.conv3x3(x, w, o, B, Hi, Hi, Cin, Cout, ups=ups)
            fn(); fn()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            if ref is None:
                ref = o.clone()
            out.append((best, f"{bm}x{bn}/v{v} {best:7.1f}us{'' if torch.equal(ref, o) else ' DIFF'}"))
    ops.plan_reset()
    fl = 2.0 * M * Cout * taps * Cin
    out.sort()
    print(f"conv B{B} {H}x{H} {Cin}->{Cout} ups{ups}: best {fl / out[0][0] / 1e6:5.0f} TF | " + "  ".join(t for _, t in out), flush=True)


if __name__ == "__main__":
    case(8, 512, 128, 128)
    case(8, 256, 256, 256)
    case(8, 128, 512, 512)
    case(8, 64, 512, 512, tiles=((128, 128), (128, 64)))
    case(8, 64, 320, 320, tiles=((128, 64), (64, 64)))
    case(8, 32, 640, 640, tiles=((128, 128), (128, 64)))
    case(8, 512, 256, 256, ups=2)
    case(1, 512, 128, 128, tiles=((128, 128), (64, 128)))
    case(1, 128, 512, 512, tiles=((128, 128), (128, 64)))
