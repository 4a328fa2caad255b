#!/usr/bin/env python3
"""What the epilogue options of the GroupNorm-fused halo conv cost in isolation (GPU box): plain output, + fused statistics,
+ residual, + both, on the AutoencoderKL 512^2 / 256^2 shapes at batch 8."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"


def case(B, H, Cin, Cout, xform=True, iters=6):
    x = torch.randn(B * H * H, Cin, device=DEV, dtype=torch.float16)
    w = torch.randn(Cout, 9 * Cin, device=DEV, dtype=torch.float16) * (9 * Cin) ** -0.5
    sc = torch.rand(B, Cin, device=DEV) + 0.5
    sh = torch.randn(B, Cin, device=DEV) * 0.1
    res = torch.randn(B * H * H, Cout, device=DEV, dtype=torch.float16)
    bias = torch.randn(Cout, device=DEV, dtype=torch.float16)
    o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
    st = ops.Stats(torch.zeros(ops.stats_floats(B * H * H, Cout, H * H), dtype=torch.float32, device=DEV))
    out = []
    for name, kw in (("bare", {}), ("+bias", dict(bias=bias)), ("+stats", dict(bias=bias, stats=st)), ("+res", dict(bias=bias, res=res)),
                     ("+stats+res", dict(bias=bias, stats=st, res=res))):
        if xform:
            fn = lambda: ops.conv3x3_gn(x, w, o, B, H, H, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True, **kw)
        else:
            fn = lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout, **kw)
        best = 1e9
        for r in range(3):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        out.append(f"{name} {best:7.1f}us")
    print(f"conv{'_gn' if xform else ''} B{B} {H}x{H} {Cin}->{Cout}: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        ops.set_staged_epilogue(int(sys.argv[1]))
        print(f"staged epilogue: {sys.argv[1]}")
    case(8, 512, 128, 128)
    case(8, 256, 256, 256)
    case(8, 512, 128, 128, xform=False)
    case(8, 128, 512, 512, xform=False)
    case(8, 64, 320, 320, xform=False)
    case(8, 32, 640, 640, xform=False)          # segmented accumulation: conv_halo_pipe_kernel<8,16,160,3,0,1,1[,1]>
    case(8, 16, 1280, 1280, xform=False)
