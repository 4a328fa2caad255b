#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels on representative SD1.5 shapes (GPU box only).
Prints TFLOP/s (MFMA kernels) or GB/s (HBM kernels) per shape; used to steer kernel work."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s):
    return torch.randn(*s, device=DEV, dtype=torch.float16)


def splitk_sweep():
    """B=1 UNet conv shapes: no split vs split-K under a few heuristics."""
    ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
    shapes = ((64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280),
              (8, 1280, 1280), (8, 2560, 1280))
    cfgs = (("nosplit", None), ("t384/s16/m256", (384, 16, 256)), ("t256/s8/m256", (256, 8, 256)),
            ("t512/s32/m256", (512, 32, 256)), ("t384/s16/m160", (384, 16, 160)), ("t768/s32/m384", (768, 32, 384)))
    for (H, Cin, Cout) in shapes:
        x, w = rnd(H * H, Cin), rnd(Cout, 9 * Cin)
        o = torch.empty(H * H, Cout, device=DEV, dtype=torch.float16)
        line = f"conv B1 {H}x{H} {Cin}->{Cout}:"
        for name, cfg in cfgs:
            if cfg is None:
                ops.set_workspace(None)
            else:
                ops.set_workspace(ws)
                ops.set_tuning(*cfg)
            t = timeit(lambda: ops.conv3x3(x, w, o, 1, H, H, Cin, Cout))
            line += f"  {name} {t * 1e6:.1f}us"
        print(line)
    ops.set_workspace(None)
    ops.set_tuning(384, 16, 256)


def variant_sweep():
    ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    convs = ((1, 64, 320, 320), (1, 16, 1280, 1280), (1, 8, 1280, 1280), (8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280),
             (8, 8, 1280, 1280), (8, 64, 960, 320), (1, 128, 512, 512), (1, 256, 256, 256), (1, 512, 128, 128), (8, 128, 512, 512))
    gemms = ((4096, 320, 320), (4096, 2560, 320), (4096, 320, 1280), (1024, 640, 640), (256, 1280, 1280), (32768, 320, 320),
             (32768, 2560, 320), (32768, 320, 1280), (8192, 5120, 640))
    for (B, H, Cin, Cout) in convs:
        x, w = rnd(B * H * H, Cin), rnd(Cout, 9 * Cin)
        o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        line = f"conv B{B} {H}x{H} {Cin}->{Cout}:"
        for v in (0, 2, 1):
            ops.set_kernel_variant(v)
            t = timeit(lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout))
            line += f"  v{v} {t * 1e6:7.1f}us {fl / t / 1e12:6.0f}TF"
        print(line)
    for (M, N, K) in gemms:
        a, w = rnd(M, K), rnd(N, K)
        o = torch.empty(M, N, device=DEV, dtype=torch.float16)
        line = f"gemm M{M} N{N} K{K}:"
        for v in (0, 2, 1):
            ops.set_kernel_variant(v)
            t = timeit(lambda: ops.gemm(a, w, o))
            line += f"  v{v} {t * 1e6:7.1f}us {2.0 * M * N * K / t / 1e12:6.0f}TF"
        print(line)
    ops.set_kernel_variant(-1)
    ops.set_workspace(None)


def halo_sweep():
    ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    convs = ((1, 64, 320, 320), (1, 32, 640, 640), (1, 16, 1280, 1280), (1, 8, 1280, 1280), (8, 64, 320, 320), (8, 32, 640, 640),
             (8, 16, 1280, 1280), (8, 8, 1280, 1280), (8, 64, 960, 320), (8, 16, 2560, 1280), (1, 128, 512, 512), (1, 256, 256, 256),
             (1, 512, 128, 128), (8, 128, 512, 512), (8, 256, 256, 256))
    for (B, H, Cin, Cout) in convs:
        x, w = rnd(B * H * H, Cin), rnd(Cout, 9 * Cin)
        o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
        sc = torch.ones(B, Cin, device=DEV)
        sh = torch.zeros(B, Cin, device=DEV)
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        line = f"conv B{B} {H}x{H} {Cin}->{Cout}:"
        for name, impl in (("igemm", 0), ("halo", 1)):
            ops.set_conv_impl(impl)
            t = timeit(lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout))
            line += f"  {name} {t * 1e6:7.1f}us {fl / t / 1e12:6.0f}TF"
        t = timeit(lambda: ops.conv3x3_gn(x, w, o, B, H, H, Cin, Cout, gn_scale=sc, gn_shift=sh, silu=True))
        line += f"  halo+gn {t * 1e6:7.1f}us {fl / t / 1e12:6.0f}TF"
        print(line)
    ops.set_conv_impl(1)
    ops.set_workspace(None)


def persist_sweep():
    gemms = ((32768, 320, 320, 0), (32768, 960, 320, 0), (32768, 2560, 320, 1), (32768, 320, 1280, 0), (8192, 640, 640, 0),
             (8192, 1920, 640, 0), (8192, 5120, 640, 1), (8192, 640, 2560, 0), (2048, 1280, 1280, 0), (2048, 10240, 1280, 1),
             (2048, 1280, 5120, 0), (4096, 320, 320, 0), (4096, 2560, 320, 1))
    for (M, N, K, epi) in gemms:
        a, w = rnd(M, K), rnd(N, K)
        o = torch.empty(M, N // 2 if epi else N, device=DEV, dtype=torch.float16)
        line = f"gemm M{M} N{N} K{K} epi{epi}:"
        for on in (0, 1):
            ops.set_persist_n(on)
            t = timeit(lambda: ops.gemm(a, w, o, epilogue=epi), iters=50)
            line += f"  persist{on} {t * 1e6:7.1f}us {2.0 * M * N * K / t / 1e12:6.0f}TF"
        print(line)
    ops.set_persist_n(0)


def main():
    print("device", torch.cuda.get_device_name(0))
    if len(sys.argv) > 1 and sys.argv[1] == "persist":
        return persist_sweep()
    if len(sys.argv) > 1 and sys.argv[1] == "halo":
        return halo_sweep()
    if len(sys.argv) > 1 and sys.argv[1] == "splitk":
        return splitk_sweep()
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        return variant_sweep()
    rows = []
    for B in (1, 8):
        for (H, Cin, Cout) in ((64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280), (64, 960, 320),
                               (64, 512, 512), (128, 512, 512), (256, 256, 256), (512, 128, 128)):
            if B == 8 and H >= 256:
                continue
            x = rnd(B * H * H, Cin)
            w = rnd(Cout, 9 * Cin)
            o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
            t = timeit(lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout))
            fl = 2.0 * B * H * H * Cout * 9 * Cin
            rows.append(("conv3x3", f"B{B} {H}x{H} {Cin}->{Cout}", t * 1e6, fl / t / 1e12, "TF"))
        for (M, N, K) in ((4096, 320, 320), (4096, 2560, 320), (4096, 320, 1280), (1024, 5120, 640), (256, 10240, 1280),
                          (4096, 960, 320)):
            a, w = rnd(B * M, K), rnd(N, K)
            o = torch.empty(B * M, N, device=DEV, dtype=torch.float16)
            t = timeit(lambda: ops.gemm(a, w, o))
            rows.append(("gemm", f"M{B*M} N{N} K{K}", t * 1e6, 2.0 * B * M * N * K / t / 1e12, "TF"))
        for (S, Sk, d) in ((4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (4096, 77, 40)):
            C = 8 * d
            q, k, v = rnd(B * S, C), rnd(B * Sk, C), rnd(B * Sk, C)
            o = torch.empty(B * S, C, device=DEV, dtype=torch.float16)
            t = timeit(lambda: ops.attention(q, k, v, o, B, 8, S, Sk, d, ldq=C, ldk=C, ldv=C, ldo=C))
            rows.append(("attn", f"B{B} S{S}x{Sk} d{d}", t * 1e6, 4.0 * B * 8 * S * Sk * d / t / 1e12, "TF"))
        for (HW, C) in ((4096, 320), (1024, 640), (65536, 256), (262144, 128)):
            if B == 8 and HW > 65536:
                continue
            x = rnd(B * HW, C)
            g, b2 = rnd(C), rnd(C)
            o = torch.empty_like(x)
            ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C) // 4, device=DEV)
            t = timeit(lambda: ops.groupnorm(x, g, b2, o, B, HW, C, ws))
            rows.append(("groupnorm", f"B{B} HW{HW} C{C}", t * 1e6, 3.0 * B * HW * C * 2 / t / 1e9, "GB/s"))
    x = rnd(512 * 512, 128)
    w = rnd(3, 9 * 128)
    o8 = torch.empty(512 * 512, 3, device=DEV, dtype=torch.uint8)
    t = timeit(lambda: ops.conv3x3_smalln(x, w, o8, 1, 512, 512, 128, 3, mode=1))
    rows.append(("conv_smalln", "512x512 128->3 u8", t * 1e6, 512 * 512 * 128 * 2 / t / 1e9, "GB/s"))
    for r in rows:
        print(f"{r[0]:12s} {r[1]:28s} {r[2]:10.1f} us  {r[3]:9.1f} {r[4]}")


if __name__ == "__main__":
    main()
