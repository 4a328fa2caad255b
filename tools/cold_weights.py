#!/usr/bin/env python3
"""What cold weights cost a batch-1 layer: per-launch time with every cache flushed first (in situ a layer's weights always
come from HBM: 1.7 GB per UNet step), with the caches flushed and then ONLY the weights touched by a streaming read (what a
prefetch issued under the previous kernel would leave behind), and warm (back-to-back replay)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops

DEV = "cuda"
flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)


def timed(fn, pre, reps=12):
    ts = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return sum(ts[:len(ts) // 2]) / (len(ts) // 2)


def case(name, fn, w):
    fn(); fn()
    cold = timed(fn, lambda: flush.fill_(1))
    def touch():
        flush.fill_(1)
        w.view(torch.int32).sum()          # streaming read of the weights only
    touched = timed(fn, touch)
    warm = timed(fn, lambda: None)
    print(f"{name:44s} W {w.numel() * 2 / 1e6:6.1f} MB  cold {cold:7.1f}us  weights-touched {touched:7.1f}us  warm {warm:7.1f}us", flush=True)


def conv(B, H, Cin, Cout):
    x = torch.randn(B * H * H, Cin, device=DEV, dtype=torch.float16)
    w = torch.randn(Cout, 9 * Cin, device=DEV, dtype=torch.float16) * 0.01
    o = torch.empty(B * H * H, Cout, device=DEV, dtype=torch.float16)
    case(f"conv3x3 B{B} {H}x{H} {Cin}->{Cout}", lambda: ops.conv3x3(x, w, o, B, H, H, Cin, Cout), w)


def gemm(M, N, K, geglu=False):
    x = torch.randn(M, K, device=DEV, dtype=torch.float16)
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.01
    o = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
    b = torch.zeros(N, device=DEV, dtype=torch.float16)
    case(f"gemm M{M} N{N} K{K}{' geglu' if geglu else ''}", lambda: ops.gemm(x, w, o, bias=b, epilogue=1 if geglu else 0, img_rows=M), w)


if __name__ == "__main__":
    ws = torch.empty(256 << 18, dtype=torch.float32, device=DEV)
    ops.set_workspace(ws)
    conv(1, 8, 1280, 1280); conv(1, 16, 1280, 1280); conv(1, 16, 2560, 1280); conv(1, 32, 640, 640); conv(1, 32, 1280, 640); conv(1, 64, 320, 320)
    gemm(64, 1280, 1280); gemm(256, 1280, 1280); gemm(256, 10240, 1280, True); gemm(256, 1280, 5120); gemm(1024, 640, 640); gemm(4096, 320, 320)
    ops.set_workspace(None)
