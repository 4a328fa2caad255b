#!/usr/bin/env python3
"""Reads the s_memtime stamps of a -DLCM_TRACE build of conv_halo_pipe_kernel (debug only)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
from sdlcm_amd import ops, lib
DEV = "cuda"
rnd = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16)
ws = torch.empty(64 << 18, dtype=torch.float32, device=DEV)
ops.set_workspace(ws)
L = lib.load()
flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
for (B, H, Cin, Cout, bm, bn, sp) in [(1, 64, 320, 320, 64, 64, 1), (1, 32, 640, 640, 128, 64, 5), (1, 16, 1280, 1280, 128, 64, 10)]:
    M = B * H * H
    x, w, o, bias = rnd(M, Cin), rnd(Cout, 9 * Cin), torch.empty(M, Cout, device=DEV, dtype=torch.float16), rnd(Cout)
    ops.plan_clear(); ops.plan_set(2, M, Cout, 9 * Cin, H << 1, bm, bn, sp, 2)
    for rep in range(3):
        flush.fill_(rep)
        # keep activations warm as in situ
        x.add_(0)
        ops.conv3x3(x, w, o, B, H, H, Cin, Cout, bias=bias)
        torch.cuda.synchronize()
    buf = np.zeros(64 * 64, dtype=np.uint64)
    assert L.lcm_debug_get_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(64, 64).astype(np.int64)
    valid = t[:, 0] > 0
    t = t[valid]
    real = (t[:, 62] - t[:, 63]) / 100.0          # us (100 MHz)
    cyc = t[:, 61] - t[:, 0]
    nst = 0
    while 3 + 2 * nst < 60 and t[0, 3 + 2 * nst] > 0 and nst < 28:
        nst += 1
    print(f"conv {H}x{H} {Cin}->{Cout} tile {bm}x{bn} s{sp}: {len(t)} traced WGs, steps/WG traced {nst}, WG lifetime us: "
          f"min {real.min():.2f} med {np.median(real):.2f} max {real.max():.2f}; clock ~{np.median(cyc / real):.0f} cyc/us")
    start_spread = (t[:, 63] - t[:, 63].min()) / 100.0
    print(f"   WG start spread us: med {np.median(start_spread):.2f} max {start_spread.max():.2f};  end spread: {((t[:, 62] - t[:, 63].min()) / 100.0).max():.2f}")
    mhz = np.median(cyc / real)
    pro = (t[:, 1] - t[:, 0]) / mhz
    first = (t[:, 2] - t[:, 1]) / mhz
    print(f"   prologue (setup+issue) med {np.median(pro):.2f} us; wait for first tile med {np.median(first):.2f} us")
    waits = [(t[:, 2 + 2 * k] - t[:, 1 + 2 * k]) / mhz for k in range(1, nst)]
    comps = [(t[:, 3 + 2 * k] - t[:, 2 + 2 * k]) / mhz for k in range(nst)]
    print("   per-step wait+barrier med us: " + " ".join(f"{np.median(wv):.2f}" for wv in waits))
    print("   per-step issue+lds+mfma med us: " + " ".join(f"{np.median(c):.2f}" for c in comps))
    epi = (t[:, 61] - t[:, 60]) / mhz
    print(f"   epilogue med {np.median(epi):.2f} us")
