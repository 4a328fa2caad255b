#!/usr/bin/env python3
"""profiles/<round>_pmc_<tag>.txt (tools/pmc_kernel.sh) -> profiles/<round>_pmc_summary.json (round = argv[1], default r04): per
kernel the MFMA-busy fraction and the HBM-side traffic, derived as MI355X_MICROARCH.md prescribes:
  mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)      [busy cycles per SIMD / kernel cycles]
  hbm_bytes      = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (gfx950: FETCH_SIZE reads half the bytes of a wide coalesced stream)
  hbm_gbs        = hbm_bytes / average kernel duration (rocprofv3 --kernel-trace --stats of the same target)
GRBM_GUI_ACTIVE reads high on dispatches far below 0.3 ms (the guide's DVFS note): for those the fraction is also given against
duration x 2.1 GHz (`mfma_busy_frac_at_2p1ghz`).
Every entry records the sha256 of the kernel's sources as they are NOW (`source_sha256`): run this right after the PMC passes, on
the tree they were taken on.  bench.py nulls counters whose sources have changed since; tools/check_profiles_fresh.py (and the
CPU test of the same name) fails on them."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.check_profiles_fresh import kernel_sources, sources_sha256  # noqa: E402
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"
TAGS = {"gemm64_b1": "gemm M4096 N320 K320 (batch-1 transformer GEMM, 64x64 tile, 4-stage ring)",
        "convgn_b8": "GroupNorm-fused conv3x3 8 x 512^2 x 128 -> 128 (AutoencoderKL top level, batch 8)",
        "attn2_b8": "self-attention 8 images x 8 heads, S = 4096, d = 40 (UNet level 0, batch 8; key-split form, 16 waves)",
        "attn2_b1": "self-attention 1 image x 8 heads, S = 4096, d = 40 (UNet level 0, batch 1; key-split form, 8 waves)",
        "attn2_b8_unsplit": "the same launch with lcm_set_attention_ksplit(0)",
        "attn2_b1_unsplit": "the same launch with lcm_set_attention_ksplit(0)",
        "convgnres_b8": "the same convolution with a residual (conv2 of a ResnetBlock2D): staged epilogue",
        "mlp_b8": "fused FeedForward (norm3 -> ff.net.0 -> GEGLU -> ff.net.2 -> + h), 32768 rows x C 320 (UNet level 0, batch 8)",
        "ff1_b8": "LayerNorm-folded GEGLU projection M32768 N2560 K320 (the first of the two launches the fused kernel replaces)",
        "ff2_b8": "ff.net.2 + residual M32768 N320 K1280 (the second of the two launches)",
        "o1_b8": "attn.to_out + residual M32768 N320 K320 (batch 8)"}
out = {}
for tag, what in TAGS.items():
    p = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_{tag}.txt")
    if not os.path.exists(p):
        continue
    txt = open(p).read()
    vals, cur, kern = {}, None, None
    for line in txt.splitlines():
        if line.startswith("-- "):
            cur = line[3:].strip()
        elif cur and "n=" in line and "at::native" not in line:
            m = re.match(r"(?:void )?(.+?)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
            if m:
                kern = m.group(1).strip()
                vals[cur] = float(m.group(3))
                cur = None
    m = re.search(r'^"void ([^"]+?)\(.*?",(\d+),(\d+),([\d.]+)', txt, re.M)
    avg_us = float(m.group(4)) / 1e3 if m else None
    busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
    gui = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    hbm = (2.0 * vals.get("FETCH_SIZE", 0.0) + vals.get("WRITE_SIZE", 0.0)) * 1024.0
    if kern in out:
        kern = f"{kern} [{tag}]"          # a second shape of the same instantiation keeps its own entry
    out[kern] = dict(what=what, source=f"profiles/{ROUND}_pmc_{tag}.txt", sources=kernel_sources(kern.split(' [')[0]), source_sha256=sources_sha256(kernel_sources(kern.split(' [')[0])), avg_us=round(avg_us, 2) if avg_us else None,
                     mfma_busy_cycles_per_simd=round(busy), kernel_cycles=round(gui),
                     mfma_busy_frac=round(busy / gui, 4) if gui else None,
                     mfma_busy_frac_at_2p1ghz=round(busy / (avg_us * 2100.0), 4) if avg_us else None,
                     effective_clock_ghz=round(gui / (avg_us * 1e3), 3) if avg_us else None,
                     fetch_size_kb=vals.get("FETCH_SIZE"), write_size_kb=vals.get("WRITE_SIZE"), hbm_bytes_per_launch=round(hbm),
                     hbm_gbs=round(hbm / (avg_us * 1e-6) / 1e9, 1) if avg_us else None,
                     valu_active_frac=round(vals.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024.0 / gui, 4) if gui else None,
                     wait_inst_frac_of_wave_cycles=round(vals.get("SQ_WAIT_INST_ANY", 0.0) / vals["SQ_WAVE_CYCLES"], 4) if vals.get("SQ_WAVE_CYCLES") else None,
                     lds_bank_conflict_frac=round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / vals["SQ_LDS_IDX_ACTIVE"], 4) if vals.get("SQ_LDS_IDX_ACTIVE") else None)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(k, {a: b for a, b in v.items() if a not in ("what", "source")})
