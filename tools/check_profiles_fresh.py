#!/usr/bin/env python3
"""Counter evidence must describe the kernels that ship.  Every entry of profiles/<round>_pmc_summary.json (and of
profiles/<round>_traffic.json) records the sha256 of the kernel's source files at the time the counters were taken
(tools/pmc_summary.py / tools/pmc_traffic.py); this check recomputes it from the tree and FAILS on a mismatch -- a kernel was
edited after its counters were collected: re-run tools/pmc_round4.sh on the GPU box, then tools/pmc_summary.py.

  python tools/check_profiles_fresh.py [round]        exit 0 = fresh, 1 = stale entries listed

bench.py uses `stale_entries` too: counters of a stale entry are reported as null with the reason, never silently."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("stable-diffusion-1.5-lcm-onnx-rknn2_amd", "csrc")
COMMON = ("common.h",)
_BY_PREFIX = (("igemm", ("igemm.hip", "igemm_common.h")), ("splitk_reduce", ("igemm.hip", "igemm_common.h")),
              ("conv_halo", ("conv_halo.hip", "igemm_common.h")), ("attn", ("attention.hip",)),
              ("mlp_geglu", ("mlp_fused.hip", "igemm_common.h")), ("gn_", ("norm.hip",)), ("layernorm", ("norm.hip",)),
              ("conv_c4", ("misc.hip",)), ("conv_fewout", ("misc.hip",)))


def kernel_sources(kernel_name):
    """Source files (relative to the repository) that define the kernel `kernel_name` (as rocprofv3 prints it)."""
    k = kernel_name.replace("void ", "").strip()
    k = k.split("(")[0]
    if k.startswith("_Z"):                       # mangled: _Z<len><name>...
        k = k.lstrip("_Z0123456789")
    for pre, files in _BY_PREFIX:
        if k.startswith(pre):
            return [os.path.join(CSRC, f) for f in files + COMMON]
    return [os.path.join(CSRC, f) for f in COMMON]


def sources_sha256(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def stale_entries(summary_path):
    """-> {kernel: reason} for entries whose recorded source hash is absent or no longer matches the tree."""
    with open(summary_path) as f:
        tab = json.load(f)
    out = {}
    for k, v in tab.items():
        if not isinstance(v, dict):
            continue
        want = v.get("source_sha256")
        if not want:
            out[k] = "no source hash recorded"
            continue
        have = sources_sha256(v.get("sources") or kernel_sources(k))
        if have != want:
            out[k] = f"sources changed since the counters were taken ({', '.join(os.path.basename(s) for s in (v.get('sources') or kernel_sources(k)))})"
    return out


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    p = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
    if not os.path.exists(p):
        print(f"{p}: missing")
        return 1
    bad = stale_entries(p)
    t = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
    if os.path.exists(t):
        doc = json.load(open(t))
        if not doc.get("csrc_sha256") or sources_sha256(doc.get("csrc_files", [])) != doc["csrc_sha256"]:
            bad[os.path.relpath(t, ROOT)] = "taken on other kernel sources than this tree's"
    for k, why in bad.items():
        print(f"STALE {k}: {why}")
    if not bad:
        print(f"{os.path.relpath(p, ROOT)}: {len(json.load(open(p)))} entries, all taken on the sources in the tree")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
