#!/usr/bin/env python3
"""rocprofv3 --pmc counter CSVs (one directory per counter and batch) -> the per-kernel traffic table bench.py reads.

  pmc_traffic.py out.json  B:COUNTER:dir  [B:COUNTER:dir ...]

Kernel names are normalised to the instantiation names the library reports ("igemm2_kernel<64, 64, 0, 4>")."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.check_profiles_fresh import CSRC, ROOT, sources_sha256  # noqa: E402


def norm(name):
    n = name.split("(")[0].strip()
    n = re.sub(r"^void\s+", "", n)
    return n


out = {"unit": "KB per launch (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate runs over one eager pass, tools/pmc_traffic.sh)",
       "correction": "gfx950: FETCH_SIZE counts 128-byte requests as 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM "
                     "section; calibration in profiles/r01_traffic_pmc.md)", "batch": {}}
for spec in sys.argv[2:]:
    B, counter, d = spec.split(":", 2)
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a = acc[norm(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    tab = out["batch"].setdefault(B, {})
    for k, (n, v) in acc.items():
        e = tab.setdefault(k, {})
        e[counter] = round(v / n, 1)
        e["n_" + counter] = n
for B in out["batch"]:
    out["batch"][B] = {k: v for k, v in out["batch"][B].items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
# the table describes the kernels of THIS tree: bench.py reports no traffic from a table taken on other sources
out["csrc_files"] = sorted(os.path.join(CSRC, f) for f in os.listdir(os.path.join(ROOT, CSRC)) if f.endswith((".hip", ".h", ".cpp")))
out["csrc_sha256"] = sources_sha256(out["csrc_files"])
json.dump(out, open(sys.argv[1], "w"), indent=1)
print({B: len(t) for B, t in out["batch"].items()})
