#!/usr/bin/env python3
"""Per-sampler-pass kernel breakdown from a rocprofv3 kernel trace CSV (one graph replay = one pass)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ends = [i for i, k in enumerate(ks) if "conv_smalln_kernelILi16" in k[2]]
a, b = ends[-3] + 1, ends[-2] + 1
seg = ks[a:b]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    n = n.split("(")[0][:64]
    tot[n][0] += 1
    tot[n][1] += e - s
print("pass kernels", len(seg), "wall_us", (seg[-1][1] - seg[0][0]) / 1e3, "busy_us", sum(v[1] for v in tot.values()) / 1e3)
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{n:66s} {c:5d} {t / 1e3:9.1f}us  avg {t / c / 1e3:7.2f}")
