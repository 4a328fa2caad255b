#!/usr/bin/env python3
"""Average a PMC counter per kernel name from a rocprofv3 counter_collection CSV: pmc_avg.py <dir> <counter>."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2]:
        a = acc[r["Kernel_Name"].split("(")[0]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k[:70]:70s} n={n:6d} avg={v / n:14.1f}")
