#!/usr/bin/env python3
"""Per-sampler-pass kernel breakdown from a rocprofv3 kernel-trace CSV.

  pass_breakdown.py <rocprof output dir> [rows]

One pass = one hipGraph replay of the sampler = the kernels between two consecutive latents_pool8 launches (once per
replay, between the last scheduler step and the VAE decode: the window is a cyclic shift of one replay).  Run it on a trace of `bench.py --no-roofline --no-extra --no-cpu-baseline`:
the roofline leg replays the dominant kernel back to back outside any pass and would otherwise land inside the window
(round 1's breakdowns were 2x off for that kernel).  The window is taken from the timed region (the last replays) and
is checked: every one of the last three windows must hold the same number of kernels."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ends = [i for i, k in enumerate(ks) if "latents_pool8" in k[2]]      # once per replay (any once-per-pass kernel delimits a window)
assert len(ends) >= 5, "need at least 4 passes in the trace"
wins = [(ends[-k - 1] + 1, ends[-k] + 1) for k in (1, 2, 3)]
sizes = [b - a for a, b in wins]
assert len(set(sizes)) == 1, f"the last three windows differ in kernel count {sizes}: not clean graph replays"
a, b = wins[1]
seg = ks[a:b]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    n = n.split("(")[0][:64]
    tot[n][0] += 1
    tot[n][1] += e - s
print("pass kernels", len(seg), "wall_us", (seg[-1][1] - seg[0][0]) / 1e3, "busy_us", sum(v[1] for v in tot.values()) / 1e3)
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{n:66s} {c:5d} {t / 1e3:9.1f}us  avg {t / c / 1e3:7.2f}")
