#!/usr/bin/env python3
"""Launch-by-launch listing of ONE sampler pass from a rocprofv3 kernel-trace CSV (same window rule as pass_breakdown.py).

  pass_sequence.py <rocprof output dir> > sequence.txt

Columns: index, start offset in the pass (us), duration (us), gap to the previous kernel's end (us), grid (workgroups),
kernel.  Used to attribute time to layers (the launch order is model.py's) and to see the inter-kernel gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1),
             int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1))
            for r in rows)
ends = [i for i, k in enumerate(ks) if "latents_pool8" in k[2]]      # once per replay (any once-per-pass kernel delimits a window)
assert len(ends) >= 4, "need at least 3 passes in the trace"
a, b = ends[-3] + 1, ends[-2] + 1
seg = ks[a:b]
t0, prev = seg[0][0], seg[0][0]
gaps = 0
for i, (s, e, n, g, wg) in enumerate(seg):
    n = n.split("(")[0].replace("void ", "")[:60]
    print(f"{i:5d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} {(s - prev) / 1e3:6.2f} {g // max(wg, 1):7d} {n}")
    gaps += max(0, s - prev)
    prev = max(prev, e)
print(f"# kernels {len(seg)} wall_us {(seg[-1][1] - t0) / 1e3:.1f} sum_gaps_us {gaps / 1e3:.1f}")
