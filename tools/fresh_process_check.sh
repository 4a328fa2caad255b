#!/bin/bash
# Same request + seed => same PNG bytes in every process and under every launch-plan setting (GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fresh
python tools/png_hash.py > gpurun_out/fresh/p1.txt 2>/dev/null || exit 1
python tools/png_hash.py > gpurun_out/fresh/p2.txt 2>/dev/null || exit 2
LCM_TUNED_PLANS=0 python tools/png_hash.py > gpurun_out/fresh/p3_no_shipped_tile_plans.txt 2>/dev/null || exit 3
LCM_AUTOTUNE=0 LCM_LANES=1 python tools/png_hash.py > gpurun_out/fresh/p4_no_autotune_one_lane.txt 2>/dev/null || exit 4
grep -v "^\[" gpurun_out/fresh/p1.txt > gpurun_out/fresh/a.txt
for f in p2 p4_no_autotune_one_lane; do grep -v "^\[" gpurun_out/fresh/$f.txt | diff -q - gpurun_out/fresh/a.txt > /dev/null && echo "$f: identical to p1" || echo "$f: DIFFERS from p1"; done
grep -E "^batch8\[0\]|^solo\[0\]|^batch8\[5\]|^solo\[5\]" gpurun_out/fresh/a.txt
echo "--- p3 (shipped table ignored: other K partitions by design) vs p1:"; grep -v "^\[" gpurun_out/fresh/p3_no_shipped_tile_plans.txt | diff - gpurun_out/fresh/a.txt | head -4
cat gpurun_out/fresh/a.txt
