#!/bin/bash
# batch-1 pass time against the single-launch GroupNorm threshold (bit-neutral launch choice)
for v in 8 16 32 64; do
  LCM_GN_FUSED_BYTES=$((v<<20)) python bench.py --steps 16 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null > gpurun_out/gn_$v.json
  python -c "import json; d=json.loads(open('gpurun_out/gn_$v.json').read()); print($v, d['value'], d['ms_per_step'])"
done
