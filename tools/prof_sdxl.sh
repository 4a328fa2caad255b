#!/bin/bash
# rocprofv3 kernel trace of the SDXL 1024x1024 30-step graph replay -> per-kernel stats + pass breakdown under gpurun_out/$1
set -o pipefail
OUT=gpurun_out/${1:-prof_sdxl}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x -- python3 bench.py --model sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-roofline > $OUT/bench.json 2> $OUT/x.err || { tail -20 $OUT/x.err; exit 2; }
f=$(find $OUT/x -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 tools/pass_breakdown.py $OUT/x 70 > $OUT/pass_breakdown.txt
rm -rf $OUT/x
cut -c1-200 $OUT/bench.json; head -50 $OUT/pass_breakdown.txt
