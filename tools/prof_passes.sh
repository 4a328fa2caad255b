#!/bin/bash
# One box session: bench JSON (default flags), then rocprofv3 kernel traces of batch 1 and batch 8 graph replays ->
# per-kernel stats, pass breakdown and launch-by-launch sequence under gpurun_out/$1 (default "prof").
set -o pipefail
OUT=gpurun_out/${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LCM_TUNE_CACHE=/tmp/tc.json
mkdir -p $OUT
if [ -z "$SKIP_BENCH" ]; then
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
echo "bench done" && cut -c1-300 $OUT/bench.json
fi
for t in b1 b8; do
  bs=1; [ $t = b8 ] && bs=8
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 bench.py --batch $bs --no-cpu-baseline --no-extra --no-roofline > $OUT/${t}_bench.json 2> $OUT/$t.err || { tail -20 $OUT/$t.err; exit 2; }
  echo "$t prof done"
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${t}_kernel_stats.csv
  python3 tools/pass_breakdown.py $OUT/$t 60 > $OUT/${t}_pass_breakdown.txt
  python3 tools/pass_sequence.py $OUT/$t > $OUT/${t}_pass_sequence.txt
  rm -rf $OUT/$t
done
ls -la $OUT
