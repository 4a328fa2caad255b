#!/bin/bash
# Regenerates the round's judged artefacts in ONE box session: bench JSON (default flags), then the same command under
# rocprofv3 --kernel-trace --stats for batch 1 and batch 8 (tuning plans cached so all three runs use the same plans).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LCM_TUNE_CACHE=/tmp/tc.json
mkdir -p gpurun_out/final
timeout -k 10 700 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || exit 1
echo "bench done" && cat gpurun_out/final/bench.json | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/b1 -- python3 bench.py --no-cpu-baseline --no-extra --no-roofline > gpurun_out/final/b1_bench.json 2> gpurun_out/final/b1.err || exit 2
echo "b1 prof done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/b8 -- python3 bench.py --batch 8 --no-cpu-baseline --no-extra --no-roofline > gpurun_out/final/b8_bench.json 2> gpurun_out/final/b8.err || exit 3
echo "b8 prof done"
for t in b1 b8; do
  f=$(find gpurun_out/final/$t -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/final/${t}_kernel_stats.csv
  python tools/pass_breakdown.py gpurun_out/final/$t 60 > gpurun_out/final/${t}_pass_breakdown.txt
  python tools/pass_sequence.py gpurun_out/final/$t | gzip -c > gpurun_out/final/${t}_pass_sequence.txt.gz
  rm -rf gpurun_out/final/$t
done
ls -la gpurun_out/final
