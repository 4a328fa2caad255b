#!/bin/bash
# Round-3 judged artefacts in one box session: traffic PMC passes (eager pass, FETCH_SIZE / WRITE_SIZE separately), the default
# bench line, rocprofv3 kernel traces of batch-1 / batch-8 graph replays (stats, pass breakdown, launch sequence), worker latency.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export LCM_TUNE_CACHE=/tmp/tc.json
if [ -z "$SKIP_TRAFFIC" ]; then
  bash tools/pmc_traffic.sh > gpurun_out/final/traffic.log 2>&1 || { tail -5 gpurun_out/final/traffic.log; echo "traffic failed"; }
  cp gpurun_out/r03_traffic.json profiles/r03_traffic.json 2>/dev/null
  cp gpurun_out/r03_traffic.json gpurun_out/final/ 2>/dev/null; cp gpurun_out/r03_pmc_pass_b*.txt gpurun_out/final/ 2>/dev/null
  echo "traffic done"
fi
SKIP_BENCH= bash tools/prof_passes.sh final || exit 2
timeout -k 10 300 python3 tools/worker_latency.py > gpurun_out/final/worker_latency.txt 2>&1 || tail -5 gpurun_out/final/worker_latency.txt
tail -12 gpurun_out/final/worker_latency.txt
ls -la gpurun_out/final
