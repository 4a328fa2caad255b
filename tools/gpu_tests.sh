#!/bin/bash
# GPU test run that keeps writing (a silent run is killed after 7 minutes): pytest output goes to gpurun_out/<tag>.log, one line per test.
tag=${1:-tests}; shift
mkdir -p gpurun_out
python -m pytest "$@" -x -v --durations=15 2>&1 | tee gpurun_out/$tag.log | grep -E "PASSED|FAILED|ERROR|passed|failed|error" 
