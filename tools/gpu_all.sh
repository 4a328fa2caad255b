#!/bin/bash
# Whole GPU suite in ONE pytest process, output to gpurun_out/<tag>/tests.log (keeps writing: a silent run is killed after 7 minutes)
tag=${1:-all}; mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider 2>&1 | tee gpurun_out/$tag/tests.log | grep -E "passed|failed|error|FAILED|ERROR" | tail -30
grep "\[parity\]" gpurun_out/$tag/tests.log | tail -20
