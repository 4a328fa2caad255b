#!/bin/bash
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc runs over one eager pass; per-kernel averages to gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-1}
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $c (batch $B)" | tee -a gpurun_out/pmc_pass_b$B.txt
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_pass_$c -- python3 tools/pmc_eager_pass.py $B > gpurun_out/pmc_pass_$c.log 2>&1
  rc=$?; tail -2 gpurun_out/pmc_pass_$c.log
  if [ $rc -ne 0 ]; then echo "pass $c failed rc=$rc" | tee -a gpurun_out/pmc_pass_b$B.txt; exit $rc; fi
  python tools/pmc_avg.py gpurun_out/pmc_pass_$c $c >> gpurun_out/pmc_pass_b$B.txt 2>&1
  rm -rf gpurun_out/pmc_pass_$c
done
cat gpurun_out/pmc_pass_b$B.txt
