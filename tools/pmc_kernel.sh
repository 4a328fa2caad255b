#!/bin/bash
# PMC passes over one single-kernel target:  pmc_kernel.sh <tag> <python script + args...>
# Counters in separate passes (SQ: 8 slots; FETCH_SIZE / WRITE_SIZE cannot share a pass), program directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmc_$tag.txt
: > $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | cut -d' ' -f1)
  echo "== $grp" >> $out
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_$g -- python3 "$@" > gpurun_out/pmc_${tag}_$g.log 2>&1 || { echo "pass $g failed" >> $out; tail -3 gpurun_out/pmc_${tag}_$g.log >> $out; continue; }
  for c in $grp; do echo "-- $c" >> $out; python3 tools/pmc_avg.py gpurun_out/pmc_${tag}_$g $c 2>&1 | head -4 >> $out; done
  rm -rf gpurun_out/pmc_${tag}_$g gpurun_out/pmc_${tag}_$g.log
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_${tag}_kt -- python3 "$@" > /dev/null 2>&1
f=$(find gpurun_out/pmc_${tag}_kt -name "*kernel_stats.csv" | head -1)
echo "== kernel stats" >> $out; head -6 "$f" >> $out
rm -rf gpurun_out/pmc_${tag}_kt
cat $out
