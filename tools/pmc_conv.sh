cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "== $grp" >> gpurun_out/pmc_conv.txt
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_conv_$tag -- python3 tools/one_conv.py 8 512 128 128 > gpurun_out/pmc_conv_$tag.log 2>&1 || { echo "pass $tag failed" >> gpurun_out/pmc_conv.txt; tail -3 gpurun_out/pmc_conv_$tag.log >> gpurun_out/pmc_conv.txt; continue; }
  for c in $grp; do echo "-- $c" >> gpurun_out/pmc_conv.txt; python tools/pmc_avg.py gpurun_out/pmc_conv_$tag $c 2>&1 | head -3 >> gpurun_out/pmc_conv.txt; done
  rm -rf gpurun_out/pmc_conv_$tag
done
cat gpurun_out/pmc_conv.txt
