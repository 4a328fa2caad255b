#!/bin/bash
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc runs over one eager pass (batch 1 and batch 8) -> gpurun_out/${TRAFFIC_OUT:-r04_traffic.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
specs=""
for B in 1 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_${B}_$c
    rm -rf $d
    timeout -k 10 280 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/pmc_eager_pass.py $B > gpurun_out/pmc_${B}_$c.log 2>&1
    rc=$?; tail -1 gpurun_out/pmc_${B}_$c.log
    if [ $rc -ne 0 ]; then echo "pass B=$B $c failed rc=$rc"; exit $rc; fi
    specs="$specs $B:$c:$d"
  done
done
python tools/pmc_traffic.py gpurun_out/${TRAFFIC_OUT:-r04_traffic.json} $specs
for B in 1 8; do for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_avg.py gpurun_out/pmc_${B}_$c $c > gpurun_out/r04_pmc_pass_b${B}_$c.txt; rm -rf gpurun_out/pmc_${B}_$c; done; done
ls -la gpurun_out/${TRAFFIC_OUT:-r04_traffic.json}
