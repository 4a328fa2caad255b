#!/bin/bash
# Round artefacts in one box session: PMC traffic table -> bench (default flags) -> rocprof kernel stats / pass breakdowns -> serving.
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1 || { tail -20 gpurun_out/pmc_traffic.log; exit 1; }
tail -3 gpurun_out/pmc_traffic.log
cp gpurun_out/r02_traffic.json profiles/r02_traffic.json
bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1 || { tail -20 gpurun_out/final_profiles.log; exit 2; }
cut -c1-300 gpurun_out/final/bench.json
LCM_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-extra > gpurun_out/final/bench_2ranks_one_device.json 2> gpurun_out/final/bench_2ranks.err; cut -c1-300 gpurun_out/final/bench_2ranks_one_device.json
timeout -k 10 300 python bench.py --model sdxl --no-cpu-baseline --no-extra --no-roofline --steps 3 --warmup 1 > gpurun_out/final/bench_sdxl.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_sdxl.json
timeout -k 10 300 python bench.py --size 768 --lcm-steps 8 --batch 8 --no-cpu-baseline --no-extra --no-roofline --steps 3 --warmup 1 > gpurun_out/final/bench_768.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_768.json
timeout -k 10 400 python tools/worker_latency.py > gpurun_out/final/worker_latency.txt 2>&1; grep -E "callers|png level|sampler only" gpurun_out/final/worker_latency.txt
