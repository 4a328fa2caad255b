cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/b4.json 2> gpurun_out/b4.err; cut -c1-250 gpurun_out/b4.json
LCM_LN_FOLD=0 python bench.py --no-cpu-baseline --no-extra --no-roofline > gpurun_out/b4_nofold.json 2>/dev/null; cut -c1-200 gpurun_out/b4_nofold.json
python tools/worker_latency.py > gpurun_out/wl2.log 2>&1; grep -E "callers|png level" gpurun_out/wl2.log
LCM_LANES=1 python tools/worker_latency.py > gpurun_out/wl1.log 2>&1; grep -E "callers" gpurun_out/wl1.log
timeout -k 10 300 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "lora" > gpurun_out/t7_lora.log 2>&1; tail -3 gpurun_out/t7_lora.log
