cd $GRAFT_REPO_ROOT
python tools/diag_batch.py 768 8 1 > gpurun_out/diag768.log 2>&1; grep -v "^==" gpurun_out/diag768.log | head -30
python -m pytest tests/test_ops_gpu.py -m gpu -q > gpurun_out/t5_ops.log 2>&1; tail -5 gpurun_out/t5_ops.log
python -m pytest tests/test_pipeline_gpu.py tests/test_pool_lifecycle_gpu.py tests/test_worker_gpu.py -m gpu -q > gpurun_out/t5_pipe.log 2>&1; tail -8 gpurun_out/t5_pipe.log
python bench.py --no-cpu-baseline > gpurun_out/b3.json 2> gpurun_out/b3.err; cut -c1-250 gpurun_out/b3.json
LCM_LN_FOLD=0 python bench.py --no-cpu-baseline --no-extra --no-roofline > gpurun_out/b3_nofold.json 2>/dev/null; cut -c1-200 gpurun_out/b3_nofold.json
python tools/worker_latency.py > gpurun_out/wl_lanes2.log 2>&1; tail -8 gpurun_out/wl_lanes2.log
