cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/t6_ops.log 2>&1 || { tail -30 gpurun_out/t6_ops.log; exit 1; }
tail -2 gpurun_out/t6_ops.log
timeout -k 10 900 python tools/make_plans.py gpurun_out/tuned_plans_gfx950_r02.json all > gpurun_out/make_plans.log 2>&1 || { tail -20 gpurun_out/make_plans.log; exit 2; }
tail -3 gpurun_out/make_plans.log
timeout -k 10 400 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "multiple_of_8 or two_lanes or batched_requests or lora" > gpurun_out/t6_pipe.log 2>&1; tail -6 gpurun_out/t6_pipe.log
