cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/t8_ops.log 2>&1 || { tail -30 gpurun_out/t8_ops.log; exit 1; }
tail -2 gpurun_out/t8_ops.log
timeout -k 10 900 python tools/make_plans.py gpurun_out/tuned_plans_gfx950_r02b.json all > gpurun_out/make_plans2.log 2>&1 || { tail -20 gpurun_out/make_plans2.log; exit 2; }
tail -2 gpurun_out/make_plans2.log
cp gpurun_out/tuned_plans_gfx950_r02b.json stable-diffusion-1.5-lcm-onnx-rknn2_amd/tuned_plans_gfx950.json
python bench.py --no-cpu-baseline > gpurun_out/b5.json 2> gpurun_out/b5.err; cut -c1-250 gpurun_out/b5.json
timeout -k 10 500 python -m pytest tests/test_pipeline_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "batched_requests or two_lanes or config2 or config3_768_batch8 or multiple_of_8 or lora" > gpurun_out/t8_pipe.log 2>&1; tail -5 gpurun_out/t8_pipe.log
