#!/bin/bash
# 256-pixel conv tile: bit-neutrality test, then batch-8 / batch-1 bench lines with the switch off / on, and the GN-fusion threshold
mkdir -p gpurun_out/bm256
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "256_pixel or gn_fused_conv_is_bit" > gpurun_out/bm256/t_ops.log 2>&1 || { tail -30 gpurun_out/bm256/t_ops.log; exit 1; }
tail -2 gpurun_out/bm256/t_ops.log
B8="--steps 12 --warmup 3 --batch 8 --no-roofline --no-cpu-baseline --no-extra"
python bench.py $B8 > gpurun_out/bm256/b8_off.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b8_off.json
LCM_HALO_BM256=1 python bench.py $B8 > gpurun_out/bm256/b8_on1.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b8_on1.json
LCM_HALO_BM256=2 LCM_HALO_BM256_MIN_TILES=512 python bench.py $B8 > gpurun_out/bm256/b8_on2.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b8_on2.json
LCM_HALO_BM256=1 LCM_FUSE_GN_MIN_BYTES=1099511627776 python bench.py $B8 > gpurun_out/bm256/b8_on1_nofuse.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b8_on1_nofuse.json
LCM_FUSE_GN_MIN_BYTES=1099511627776 python bench.py $B8 > gpurun_out/bm256/b8_off_nofuse.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b8_off_nofuse.json
B1="--steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra"
LCM_HALO_BM256=1 python bench.py $B1 > gpurun_out/bm256/b1_on1.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b1_on1.json
LCM_FUSE_GN_MIN_BYTES=0 python bench.py $B1 > gpurun_out/bm256/b1_fuseall.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b1_fuseall.json
LCM_FUSE_GN_MIN_BYTES=8388608 python bench.py $B1 > gpurun_out/bm256/b1_fuse8m.json 2>/dev/null && grep -o '"value": [0-9.]*' gpurun_out/bm256/b1_fuse8m.json
