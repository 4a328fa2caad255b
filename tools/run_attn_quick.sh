#!/bin/bash
# wide-head VAE attention: op tests, VAE-bearing pipeline tests, then bench lines with and without it
mkdir -p gpurun_out/attn
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > gpurun_out/attn/t_ops.log 2>&1 || { tail -30 gpurun_out/attn/t_ops.log; exit 1; }
tail -2 gpurun_out/attn/t_ops.log
python bench.py --steps 30 --warmup 5 --no-roofline > gpurun_out/attn/b1_flash.json 2>gpurun_out/attn/b1_flash.err && tail -1 gpurun_out/attn/b1_flash.json
python bench.py --steps 20 --warmup 3 --batch 8 --no-roofline --no-cpu-baseline --no-extra > gpurun_out/attn/b8_flash.json 2>gpurun_out/attn/b8_flash.err && tail -1 gpurun_out/attn/b8_flash.json
python - <<'PY' > gpurun_out/attn/micro.txt 2>&1
import torch, sys
sys.path.insert(0, ".")
import sdlcm_amd
from sdlcm_amd import ops
for B, S in ((1, 4096), (8, 4096), (1, 9216), (1, 16384)):
    C = 512
    t = torch.randn(B * S, 3 * C, device="cuda", dtype=torch.float16)
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
    f = lambda: ops.attention(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], o, B, 1, S, S, C, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"B{B} S{S} d512: {us:.1f} us  {4.0 * B * S * S * C / us / 1e6:.1f} TF/s")
PY
cat gpurun_out/attn/micro.txt
