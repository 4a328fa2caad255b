#!/bin/bash
# Round-4 PMC evidence (separate --pmc passes per counter group, program directly after `--`), on the kernels that ship:
#   gemm64_b1   igemm2<64,64,0,4,0,0>            batch-1 dominant kernel (bench.py roofline)
#   convgn_b8   conv_halo<8,16,128,1,0,0,0>      VAE 512^2 conv, GroupNorm fused, no residual
#   convgnres_b8 conv_halo<8,16,128,1,0,0,1>     the same with a residual: staged epilogue
#   attn2_b8 / attn2_b1                          streaming self-attention, key-split form, S = 4096, d = 40
#   mlp_b8      mlp_geglu_kernel<320>            fused FeedForward, 32768 rows
#   ff1_b8 / ff2_b8                              the two launches it replaces
#   o1_b8       igemm2<128,160,...>              attn.to_out + residual
# Results: gpurun_out/pmc_<tag>.txt -> profiles/r04_pmc_<tag>.txt; then tools/pmc_summary.py r04 (records the source hashes).
bash tools/pmc_kernel.sh gemm64_b1 tools/one_gemm.py 4096 320 320 > /dev/null
bash tools/pmc_kernel.sh convgn_b8 tools/one_conv.py 8 512 128 128 -1 gn > /dev/null
bash tools/pmc_kernel.sh convgnres_b8 tools/one_conv.py 8 512 128 128 -1 gn res > /dev/null
bash tools/pmc_kernel.sh attn2_b8 tools/one_attn.py 8 4096 4096 40 1 8 > /dev/null
bash tools/pmc_kernel.sh attn2_b1 tools/one_attn.py 1 4096 4096 40 1 4 > /dev/null
bash tools/pmc_kernel.sh mlp_b8 tools/one_mlp.py 32768 4096 > /dev/null
bash tools/pmc_kernel.sh ff1_b8 tools/one_gemm.py 32768 2560 320 ln geglu > /dev/null
bash tools/pmc_kernel.sh ff2_b8 tools/one_gemm.py 32768 320 1280 res > /dev/null
bash tools/pmc_kernel.sh o1_b8 tools/one_gemm.py 32768 320 320 res > /dev/null
mkdir -p profiles
for t in gemm64_b1 convgn_b8 convgnres_b8 attn2_b8 attn2_b1 mlp_b8 ff1_b8 ff2_b8 o1_b8; do cp gpurun_out/pmc_$t.txt gpurun_out/r04_pmc_$t.txt; echo "=== $t"; grep -A1 "^-- SQ_VALU_MFMA_BUSY\|^-- GRBM_GUI\|kernel stats" gpurun_out/pmc_$t.txt | grep -v "at::native\|^--$" | head -12; done
