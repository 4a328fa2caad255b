#!/bin/bash
# Round-3 PMC evidence for the kernels that dominate the passes now (separate --pmc passes per counter group, program directly
# after `--`): igemm2<64,64,0,4,0,0> (batch 1), conv_halo<8,16,128,1,0> (batch 8 VAE), attn2<40,8> (batch 8) and attn2<40,4>
# (batch 1).  Results: gpurun_out/pmc_<tag>.txt -> copied to profiles/r03_pmc_<tag>.txt.
bash tools/pmc_kernel.sh gemm64_b1 tools/one_gemm.py 4096 320 320 > /dev/null
bash tools/pmc_kernel.sh convgn_b8 tools/one_conv.py 8 512 128 128 -1 gn > /dev/null
bash tools/pmc_kernel.sh attn2_b8 tools/one_attn.py 8 4096 4096 40 1 8 > /dev/null
bash tools/pmc_kernel.sh attn2_b1 tools/one_attn.py 1 4096 4096 40 1 4 > /dev/null
for t in gemm64_b1 convgn_b8 attn2_b8 attn2_b1; do echo "=== $t"; grep -A1 "^-- \|kernel stats" gpurun_out/pmc_$t.txt | grep -v "at::native\|^--$" | head -60; done
