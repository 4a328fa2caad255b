for a in 0 1 2 3 4; do echo "ABLATE $a"; LCM_MLP_ABLATE=$a LCM_MLP_FUSED_MIN_ROWS=1 python tools/mlp_fused_ab.py 2>&1 | grep -E "M  32768.*one|M  16384.*one"; done
