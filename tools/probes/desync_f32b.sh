for d in ${DS:-0 1 2 3 5 8}; do echo "== desync $d"; TTC_F32B_DESYNC=$d python tools/gpu_probe.py 172 4 36 fp32b 2>&1 | grep -E "conv_gates|conv_cand|conv_concat|up3"; done
