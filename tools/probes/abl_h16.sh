for a in ${ABLS:-0 1 2 4 5 3 6 7}; do echo "== abl $a"; TTC_H16_ABL=$a python tools/gpu_probe.py 172 4 36 ${PREC:-fp16} 2>&1 | grep -E "conv_gates|conv_cand|conv_concat|up3"; done
