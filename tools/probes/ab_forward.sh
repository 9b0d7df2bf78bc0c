#!/bin/bash
# A/B of library builds on ONE box: tools/probes/ab_forward.sh <rounds> <lib> [<lib> ...]   (TTC_LIB selects the build; each round runs
# every build once, so clock / box differences hit all of them alike).  Prints the forward time and the Winograd layers per run.
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    out=$(TTC_LIB=$lib timeout 200 python tools/gpu_probe.py 172 4 36 ${AB_PREC:-fp32} 2>&1)
    fwd=$(echo "$out" | grep -E "^forward" | sed 's/.*: \([0-9.]*\) ms.*/\1/')
    lay=$(echo "$out" | grep -E "conv_gates|conv_cand|conv_median|conv_concat|conv1|conv2|up2 |up3|out_conv" | awk '{printf "%s %s  ", $1, $3}')
    echo "$(basename $lib) round $r: forward $fwd ms | $lay"
  done
done
