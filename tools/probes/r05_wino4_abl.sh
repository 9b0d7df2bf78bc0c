#!/bin/bash
# round 5: ablations of the F(4x4) kernel (TTC_WINO4_PROBE bits; the probe instantiation) and ring-depth variants (TTC_LIB)
O=gpurun_out; mkdir -p $O
run() { timeout 200 python tools/gpu_probe.py 172 4 36 fp32 2>&1 | grep -E "^forward|conv_gates|conv_concat|up2 |out_conv" | awk '{printf "%s %s | ", $1, $3}'; echo; }
{
for pb in "$@"; do
  case $pb in
    lib:*) echo -n "LIB ${pb#lib:}: "; TTC_LIB=$PWD/sentinel-tree-cover_amd/variants/${pb#lib:} run;;
    *) echo -n "PROBE $pb: "; TTC_WINO4_PROBE=$pb run;;
  esac
done
} 2>&1 | tee $O/r05_wino4_abl_$(date +%s).txt
