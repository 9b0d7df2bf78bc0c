#!/bin/bash
# round 5: first contact of the Winograd F(4x4) kernel (csrc/conv3x3_wino4.hip) with the GPU: parity, then A/B against F(2x2) per layer
# usage (GPU box, repo root): bash tools/probes/r05_wino4.sh <tag> [quick]
TAG=${1:-r05_a}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
{
echo "== F(2x2) baseline (TTC_WINO4=0)"
TTC_WINO4=0 TTC_PROBE_DUMP=/tmp/p_w2.npy timeout 300 python tools/gpu_probe.py 172 4 36 fp32 2>&1 | grep -E "forward|conv_|up2|up3|out_conv|conv1|conv2|sum|max"
echo "== F(4x4) (default)"
TTC_PROBE_REF=/tmp/p_w2.npy timeout 300 python tools/gpu_probe.py 172 4 36 fp32 2>&1 | grep -E "forward|conv_|up2|up3|out_conv|conv1|conv2|sum|max|rror"
if [ "$2" != "quick" ]; then
for pb in 1 2 4 6 7; do
echo "== F(4x4) ablation TTC_WINO4_PROBE=$pb (1 no stores, 2 no transform, 4 no epilogue)"
TTC_WINO4_PROBE=$pb timeout 300 python tools/gpu_probe.py 172 4 36 fp32 2>&1 | grep -E "forward|conv_gates|conv_concat|up2 |out_conv"
done
fi
} > $O/${TAG}_wino4_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py -x -q > $O/${TAG}_wino4_tests.log 2>&1
tail -5 $O/${TAG}_wino4_tests.log
cat $O/${TAG}_wino4_ab.txt
