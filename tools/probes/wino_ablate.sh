# ablation of the Winograd fp32 kernel (TTC_WINO_PROBE bits: see conv3x3_wino.hip WinoArgs.probe); timing only.
# A non-zero probe value runs the 64-cout layers on the probe instantiation (conv3x3_wino<2, EPI, 1>: the switches are branches in its chunk
# loop, so its own baseline -- probe 0 is the PRODUCT kernel -- is a few percent slower); the 32-cout kernels ignore the bits.
cd /root/repo
for p in ${WINO_PROBES:-0 30}; do
  echo "== probe $p"; TTC_WINO_PROBE=$p timeout 120 python tools/gpu_probe.py 172 4 36 fp32 2>&1 | grep -E "max|forward|conv_gates|conv_cand|up2 |conv_concat|conv_median"
done
