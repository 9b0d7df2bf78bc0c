"""GPU-box probe: per-kernel device time of one batched forward (HIP events inside the library)."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttc
from ttc import _lib, synth, weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 172
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 36
PREC = sys.argv[4] if len(sys.argv) > 4 else "fp32"        # fp32 | fp16 | bf16
MASK = int(sys.argv[5], 0) if len(sys.argv) > 5 else None     # one_term_layers of the 16-bit engine
ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision=PREC, one_term_layers=MASK)
ctx.load_weights(weights.synth_weights(0))
x = torch.from_numpy(synth.synth_windows(seed=1, N=N, L=L, W=W)).cuda()
print("device bytes", ctx.device_bytes / 1e9, "GB")
for _ in range(2):
    out0 = ctx.forward_windows(x)
torch.cuda.synchronize()
if os.environ.get("TTC_PROBE_DUMP"):                           # A/B of two builds / switches: dump the probabilities
    np.save(os.environ["TTC_PROBE_DUMP"], out0.cpu().numpy())
if os.environ.get("TTC_PROBE_REF"):                            # ... and compare with an earlier dump
    ref = np.load(os.environ["TTC_PROBE_REF"])
    d = np.abs(out0.cpu().numpy().astype(np.float64) - ref)
    print(f"max|dprob| vs {os.environ['TTC_PROBE_REF']}: {d.max():.3e} (mean {d.mean():.3e}, nan {int(np.isnan(d).sum())})")
t = time.time()
K = 5
for _ in range(K):
    ctx.forward_windows(x)
torch.cuda.synchronize()
dt = (time.time() - t) / K
from oracle.restate_model import model_flops
fl = model_flops(W, L) * N
print(f"forward {N} windows W={W} L={L}: {dt*1e3:.2f} ms  -> {fl/dt/1e12:.1f} TFLOP/s (algorithmic), {N/36*618*618/dt/1e6:.1f} Mpx/s")
ctx.timing(True)
for _ in range(3):
    ctx.forward_windows(x)
tot = 0
for k in ["frames_from_nhwc", "frames_to_b16", "conv_gates", "conv_cand", "gn_finalize", "gru_apply1", "gru_apply2", "conv_median",
          "conv_concat", "conv1", "conv2", "conv_up2", "conv_up2_out", "conv_up3", "out_conv", "block_finalize", "head"]:
    ms, n = ctx.kernel_ms(k)
    per_fwd = ms * n / 3
    tot += per_fwd
    print(f"  {k:18s} avg {ms:8.3f} ms x {n/3:5.1f}/fwd = {per_fwd:8.3f} ms")
print("  sum", tot)
