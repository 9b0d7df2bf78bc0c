"""GPU-box probe: run the fp16 forward in a loop for a few seconds and print the average gates-conv launch time.  Start two of
these at once with different TTC_H16_ABL values (5 = MFMA only, 2 = copies + epilogue only) to see whether the phases of the
16-bit conv kernel can overlap when they come from DIFFERENT kernels sharing the CUs."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttc import _lib, synth, weights
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
start_at = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = _lib.Context(win_in=172, length=4, max_windows=36, precision="fp16")
ctx.load_weights(weights.synth_weights(0))
x = torch.from_numpy(synth.synth_windows(seed=1, N=36, L=4, W=172)).cuda()
for _ in range(3): ctx.forward_windows(x)
torch.cuda.synchronize()
while time.time() < start_at: time.sleep(0.001)
ctx.timing(True)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    ctx.forward_windows(x); n += 1
    if n % 8 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.time() - t0
ms, k = ctx.kernel_ms("conv_gates")
print(f"abl {os.environ.get('TTC_H16_ABL', '0')}: {n} forwards in {dt:.2f} s = {dt / n * 1e3:.2f} ms each; conv_gates avg {ms:.3f} ms over {k} launches")
