"""GPU probe: shader clock / power as rocm-smi reports them while a forward loop of the given precision runs.
python tools/probes/power_clock.py [fp32|fp16|bf16]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa
from ttc import _lib, synth, weights

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16"
ctx = _lib.Context(win_in=172, length=4, max_windows=36, precision=PREC)
ctx.load_weights(weights.synth_weights(0))
x = torch.from_numpy(synth.synth_windows(seed=1, N=36, L=4, W=172)).cuda()
stop = False


def loop():
    while not stop:
        for _ in range(20):
            ctx.forward_windows(x)
        torch.cuda.synchronize()


print(subprocess.run("rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E 'Max Graphics|Average Graphics|Current Socket|sclk|mclk' | head -8", shell=True, capture_output=True, text=True).stdout)
th = threading.Thread(target=loop); th.start()
time.sleep(1.5)
for i in range(4):
    out = subprocess.run("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Average Graphics|Current Socket|sclk' | head -4", shell=True, capture_output=True, text=True).stdout
    print(f"[{PREC} under load, sample {i}]\n{out}")
    time.sleep(0.7)
stop = True; th.join()
