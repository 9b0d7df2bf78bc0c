"""GPU probe: sweep the 16-bit conv engine's knobs (persistent grid size, start offset of the odd wave slot) in ONE process and
print the per-layer conv times of a 36-window forward + the DSen2 convs of one tile.  python tools/probes/h16_knobs.py [fp16|bf16]"""
import os
os.environ["TTC_ENABLE_PROBE_KNOBS"] = "1"     # ttc_debug_knob is refused otherwise (process-wide probe state)
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa: F401
from ttc import _lib, synth, weights

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16"
W, L, N = 172, 4, 36
ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision=PREC)
ctx.load_weights(weights.synth_weights(0))
ctx.load_dsen2_weights(weights.load_dsen2())
x = torch.from_numpy(synth.synth_windows(seed=1, N=N, L=L, W=W)).cuda()
s2 = torch.rand((12, 618, 618, 10), device="cuda") * 0.5
lib = _lib.load()
LAYERS = ["conv_gates", "conv_cand", "conv_median", "conv_concat", "conv1", "conv2", "up2", "up2_out", "up3", "out_conv"]


def run(grid, desync):
    lib.ttc_debug_knob(0, grid)
    lib.ttc_debug_knob(1, desync)
    for _ in range(2):
        ctx.forward_windows(x)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        ctx.forward_windows(x)
    torch.cuda.synchronize()
    fwd = (time.time() - t) / 5 * 1e3
    ctx.timing(1); ctx.kernel_ms(None)
    for _ in range(3):
        ctx.forward_windows(x)
        ctx.superresolve_tile(s2.clone(), quirks=True)
    torch.cuda.synchronize()
    row = {k: ctx.kernel_ms(k)[0] for k in LAYERS}
    ds, n = ctx.kernel_ms("dsen2_conv")
    ctx.timing(0)
    conv = row["conv_gates"] * 4 + row["conv_cand"] * 4 + sum(row[k] for k in LAYERS[2:])
    print(f"grid {grid:4d} desync {desync:2d}: fwd {fwd:6.2f} ms | model conv {conv:5.2f} | " + " ".join(f"{k.replace('conv_', '')} {row[k]:.3f}" for k in LAYERS)
          + f" | dsen2 conv {ds:.3f} x {n // 3} = {ds * n / 3:.2f} ms/tile", flush=True)


for grid in (-1, 0):
    for d in (0, 1, 2, 3, 4, 6, 8, 12):
        run(grid, d)
for grid in (384, 448, 640):
    run(grid, 3)
