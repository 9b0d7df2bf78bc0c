"""GPU-box probe: wall time of the gap-fill stage on a 618x618, T=12 tile (deterministic sampler)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttc
from ttc import job, synth, weights
sess = job.TTCSession(weights.synth_weights(0), win_in=44, length=4, dsen2_weights=None)
tiles, dates, probs, pf = synth.synth_gapfill_scene(5, 12, 618, 618)
td = torch.from_numpy(tiles).cuda(); pr = torch.from_numpy(probs).cuda()
ctx = sess.ctx
for mode in ("expected",):
    for rep in range(3):
        x = td.clone(); torch.cuda.synchronize(); t = time.time()
        interp, rem, _ = ctx.remove_cloud_and_shadows(x, pr, None, None)
        torch.cuda.synchronize(); print(mode, "total", (time.time() - t) * 1e3, "ms")
ctx.timing(1)
x = td.clone(); ctx.remove_cloud_and_shadows(x, pr, None, None)
for k in ("feather", "aligned_mosaic", "gapfill_dates"):
    print(k, ctx.kernel_ms(k))
