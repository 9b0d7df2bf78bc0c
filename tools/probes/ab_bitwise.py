"""Bitwise A/B of two library builds on the single-call tile path: model inputs (frames) and the uint8 / float rasters of the bench tile.
usage: TTC_LIB=<a.so> python tools/probes/ab_bitwise.py out_a.npz ; TTC_LIB=<b.so> python ... out_b.npz ; python tools/probes/ab_bitwise.py --cmp a b"""
import os
import sys
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        same = np.array_equal(x, y, equal_nan=True)
        d = np.abs(np.nan_to_num(x.astype(np.float64)) - np.nan_to_num(y.astype(np.float64))).max()
        print(f"{k}: identical = {same}, max|d| = {d:.3e}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa
from ttc import job, synth, weights as Wt
TILE, T = 618, 12


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234, T=T, H=TILE, W=TILE)
_, _, _, s1, dem = synth.synth_tile(seed=1234, T=2, H=TILE, W=TILE)
sess = job.TTCSession(Wt.synth_weights(0), win_in=172, length=4, max_windows=36, precision="fp32")
u8, f32, frames, status = sess.ctx.predict_tile_raw(u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), u16(s1), dem, probs, np.asarray(dates), job.min_all, job.max_all, 158,
                                                    want_float=True, want_inputs=True)
torch.cuda.synchronize()
np.savez(sys.argv[1], u8=u8.cpu().numpy(), f32=f32.cpu().numpy(), frames=frames.cpu().numpy(), status=status.cpu().numpy())
print("saved", sys.argv[1], status.cpu().numpy())
