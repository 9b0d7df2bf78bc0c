"""GPU probe: max |dprob| of a whole 618^2 tile (pre-rounding, vs the fp32 oracle model fed by the oracle preprocessing) for a
given precision / one_term_layers mask.   python tools/probes/tile_dprob_probe.py fp16 0xFC00 [0x400 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ttc
from ttc import job, weights as Wt
from oracle import restate_model as M, restate_numpy as O
from tests.helpers import golden, e2e_inputs

prec = sys.argv[1]
masks = [int(a, 0) for a in sys.argv[2:]] or [0]
size, length = 158, 4
w = Wt.synth_weights(0, stored_scale=True)
g = golden("e2e_cloudy.npz")
s2, dates, interp, s1, dem = e2e_inputs(g)
# the oracle side runs DSen2 too, so that the super-resolution's precision is part of what is measured
ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
s2_ref = O.superresolve_large_tile(s2.copy(), ds)
net = M.TreeCoverNet(w, dtype=torch.float32)
raw_ref = []
def model(win):
    p = O.predict_subtile(win, net, size); raw_ref.append(np.array(p, copy=True)); return p
ref_w, feeds = O.process_subtiles(s2_ref.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), model, size=size, length=length, return_inputs=True)
for mask in masks:
    sess = job.TTCSession(w, win_in=size + 14, length=length, max_windows=36, precision=prec, one_term_layers=mask)
    d = torch.from_numpy(s2.copy()).cuda()
    sess.ctx.superresolve_tile(d, quirks=True)
    sr = d.cpu().numpy()
    wins, raw = job.process_subtiles(0, 0, d, dates, interp, s1, dem, sess, size=size, return_raw=True)
    worst = max(float(np.abs(raw[k].astype(np.float64) - r).max()) for k, r in zip(feeds.keys(), raw_ref))
    allv = np.concatenate([np.abs(raw[k].astype(np.float64) - r).ravel() for k, r in zip(feeds.keys(), raw_ref)])
    print(f"{prec} one_term_layers={mask:#x}: superresolved reflectance max|d| {np.abs(sr - s2_ref).max():.2e}; tile max|dprob| {worst:.3e} p99.9 {np.quantile(allv, 0.999):.2e} rms {np.sqrt((allv**2).mean()):.2e}", flush=True)
    sess.close()
