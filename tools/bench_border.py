"""End-to-end timing of one tile border (src/resegment_tiles_wide.py:847-1161 + the border mosaic of both tiles) at
production size on one MI355X: two 618 x 618 tiles with T dates as process_tile returns them (resident in HBM) ->
four re-predicted 220 x 684 windows -> both tiles' rasters re-mosaicked.  Prints one JSON line.

    python tools/bench_border.py [--precision fp32|fp16|bf16] [--dates 12] [--iters 5] [--separate]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RSG = importlib.import_module("sentinel-tree-cover_amd.resegment")
Wt = importlib.import_module("sentinel-tree-cover_amd.weights")
from tests.helpers import synth_border_pair, synth_reseg_windows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--dates", type=int, default=12)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--separate", action="store_true", help="neighbour dates differ: per-tile preprocessing + histogram alignment")
    a = ap.parse_args()
    sess = RSG.border_session(Wt.synth_weights(0), precision=a.precision)
    ctx = sess.ctx
    tile, neighb, tif_t, tif_n = synth_border_pair(7, a.dates, 618, 618, same_dates=not a.separate)
    dev = lambda d: {k: (ctx._dev(v, torch.float32) if k != "dates" else v) for k, v in d.items()}       # noqa: E731
    tile, neighb = dev(tile), dev(neighb)
    tt, tn = tif_t.astype(np.float32), tif_n.astype(np.float32)
    tt[tt > 100] = np.nan
    tn[tn > 100] = np.nan
    fmt = {"n": "{x}/{y}.npy", "l": "{x}/left{y}.npy", "r": "right{x}/{y}.npy"}
    plain = {fmt[k].format(x=x, y=y): p for k, x, y, p in synth_reseg_windows(63, (618, 618), 670, 206, False) if k == "n"}

    def once():
        wins, info = RSG.resegment_border(tile, neighb, tt, tn, sess, sampler="expected")
        left = dict(plain); right = dict(plain)
        left.update({k: v for k, v in wins.items() if k.startswith("right")})
        right.update({"0/" + k: v for k, v in wins.items() if k.startswith("left")})
        a1 = RSG.recreate_resegmented_tifs(left, (618, 618), sess, return_sums=False)
        a2 = RSG.recreate_resegmented_tifs(right, (618, 618), sess, return_sums=False)
        return info, a1, a2
    for _ in range(2):
        info, a1, a2 = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    ctx.timing(True)
    once()
    torch.cuda.synchronize()
    names = ["strip_smooth", "dsen2_conv", "dsen2_gather", "dsen2_scatter", "border_medians", "border_hist_align", "border_assemble",
             "conv_gates", "conv_cand", "gru_apply1", "gru_apply2", "gn_finalize", "block_finalize", "conv_median", "conv_concat",
             "conv1", "conv2", "up2", "up2_out", "up3", "out_conv", "head", "border_seam_adjust", "reseg_mosaic"]
    km = {}
    for n in names:
        try:
            avg, cnt = ctx.kernel_ms(n)
            km[n] = round(avg * cnt, 3)
        except RuntimeError:
            pass
    print(json.dumps({"metric": "tile borders/s (resegment_border + both border mosaics)", "value": 1e3 / ms, "ms_per_border": ms,
                      "precision": a.precision, "dates": a.dates, "branch": "per-tile + hist-align" if a.separate else "shared strip",
                      "min_images": info["min_images"], "hist_align": info["hist_align"], "nodata_frac": [float((a1 == 255).mean()), float((a2 == 255).mean())],
                      "kernel_ms_timed_families": km, "note": "inputs resident in HBM; detection + gap-fill families are not in the timed-family list"}))


if __name__ == "__main__":
    main()
