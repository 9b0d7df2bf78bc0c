"""GPU study for ttc_calibrate_precision: python tools/study/calibration_study.py [--budget 5e-4] > gpurun_out/precision_calibration.json

For every weight set (seeds 0..3 x {O(1)-activation scale, as-stored scale}: the trained checkpoint is absent from the reference checkout, so
"several draws of stand-in weights" is the best population there is) it
  1. assembles the model feed of bench tile 1234 (618^2, T = 12, 172-px windows, L = 4) with ONE ttc_predict_tile call and calibrates an fp16
     session on those 36 windows against the in-library fp32 engine (budget: max |dprob| on the sample),
  2. measures the calibrated session END TO END (raw uint16 tile -> pre-rounding window probabilities) against the fp32 session on the
     calibration tile AND on two held-out tiles (1235, 1236) -- the number the 1e-3 contract is about,
  3. re-measures the fixed maps earlier rounds quoted on one draw: gates on one product (one_term_layers = 1), both ConvGRU convs on two
     products (two_term_layers = 3), three products everywhere.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TILE, T, W, L = 618, 12, 172, 4


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=5e-4)
    ap.add_argument("--seeds", type=int, default=4)
    args = ap.parse_args()
    import torch
    import ttc  # noqa: F401
    from ttc import job, synth, weights as Wt

    def tile(seed):
        s2, dates, probs, _ = synth.synth_gapfill_scene(seed=seed, T=T, H=TILE, W=TILE)
        _, _, _, s1, dem = synth.synth_tile(seed=seed, T=2, H=TILE, W=TILE)
        return u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), probs, np.asarray(dates), u16(s1), dem
    tiles = {s: tile(s) for s in (1234, 1235, 1236)}

    def e2e(sess, seed, want_inputs=False):
        s2_10, s2_20, mask, dates, s1, dem = tiles[seed]
        _, _, frames, _ = sess.ctx.predict_tile_raw(s2_10, s2_20, s1, dem, mask, dates, job.min_all, job.max_all, W - 14, want_inputs=want_inputs)
        torch.cuda.synchronize()
        return frames, sess.ctx.debug_fetch("pt_windows_raw", (36, W - 14, W - 14)).copy()

    def dmax(a, b):
        ok = (a <= 1.0) & (b <= 1.0)
        return float(np.abs(a.astype(np.float64) - b)[ok].max())
    out = {"budget": args.budget, "geometry": "618x618, T=12, 36 windows of 172, L=4", "calibration_tile": 1234, "held_out_tiles": [1235, 1236],
           "reference": "the fp32 engine of the same library (<= 5e-5 of the fp64 oracle)", "weight_sets": []}
    for stored in (False, True):
        for seed in range(args.seeds):
            w = Wt.synth_weights(seed, stored_scale=stored)
            s32 = job.TTCSession(w, win_in=W, length=L, precision="fp32")
            ref = {}
            frames = None
            for ts in tiles:
                f, ref[ts] = e2e(s32, ts, want_inputs=(ts == 1234))
                frames = f if f is not None else frames
            s32.close()
            x = frames[:, :, :, 1:-1, 1:-1].permute(0, 1, 3, 4, 2).contiguous()           # [36, L+1, W, W, 17]
            row = {"weights": "synth_weights(%d, stored_scale=%s)" % (seed, stored)}
            sa = job.TTCSession(w, win_in=W, length=L, precision="auto", budget=args.budget, calibration_windows=x)
            rep = dict(sa.calibration)
            row["calibrated"] = {"report": rep, "e2e_max_dprob": {str(ts): dmax(e2e(sa, ts)[1], ref[ts]) for ts in tiles}}
            sa.close()
            for name, kw in (("all_three", {}), ("gates_one_product", {"one_term_layers": 1}), ("convgru_two_products", {"two_term_layers": 3}),
                             ("all_two_products", {"two_term_layers": 0x3FF}), ("all_one_product", {"one_term_layers": 0x3FF})):
                sx = job.TTCSession(w, win_in=W, length=L, precision="fp16", **kw)
                row[name] = {"e2e_max_dprob": {str(ts): dmax(e2e(sx, ts)[1], ref[ts]) for ts in tiles}}
                sx.close()
            out["weight_sets"].append(row)
            print("[cal] %-40s map one=%#05x two=%#05x work %.3f sample %.2e | e2e cal %s | all3 %.1e gates1 %.1e gru2 %.1e all2 %.1e all1 %.1e" % (
                row["weights"], rep["one_term_layers"], rep["two_term_layers"], rep["matrix_work_ratio"], rep["max_dprob"],
                " ".join("%.1e" % v for v in row["calibrated"]["e2e_max_dprob"].values()),
                max(row["all_three"]["e2e_max_dprob"].values()), max(row["gates_one_product"]["e2e_max_dprob"].values()),
                max(row["convgru_two_products"]["e2e_max_dprob"].values()), max(row["all_two_products"]["e2e_max_dprob"].values()),
                max(row["all_one_product"]["e2e_max_dprob"].values())), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
