"""Golden vectors for the tile-border resegmentation (src/resegment_tiles_wide.py), captured by running the REFERENCE's
own functions in this container (see tools/ref_harness.py for the import shims).  Run from the repo root:

    python tools/gen_golden_reseg.py

Writes tests/golden/reseg_*.npz.  Inputs are regenerated in the tests from tests/helpers.py (synth_border_strip /
synth_reseg_windows), so only outputs (sub-sampled where large) are stored."""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import ref_harness  # noqa: E402
from tests.helpers import fake_model, fake_dsen2, synth_border_strip, synth_reseg_windows  # noqa: E402

OUT = os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))


def set_geometry(RS, size, size_y):
    RS.SIZE, RS.SIZE_Y = size, size_y
    mn, mx = np.array(MIN_ALL), np.array(MAX_ALL)
    RS.min_all = np.broadcast_to(mn, (1, 1, 1, 17)).astype(np.float32)
    RS.max_all = np.broadcast_to(mx, (1, 1, 1, 17)).astype(np.float32)
    RS.midrange = ((RS.max_all + RS.min_all) / 2).astype(np.float32)
    RS.rng = (RS.max_all - RS.min_all).astype(np.float32)


# resegment_tiles_wide.py:1664-1678 (data constants of the reference's normalisation; they live under `__main__` there)
from oracle.restate_reseg import MIN_ALL32 as MIN_ALL, MAX_ALL32 as MAX_ALL  # noqa: E402


def write_folder(folder, wins):
    """lay a window list out the way process_subtiles (job.py:1362, reseg :432-433) saves it"""
    for kind, xt, yt, p in wins:
        d, f = {"n": (f"{xt}", f"{yt}.npy"), "l": (f"{xt}", f"left{yt}.npy"), "r": (f"right{xt}", f"{yt}.npy"),
                "u": (f"{xt}", f"up{yt}.npy"), "d": (f"{xt}", f"down{yt}.npy")}[kind]
        os.makedirs(os.path.join(folder, d), exist_ok=True)
        np.save(os.path.join(folder, d, f), p)


def listing_order(folder, RS):
    """the order in which recreate_resegmented_tifs walks the folder (os.listdir order), as (kind, x, y) keys"""
    order = []
    x_tiles = [x for x in os.listdir(folder) if 'right' not in x]
    x_tiles = [int(x) for x in x_tiles if len(os.listdir(folder + "/" + x)) > 0]
    for xt in x_tiles:
        ys = [y for y in os.listdir(folder + str(xt) + "/")]
        order += [("n", xt, int(y[:-4])) for y in ys if not any(k in y for k in ("left", "down", "up"))]
    for xt in x_tiles:
        order += [("l", xt, int(y[4:-4])) for y in os.listdir(folder + str(xt) + "/") if 'left' in y]
    for name in [x for x in os.listdir(folder) if 'right' in x]:
        order += [("r", int(name[5:]), int(y[:-4])) for y in os.listdir(folder + name + "/")]
    for xt in x_tiles:
        order += [("u", xt, int(y[2:-4])) for y in os.listdir(folder + str(xt) + "/") if 'up' in y]
    for xt in x_tiles:
        order += [("d", xt, int(y[4:-4])) for y in os.listdir(folder + str(xt) + "/") if 'down' in y]
    return order


def main():
    J, CR = ref_harness.load()
    import resegment_tiles_wide as RS
    scratch = tempfile.mkdtemp(prefix="ttc_reseg_") + "/"
    os.chdir(scratch)      # the reference's cloud_removal np.save()s debug arrays into the CWD (cloud_removal.py:927-929, :972)
    RS.args = types.SimpleNamespace(local_path=scratch, year=2020)
    RS.x, RS.y = "10", "20"
    RS.predict_logits, RS.predict_inp, RS.predict_length = "logits", "inp", "len"
    g = {}

    # ---- align_dates / window table / artifact test ---------------------------------------------------------------
    cases = [([5, 40, 100, 160, 220], [5, 41, 100, 163, 220, 300]), ([10, 10, 50, 90], [10, 50, 91, 91]),
             ([1, 2, 3], [100, 200]), ([15, 45, 75, 105, 135, 165], [15, 45, 75, 105, 135, 165])]
    for i, (a, b) in enumerate(cases):
        ra, rb, left = RS.align_dates(a, b)
        g[f"dates{i}_a"], g[f"dates{i}_b"] = np.array(a), np.array(b)
        g[f"dates{i}_rm_a"], g[f"dates{i}_rm_b"], g[f"dates{i}_left"] = np.array(ra, dtype=np.int64), np.array(rb, dtype=np.int64), np.int64(left)
    for tag, (n_rows, size, size_y) in {"real": (618, 670, 206), "small": (150, 90, 46), "odd": (611, 670, 206)}.items():
        set_geometry(RS, size, size_y)
        gap_y = int(np.ceil((n_rows - size_y) / 3))
        fy = np.hstack([np.arange(0, n_rows - size_y, gap_y), np.array(n_rows - size_y)])
        # tiles_folder_x = split_to_border's 5th value = the tile's column offset of the border strip (:895, :1138)
        ta, tf = RS.make_tiles_right_neighb(n_rows - size // 2, fy)
        g[f"table_{tag}_cfg"] = np.array([n_rows, size, size_y])
        g[f"table_{tag}_array"], g[f"table_{tag}_folder"] = np.asarray(ta), np.asarray(tf)
        print("table", tag, np.asarray(ta).tolist(), np.asarray(tf).tolist(), np.asarray(ta).dtype)
    rng = np.random.default_rng(5)
    arts = []
    for i in range(12):
        base = np.clip(50 + 30 * np.sin(np.arange(618) / (20 + 3 * i))[:, None] + rng.normal(0, 4, (618, 20)), 0, 100).astype(np.float32)
        nb = base[:, ::-1] + np.float32([0, 0.5, 2, 5, 7, 14, 25][i % 7]) * (1 if i < 7 else np.sign(np.sin(np.arange(618) / 15.0))[:, None])
        nb = np.clip(nb, 0, 100).astype(np.float32)
        if i % 3 == 0:
            base[100:140, -4:] = np.nan
            nb[300:320, :2] = np.nan
        arts.append(RS.check_if_artifact(base.copy(), nb.copy()))
    g["artifact_flags"] = np.array(arts)
    print("artifact", arts)

    # ---- align_subtile_histograms ------------------------------------------------------------------------------------
    set_geometry(RS, 90, 46)
    for tag, (seed, off) in {"h0": (41, 0.06), "h1": (42, 0.0), "h2": (43, -0.1)}.items():
        s2 = synth_border_strip(seed, 60, 104, offset=off)[0]
        arr = np.median(np.reshape(np.nan_to_num(s2), (4, 3) + s2.shape[1:]), axis=1)
        ref = RS.align_subtile_histograms(arr.copy())
        g[f"{tag}_cfg"] = np.array([seed, 60, 104]); g[f"{tag}_off"] = np.float32(off)
        g[f"{tag}_out"] = ref[:, ::3, ::4, :].astype(np.float32)
        g[f"{tag}_changed"] = np.array([not np.array_equal(ref[t], arr[t]) for t in range(4)])
        print("hist", tag, g[f"{tag}_changed"])
    np.savez_compressed(os.path.join(OUT, "reseg_small.npz"), **g)

    # ---- process_subtiles (border re-prediction) with a fake session ---------------------------------------------------
    class Sess:
        def __init__(self):
            self.feeds = []

        def run(self, op, feed_dict):
            self.feeds.append(np.array(feed_dict["inp"], copy=True))
            return fake_model(feed_dict["inp"])
    st = {}
    for tag, (seed, X, size, size_y, align, off) in {"a": (51, 330, 90, 134, True, 0.06), "b": (52, 330, 90, 134, False, 0.0),
                                                    "c": (53, 300, 66, 134, True, -0.05)}.items():
        set_geometry(RS, size, size_y)
        s2, dates, interp, s1, dem, left_all, right_all, min_clear = synth_border_strip(seed, X, size + 14, offset=off)
        if tag == "c":
            s1[...] = 0; s2[...] = 0; dem[...] = 0          # all-zero windows -> 255 fill
            s2[:, 150:, :, :] = synth_border_strip(seed, X, size + 14, offset=off)[0][:, 150:]
        gap_y = int(np.ceil((X - size_y) / 3))
        fy = np.hstack([np.arange(0, X - size_y, gap_y), np.array(X - size_y)])
        ta, tf = RS.make_tiles_right_neighb(X + 9 - size // 2, fy)
        sess = Sess()
        shutil.rmtree(scratch + "10", ignore_errors=True); shutil.rmtree(scratch + "11", ignore_errors=True)
        RS.process_subtiles(10, 20, s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), sess, None, tf, ta,
                            right_all.copy(), left_all.copy(), align, min_clear.copy())
        st[f"{tag}_cfg"] = np.array([seed, X, size, size_y, int(align)]); st[f"{tag}_off"] = np.float32(off)
        st[f"{tag}_n_feeds"] = np.int64(len(sess.feeds))
        for i, f in enumerate(sess.feeds):
            st[f"{tag}_feed{i}"] = f[0, :, ::5, ::7, :].astype(np.float32)
            st[f"{tag}_feed{i}_sum"] = np.float64(f.astype(np.float64).sum())
        for t in range(len(tf)):
            fx, fyv = tf[t][1], tf[t][0]
            p1 = f"{scratch}10/20/processed//right{fyv}/{fx}.npy"
            p2 = f"{scratch}11/20/processed/0/left{fx}.npy"
            st[f"{tag}_saved{t}"] = np.array(os.path.exists(p1))
            if os.path.exists(p1):
                a, b = np.load(p1), np.load(p2)
                assert np.array_equal(a, b)
                st[f"{tag}_preds{t}"] = np.asarray(a, dtype=np.float32)
                st[f"{tag}_name{t}"] = np.array([str(fyv), str(fx)])
        print("subtiles", tag, len(sess.feeds), [bool(st[f"{tag}_saved{t}"]) for t in range(len(tf))],
              [str(st[f"{tag}_name{t}"]) for t in range(len(tf)) if f"{tag}_name{t}" in st])
    np.savez_compressed(os.path.join(OUT, "reseg_subtiles.npz"), **st)

    # ---- recreate_resegmented_tifs / mosaic_subtiles ------------------------------------------------------------------
    mo = {}
    for tag, (seed, shape, size, size_y, ud) in {"a": (61, (150, 170), 90, 46, False), "b": (62, (330, 320), 90, 46, True),
                                                 "c": (63, (618, 618), 670, 206, False)}.items():
        set_geometry(RS, size, size_y)
        wins = synth_reseg_windows(seed, shape, size, size_y, ud)
        folder = f"{scratch}mos_{tag}/"
        write_folder(folder, wins)
        order = listing_order(folder, RS)
        preds, sums = RS.recreate_resegmented_tifs(folder, shape)
        mo[f"{tag}_cfg"] = np.array([seed, shape[0], shape[1], size, size_y, int(ud)])
        mo[f"{tag}_order"] = np.array([["nlrud".index(k), x, y] for k, x, y in order])
        if preds.size > 120000:
            mo[f"{tag}_preds_sub"] = preds[::2, ::2].astype(np.float32)
            mo[f"{tag}_nodata"] = np.packbits(preds == 255)
            mo[f"{tag}_sum"] = np.float64(np.where(preds == 255, 0, preds).astype(np.float64).sum())
        else:
            mo[f"{tag}_preds"] = preds.astype(np.float32)
        mo[f"{tag}_sums_sub"] = np.asarray(sums)[::3, ::3].astype(np.float32)
        print("mosaic", tag, preds.shape, float(np.mean(preds == 255)), float(np.nanmean(np.where(preds == 255, np.nan, preds))))
    np.savez_compressed(os.path.join(OUT, "reseg_mosaic.npz"), **mo)

    os.chdir(ROOT)
    shutil.rmtree(scratch, ignore_errors=True)
    for fn in sorted(os.listdir(OUT)):
        if fn.startswith("reseg"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KB")



def gen_border(tags=("s", "d")):
    """resegment_border (:847-1161) end to end on two synthetic neighbours, with every IO function of the reference
    replaced by the arrays it would have loaded; the arguments of its process_subtiles call are captured."""
    import random
    import pandas as pd
    from tests.helpers import synth_border_pair
    J, CR = ref_harness.load()
    import resegment_tiles_wide as RS
    scratch = tempfile.mkdtemp(prefix="ttc_resegb_") + "/"
    os.chdir(scratch)      # keep the reference's debug dumps (tiles.npy, mosaic.npy, ...) out of the repository
    RS.args = types.SimpleNamespace(local_path=scratch, year=2020, process_all=True, resmooth=False, s3_bucket="none")
    RS.x, RS.y = "10", "20"
    RS.predict_logits, RS.predict_inp, RS.predict_length = "logits", "inp", "len"
    RS.superresolve_logits, RS.superresolve_inp, RS.superresolve_inp_bilinear = "sl", "si", "sb"
    RS.data = pd.DataFrame({"X_tile": [10, 11], "Y_tile": [20, 20]})
    RS.AWSKEY = RS.AWSSECRET = None

    def _boom(*a, **k):
        raise IOError("no raster")
    CR.adjust_cloudmask_in_forests = _boom
    CR.mask_nonurban_areas = _boom

    class PSess:
        def run(self, op, feed_dict):
            return fake_model(feed_dict["inp"])

    class SRSess:
        def run(self, op, feed_dict):
            return fake_dsen2(feed_dict["si"], feed_dict["sb"])
    RS.predict_sess, RS.gap_sess, RS.superresolve_sess = PSess(), None, SRSess()
    RS.check_if_processed = lambda *a, **k: True
    RS.download_raw_tile = lambda *a, **k: None
    RS.update_ard_tiles = lambda *a, **k: None
    RS.check_n_tiles = lambda *a, **k: (0, 0)
    out = {}
    for tag in tags:
        seed, T, X, Y, size, size_y, same = {"s": (81, 7, 330, 150, 114, 134, True), "d": (82, 6, 300, 140, 114, 134, False)}[tag]
        set_geometry(RS, size, size_y)
        tile, neighb, tif_t, tif_n = synth_border_pair(seed, T, X, Y, same)
        by_id = {"10": (tile, tif_t), "11": (neighb, tif_n)}
        RS.load_tif = lambda tid, lp: (by_id[str(tid[0])][1].copy(), 0)
        RS.load_dates = lambda tx, ty, lp: list(by_id[str(tx)][0]["dates"])

        def _pt(tx, ty, data, lp, bbx, _m=by_id):
            d = _m[str(tx)][0]
            return (d["s2"].copy(), np.array(d["dates"]), d["interp"].copy(), d["s1"].copy(), d["dem"].copy(), None, None)
        RS.process_tile = _pt
        cap = {}
        orig = RS.process_subtiles

        def _capture(x, y, s2, dates, interp, s1, dem, sess, gap_sess, tiles_folder, tiles_array, right_all, left_all, hist_align, min_clear):
            cap.update(strip=np.array(s2, copy=True), dates=np.array(dates), interp=np.array(interp, copy=True), s1=np.array(s1, copy=True),
                       dem=np.array(dem, copy=True), tf=np.array(tiles_folder), ta=np.array(tiles_array), right_all=np.array(right_all),
                       left_all=np.array(left_all), hist_align=bool(hist_align), min_clear=np.array(min_clear))
            return orig(x, y, s2, dates, interp, s1, dem, sess, gap_sess, tiles_folder, tiles_array, right_all, left_all, hist_align, min_clear)
        RS.process_subtiles = _capture
        shutil.rmtree(scratch + "10", ignore_errors=True); shutil.rmtree(scratch + "11", ignore_errors=True)
        random.seed(11)
        try:
            res = RS.resegment_border("10", "20", "right", scratch, [0, 0, 1, 1], [1, 0, 2, 1], 2, [0.0, 0.0, 0.0, 0.0])
        finally:
            RS.process_subtiles = orig
        out[f"{tag}_cfg"] = np.array([seed, T, X, Y, size, size_y, int(same)])
        out[f"{tag}_result"] = np.array([res[0], res[4]])
        out[f"{tag}_dates"] = cap["dates"]; out[f"{tag}_hist_align"] = np.array(cap["hist_align"])
        out[f"{tag}_strip_sub"] = cap["strip"][:, ::9, ::5, :].astype(np.float32)
        out[f"{tag}_strip_sum"] = np.float64(cap["strip"].astype(np.float64).sum())
        out[f"{tag}_interp_sub"] = cap["interp"][:, ::4, ::4].astype(np.float32)
        out[f"{tag}_min_clear"] = cap["min_clear"].astype(np.int16)
        out[f"{tag}_ta"], out[f"{tag}_tf"] = cap["ta"], cap["tf"]
        for t in range(len(cap["tf"])):
            p1 = f"{scratch}10/20/processed//right{cap['tf'][t][0]}/{cap['tf'][t][1]}.npy"
            out[f"{tag}_saved{t}"] = np.array(os.path.exists(p1))
            if os.path.exists(p1):
                out[f"{tag}_preds{t}"] = np.asarray(np.load(p1), dtype=np.float32)
        print("border", tag, res[0], res[4], cap["strip"].shape, cap["dates"], cap["hist_align"], [bool(out[f"{tag}_saved{t}"]) for t in range(len(cap["tf"]))])
    np.savez_compressed(os.path.join(OUT, "reseg_border.npz"), **out)
    os.chdir(ROOT)
    shutil.rmtree(scratch, ignore_errors=True)
    print("reseg_border.npz", os.path.getsize(os.path.join(OUT, "reseg_border.npz")) // 1024, "KB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "border":
        gen_border()
    else:
        main()
        gen_border()
