"""Generate golden vectors by RUNNING THE REFERENCE (this container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz

Imports wri/sentinel-tree-cover's own functions from /root/reference through
tools/ref_harness.py (third-party stubs only), feeds them seeded synthetic
inputs and stores inputs (or the seed that regenerates them) + outputs as small
.npz fixtures.  The fixtures are data; no reference source is stored.
Fixtures pin oracle/restate_numpy.py (tests/test_oracle_golden.py) and, through
the oracle, the HIP path.
"""
import importlib
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import ref_harness  # noqa: E402

synth = importlib.import_module("sentinel-tree-cover_amd.synth")
from tests.helpers import fake_model, fake_dsen2  # noqa: E402
OUT = os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))


class FakeSess:
    def __init__(self, J):
        self.J, self.feeds = J, []

    def run(self, op, feed_dict):
        x = feed_dict[self.J.predict_inp]
        self.feeds.append(np.array(x, copy=True))
        return fake_model(x)


def main():
    os.makedirs(OUT, exist_ok=True)
    J, CR = ref_harness.load()
    rng = np.random.default_rng(2024)
    scratch = tempfile.mkdtemp(prefix="ttc_golden_")
    os.chdir(scratch)

    # ---- codecs -------------------------------------------------------------------
    u = rng.integers(2, 65535, size=(3, 9, 7, 4)).astype(np.uint16)
    f = rng.random((3, 9, 7, 2)).astype(np.float32)
    from src.tof import tof_downloading as TD  # noqa
    s1u = rng.integers(0, 65536, size=(3, 11, 13, 2)).astype(np.uint16)
    s1u[0, :3, :3] = 65535
    s1f = np.float32(s1u) / 65535
    for i in range(s1f.shape[0]):
        s1_i = s1f[i]
        s1_i[s1_i == 1] = np.median(s1_i[s1_i < 65535], axis=0)
        s1f[i] = s1_i
    s1db = s1f.copy()
    s1db[..., -1] = J.convert_to_db(s1db[..., -1], 22)
    s1db[..., -2] = J.convert_to_db(s1db[..., -2], 22)
    fi = np.concatenate([(np.random.default_rng(99).random(500) * 80 - 40).astype(np.float32),
                         np.float32([np.nan, 0, -0.0, 32.767, 32.7675, -32.768, -32.7685, 1e9, -1e9, 0.0004999, 0.0005, -0.0009999, 12.3456])])
    np.savez_compressed(os.path.join(OUT, "float_to_int16.npz"), x=fi, y=J.float_to_int16(fi.copy()))
    np.savez_compressed(os.path.join(OUT, "codecs.npz"), u16=u, to_float32=TD.to_float32(u),
                        f32=f, to_int16=TD.to_int16(f), db_in=f, db_out=J.convert_to_db(f.copy(), 22),
                        s1_u16=s1u, s1_db=s1db.astype(np.float32))

    # ---- indices ------------------------------------------------------------------
    x = (rng.random((5, 16, 16, 10)) * 1.3 - 0.15).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "indices.npz"), x=x, evi=J.evi(x), bi=J.bi(x),
                        msavi2=J.msavi2(x), grndvi=J.grndvi(x), make_indices=J.make_indices(x))

    # ---- regrid: one-hot probes of calculate_and_save_best_images + data run -----
    date_sets = [
        [5, 40, 100, 160, 220, 280, 340],
        [-20, 12, 33, 95, 130, 171, 200, 244, 290, 301, 350, 380],
        [100, 130, 160, 200],
        [3, 17, 18, 64, 180, 181, 182, 359],
        [120, 150],
        [200],
        [-150, -90, 10, 95, 100, 104, 300],
        list(synth.synth_dates(np.random.default_rng(7), 12)),
    ]
    reg = {}
    for k, ds in enumerate(date_sets):
        T = len(ds)
        onehot = np.eye(T, dtype=np.float32).reshape(T, 1, 1, T)
        steps, maxd = J.calculate_and_save_best_images(onehot, np.array(ds))
        reg[f"dates_{k}"] = np.array(ds)
        reg[f"R_{k}"] = steps.reshape(24, T)
        data = rng.random((T, 6, 5, 3)).astype(np.float32)
        reg[f"data_{k}"] = data
        reg[f"out_{k}"] = J.calculate_and_save_best_images(data, np.array(ds))[0]
    np.savez_compressed(os.path.join(OUT, "regrid.npz"), n=len(date_sets), **reg)

    # ---- Whittaker ----------------------------------------------------------------
    y = rng.random((24, 8, 6, 10)).astype(np.float32)
    sm = J.Smoother(lmbd=100, size=24, nbands=10, dimx=8, dimy=6, outsize=12)
    y4 = rng.random((24, 5, 5, 4)).astype(np.float32) * 2 - 1
    sm4 = J.Smoother(lmbd=100, size=24, nbands=4, dimx=5, dimy=5, outsize=12)
    np.savez_compressed(os.path.join(OUT, "whittaker.npz"), y=y, z=sm.interpolate_array(y.copy()),
                        y4=y4, z4=sm4.interpolate_array(y4.copy()))

    # ---- smooth_large_tile (with zeros / ones / a missing date) -------------------
    s2, dates, interp, s1, dem = synth.synth_tile(seed=11, T=9, H=24, W=20, cloud_frac=0.1)
    s2[2, 3:6, 4:9, :] = 0.0           # zeros -> temporal median
    s2[5, 10:12, 1:3, 2] = 1.0         # ones  -> temporal median
    s2[7, :20, :, :] = 0.0             # a mostly-missing date -> dropped (>= H^2/10 px)
    o, d2, i2 = J.smooth_large_tile(s2.copy(), dates.copy(), interp.copy())
    np.savez_compressed(os.path.join(OUT, "smooth_large_tile.npz"), s2=s2, dates=dates, interp=interp,
                        out=o, dates_out=np.array(d2), interp_out=i2)

    # ---- window grid -------------------------------------------------------------
    grids = {}
    for k, (H, W, size) in enumerate([(618, 618, 158), (618, 618, 154), (618, 602, 158), (400, 330, 158)]):
        gx = int(np.ceil((H - size) / 5))
        gy = int(np.ceil((W - size) / 5))
        xs = np.hstack([np.arange(0, H - size, gx), np.array(H - size)])
        ys = np.hstack([np.arange(0, W - size, gy), np.array(W - size)])
        folder = np.array([[a, b, size, size] for a in xs for b in ys])
        grids[f"hws_{k}"] = np.array([H, W, size])
        grids[f"folder_{k}"] = folder
        grids[f"array_{k}"] = TD.make_overlapping_windows(folder, diff=7)
    np.savez_compressed(os.path.join(OUT, "window_grid.npz"), n=4, **grids)

    # ---- bright-surface attenuation ------------------------------------------------
    img = synth.synth_bright_window(seed=21)
    np.savez_compressed(os.path.join(OUT, "bright.npz"), seed=21,
                        out=J.identify_bright_bare_surfaces(img.copy()).astype(np.float32),
                        out_none=J.identify_bright_bare_surfaces(np.full((5, 172, 172, 17), 0.1, np.float32)).astype(np.float32))

    # ---- normalize ------------------------------------------------------------------
    J.min_all = [0.006576638437476157, 0.0162050812542916, 0.010040436408026246, 0.013351644159609368,
                 0.01965362020294499, 0.014229037918669413, 0.015289539940489814, 0.011993591210803388,
                 0.008239871824216068, 0.006546120393682765, 0.0, 0.0, 0.0, -0.1409399364817101,
                 -0.4973397113668104, -0.09731556326714398, -0.7193834232943873]
    J.max_all = [0.2691233691920348, 0.3740291447318227, 0.5171435111009385, 0.6027466239414053,
                 0.5650263218127718, 0.5747005416952773, 0.5933928435187305, 0.6034943160143434,
                 0.7472037842374304, 0.7000076295109483, 0.4, 0.948334642387533, 0.6729257769285485,
                 0.8177635298774327, 0.35768999002433816, 0.7545951919107605, 0.7602693339366691]
    nx = (rng.random((3, 9, 9, 17)) * 1.6 - 0.6).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "normalize.npz"), x=nx, y=J.normalize_subtile(nx.copy()))

    # ---- process_subtiles + load_mosaic_predictions end to end (618^2, L=4) --------
    for tag, seed, T, cloud in [("e2e_clear", 5, 6, 0.0), ("e2e_cloudy", 6, 5, 0.5)]:
        s2, dates, interp, s1, dem = synth.synth_tile(seed=seed, T=T, H=618, W=618, cloud_frac=cloud)
        if cloud > 0:
            interp[:, 300:520, 380:618] = 1.0      # a region with no clear image at all
            interp[:, 0:120, 500:618] = 1.0        # exercises the job.py:1395 pad quirk (x=0, y=last)
            interp[:, 520:618, 0:200] = 1.0        # ... and x=last rows
        local = os.path.join(scratch, tag) + "/"
        os.makedirs(local + "100/200/", exist_ok=True)
        J.args = types.SimpleNamespace(length=4, local_path=local, s3_bucket="b", gen_composite=False,
                                       make_training_data=False, process=True, gen_feats=False)
        J.predict_inp, J.predict_length, J.predict_logits = "inp", "len", "logits"
        J.year = 2020
        J.uploader = types.SimpleNamespace(upload=lambda **k: None)

        def _dump(obj, path, **k):
            open(path, "wb").close()
        J.hkl = types.SimpleNamespace(dump=_dump)
        J.write_ard_to_tif = lambda *a, **k: None
        sess = FakeSess(J)
        J.process_subtiles(100, 200, s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(),
                           sess, [0, 0, 1, 1], J.SIZE, None)
        proc = local + "100/200/processed/"
        wins = {}
        for a in sorted(os.listdir(proc), key=int):
            for b in sorted(os.listdir(proc + a), key=lambda s: int(s[:-4])):
                wins[(int(a), int(b[:-4]))] = np.load(proc + a + "/" + b)
        keys = np.array(sorted(wins.keys()))
        stack = np.stack([wins[tuple(k)] for k in keys])
        assert np.all(np.abs(stack * 1000 - np.round(stack * 1000)) < 1e-2)
        mosaic = J.load_mosaic_predictions(proc, depth=1)
        feeds = np.stack([f[0, :, ::19, ::19, :] for f in sess.feeds]) if sess.feeds else np.zeros((0,))
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), seed=seed, T=T, cloud=cloud,
                            interp_boxes=np.array([[300, 520, 380, 618], [0, 120, 500, 618], [520, 618, 0, 200]]
                                                  if cloud > 0 else np.zeros((0, 4), int)),
                            keys=keys, windows_permille=np.round(stack * 1000).astype(np.int32),
                            n_feeds=len(sess.feeds), feeds_sub=feeds.astype(np.float32), mosaic=mosaic)
        print(tag, "feeds", len(sess.feeds), "mosaic", mosaic.shape, mosaic.dtype,
              np.unique(mosaic).size, (mosaic == 255).mean())

    # ---- mosaic alone, random windows incl. no-data ---------------------------------
    proc = os.path.join(scratch, "mos") + "/"
    folder = grids["folder_0"]
    mw = {}
    base = synth._smooth_field(np.random.default_rng(3), 618, 618, 40)
    for (a, b, _, _) in folder:
        p = (base[a:a + 158, b:b + 158] + rng.normal(0, 0.03, (158, 158))).clip(0, 1)
        p = np.around(p, 3).astype(np.float32)
        if (a, b) == (92, 184):
            p[:] = 255.
        if (a, b) == (276, 0):
            p[40:90, 10:120] = 255.
        os.makedirs(proc + str(b), exist_ok=True)       # processed/{folder_y}/{folder_x}.npy
        np.save(proc + f"{b}/{a}.npy", p)
        mw[(int(b), int(a))] = p
    mk = np.array(sorted(mw.keys()))
    np.savez_compressed(os.path.join(OUT, "mosaic.npz"), keys=mk,
                        windows_permille=np.round(np.stack([mw[tuple(k)] for k in mk]) * 1000).astype(np.int32),
                        mosaic=J.load_mosaic_predictions(proc, depth=1))

    # ---- feature mosaic (depth > 1 branch of load_mosaic_predictions), small geometry: SIZE 30, 3 x 3 windows, depth 16 ----
    procf = os.path.join(scratch, "mosf") + "/"
    rf = np.random.default_rng(17)
    old_size = J.SIZE
    J.SIZE = 30
    fk, fw = [], []
    for a in (0, 22, 44):            # outer folder
        for b in (0, 20, 44):        # file name
            arr = rf.integers(-3000, 12000, size=(30, 30, 16)).astype(np.int16)
            os.makedirs(procf + str(a), exist_ok=True)
            np.save(procf + f"{a}/{b}.npy", arr)
            fk.append((a, b)); fw.append(arr)
    try:
        fm = J.load_mosaic_predictions(procf, depth=16)
    finally:
        J.SIZE = old_size
    np.savez_compressed(os.path.join(OUT, "mosaic_features.npz"), keys=np.array(fk), windows=np.stack(fw), mosaic=fm)
    print("feature mosaic", fm.shape, fm.dtype)

    # ---- multi-temporal cloud / shadow detection (identify_clouds_shadows + detect_pfcp) -------------------------------
    # the two WorldCover rasters are replaced by synthetic masks through the functions that would read them
    det = {}
    for tag, (seed, T_, H_, W_, with_masks) in {"a": (77, 7, 120, 112, True), "b": (78, 2, 64, 72, False), "c": (79, 4, 96, 96, True)}.items():
        scene = synth.synth_detection_scene(seed, T_, H_, W_)
        img, dem_d, forest_d, core_d, near_d = scene
        if with_masks:
            CR.adjust_cloudmask_in_forests = lambda f, b, d, _m=forest_d: _m.copy()

            def _urban(f, b, pf, _c=core_d, _n=near_d):
                pf[_c == 1] = 1.
                pf[_n == 0] = 0.
                return pf
            CR.mask_nonurban_areas = _urban
        else:
            def _boom(*a, **k):
                raise IOError("no raster")
            CR.adjust_cloudmask_in_forests = _boom
            CR.mask_nonurban_areas = _boom
        cl, fc = CR.identify_clouds_shadows(img.copy(), dem_d.copy(), [0, 0, 1, 1])
        det[f"{tag}_cfg"] = np.array([seed, T_, H_, W_, int(with_masks)])
        det[f"{tag}_clouds"] = np.packbits(cl > 0)
        det[f"{tag}_clouds_max"] = np.float32(cl.max())
        det[f"{tag}_fcps"] = np.packbits(fc)
        print("detection", tag, cl.mean(axis=(1, 2)).round(3), fc.mean().round(4))
    np.savez_compressed(os.path.join(OUT, "cloud_detection.npz"), **det)

    # ---- process_tile as a whole (job.py:641-995): the file loader is replaced by a dict of synthetic raw arrays ---------
    def _boom(*a, **k):
        raise IOError("no raster")
    CR.adjust_cloudmask_in_forests = _boom
    CR.mask_nonurban_areas = _boom
    pt = {}
    for tag, (seed, T_, w20, h20, with_clm) in {"a": (91, 6, 80, 88, True), "b": (92, 7, 100, 96, False)}.items():
        raw = synth.synth_raw_files(seed, T_, w20, h20, with_clm)

        def _key(path):
            for k, v in {"clouds/clouds_": "clouds", "clouds/cloudmask_": "clm", "raw/s1/": "s1", "raw/s2_10/": "s2_10",
                         "raw/s2_20/": "s2_20", "misc/dem_": "dem", "misc/s2_dates_": "dates"}.items():
                if k in path:
                    return v
            raise KeyError(path)
        J.hkl.load = lambda path, _r=raw: np.array(_r[_key(path)], copy=True)
        folder = os.path.join(scratch, f"pt_{tag}") + "/"
        os.makedirs(folder + "10/20/raw/clouds/", exist_ok=True)
        if with_clm:
            open(folder + "10/20/raw/clouds/cloudmask_10X20Y.hkl", "w").close()
        random.seed(4)
        s2o, do, io, s1o, demo, cso, snowo = J.process_tile(10, 20, None, folder, [0, 0, 1, 1], make_shadow=True)
        pt[f"{tag}_cfg"] = np.array([seed, T_, w20, h20, int(with_clm)])
        pt[f"{tag}_dates"] = np.asarray(do)
        pt[f"{tag}_s2_sub"] = s2o[:, ::3, ::3, :].astype(np.float32)
        pt[f"{tag}_interp_sub"] = io[:, ::2, ::2].astype(np.float32)
        pt[f"{tag}_s1_sub"] = s1o[:, ::4, ::4, :].astype(np.float32)
        pt[f"{tag}_dem"] = demo.astype(np.float32)
        pt[f"{tag}_cloudshad"] = np.packbits(cso > 0)
        pt[f"{tag}_cloudshad_shape"] = np.array(cso.shape)
        pt[f"{tag}_snow"] = np.packbits(np.asarray(snowo) > 0)
        print("process_tile", tag, s2o.shape, do, (cso > 0).mean(axis=(1, 2)).round(3))
    np.savez_compressed(os.path.join(OUT, "process_tile.npz"), **pt)

    # ---- cloud gap-fill (stdlib RNG pinned: the reference samples with random.shuffle) -----
    tiles, gdates, probs, pf = synth.synth_gapfill_scene(31, 6, 224, 224)
    ia = CR.id_areas_to_interp(tiles.copy(), probs.copy(), probs.copy(), gdates, pfcps=pf)
    random.seed(7)
    rt, ri, rr = CR.remove_cloud_and_shadows(tiles.copy(), probs.copy(), probs.copy(), gdates, pfcps=pf, sentinel1=None)
    np.savez_compressed(os.path.join(OUT, "gapfill.npz"), seed=31, T=6, H=224, W=224, rng_seed=7,
                        id_areas=ia.astype(np.float32), interp=ri.astype(np.float32),
                        tiles_sub=rt[:, ::3, ::3, :].astype(np.float32), mosaic_sub=np.load("mosaic.npy")[::2, ::2],
                        to_remove=np.array(rr, dtype=np.int64),
                        tiles_sum=np.float64(rt.astype(np.float64).sum()))

    # ---- DSen2 tiling driver with a fake session ------------------------------------
    J.superresolve_logits, J.superresolve_inp, J.superresolve_inp_bilinear = "l", "i", "b"

    class SRSess:
        def run(self, ops, feed_dict):
            return [fake_dsen2(feed_dict["i"], feed_dict["b"])]
    arr = np.random.default_rng(9).random((2, 618, 618, 10)).astype(np.float32)
    res = J.superresolve_large_tile(arr.copy(), SRSess())
    np.savez_compressed(os.path.join(OUT, "superresolve_tiling.npz"), seed=9,
                        out_sub=res[:, ::7, ::7, :], out_band4_full=res[0, :, :, 4].astype(np.float32),
                        checksum=np.float64(res.astype(np.float64).sum()))

    os.chdir(ROOT)
    shutil.rmtree(scratch, ignore_errors=True)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KB")


if __name__ == "__main__":
    random.seed(0)
    main()
