"""Golden vectors for the shape reconciliation at the top of process_tile (THIS container only; run with /opt/conda/bin/python3.9).

    /opt/conda/bin/python3.9 tools/gen_golden_shapes.py      # writes tests/golden/adjust_shape.npz, process_tile_shapes.npz

1. the reference's own adjust_shape (src/download_and_predict_job.py:260-310) on every rank it accepts and on differences of
   -4 .. +4 per axis (the odd differences of 3 are recorded with the WRONG length the reference leaves there);
2. the reference's process_tile (job.py:641-995, file loader replaced by a dict like tools/gen_golden.py does) on raw arrays whose
   10 m bands, Sentinel-1 and DEM are a pixel or two off the 20 m grid -- pins that Sentinel-1 is scaled and the DEM filtered BEFORE
   adjust_shape re-grids them (:699-721).
The fixtures are data; no reference source is stored."""
import importlib
import os
import random
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import ref_harness  # noqa: E402

synth = importlib.import_module("sentinel-tree-cover_amd.synth")
OUT = os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))

# (rows, cols) differences of the 10 m bands / Sentinel-1 / the DEM against the 20 m grid, per process_tile case
SHAPE_CASES = {"a": dict(seed=93, T=5, w20=48, h20=52, d10=(1, -1), ds1=(2, 1), ddem=(-1, 2)),
               "b": dict(seed=94, T=4, w20=51, h20=45, d10=(-1, 2), ds1=(-2, -1), ddem=(1, -4))}      # odd 20 m grid: the 40 m branch too


def main():
    J, CR = ref_harness.load()
    scratch = tempfile.mkdtemp(prefix="ttc_golden_shapes_")
    os.chdir(scratch)
    rng = np.random.default_rng(77)
    adj = {}
    k = 0
    for nd in (2, 3, 4):
        for dx in range(-4, 5):
            for dy in (-4, -3, -2, -1, 0, 1, 2, 3, 4):
                if (k % 3) and not (dx in (-1, 0, 1, 2) and dy in (-2, -1, 0, 1)):
                    k += 1
                    continue                                    # thin out the far corners, keep every small case
                k += 1
                n1, n2 = 12 + dx, 10 + dy
                shape = {2: (n1, n2), 3: (3, n1, n2), 4: (2, n1, n2, 3)}[nd]
                a = rng.random(shape).astype(np.float32)
                r = np.asarray(J.adjust_shape(a.copy(), 12, 10))
                tag = f"c{len(adj) // 3}"
                adj[tag + "_in"], adj[tag + "_out"], adj[tag + "_want"] = a, r, np.array([12, 10])
    # T = 1: the squeeze at the end drops the date axis (process_tile re-adds it, :724-727)
    a = rng.random((1, 13, 9, 4)).astype(np.float32)
    tag = f"c{len(adj) // 3}"
    adj[tag + "_in"], adj[tag + "_out"], adj[tag + "_want"] = a, np.asarray(J.adjust_shape(a.copy(), 12, 10)), np.array([12, 10])
    adj["n"] = np.array(len([k for k in adj if k.endswith("_in")]))
    np.savez_compressed(os.path.join(OUT, "adjust_shape.npz"), **adj)
    print("adjust_shape cases:", int(adj["n"]))

    def _boom(*a, **k):
        raise IOError("no raster")
    CR.adjust_cloudmask_in_forests = _boom
    CR.mask_nonurban_areas = _boom
    pt = {}
    for tag, cfg in SHAPE_CASES.items():
        raw = synth.misshape_raw(synth.synth_raw_files(cfg["seed"], cfg["T"], cfg["w20"], cfg["h20"], False), cfg["d10"], cfg["ds1"], cfg["ddem"])
        raw["clouds"] = np.zeros((cfg["T"], 2 * cfg["w20"], 2 * cfg["h20"]), np.float32)

        def _key(path):
            for k, v in {"clouds/clouds_": "clouds", "clouds/cloudmask_": "clm", "raw/s1/": "s1", "raw/s2_10/": "s2_10",
                         "raw/s2_20/": "s2_20", "misc/dem_": "dem", "misc/s2_dates_": "dates"}.items():
                if k in path:
                    return v
            raise KeyError(path)
        J.hkl.load = lambda path, _r=raw: np.array(_r[_key(path)], copy=True)
        folder = os.path.join(scratch, f"pt_{tag}") + "/"
        os.makedirs(folder + "10/20/raw/clouds/", exist_ok=True)
        random.seed(4)
        s2o, do, io, s1o, demo, cso, snowo = J.process_tile(10, 20, None, folder, [0, 0, 1, 1], make_shadow=True)
        pt[f"{tag}_cfg"] = np.array([cfg["seed"], cfg["T"], cfg["w20"], cfg["h20"], *cfg["d10"], *cfg["ds1"], *cfg["ddem"]])
        pt[f"{tag}_dates"] = np.asarray(do)
        pt[f"{tag}_s2_sub"] = s2o[:, ::3, ::3, :].astype(np.float32)
        pt[f"{tag}_s2_edges"] = np.concatenate([s2o[:, :2].reshape(s2o.shape[0], -1), s2o[:, -2:].reshape(s2o.shape[0], -1),
                                                s2o[:, :, :2].reshape(s2o.shape[0], -1), s2o[:, :, -2:].reshape(s2o.shape[0], -1)], 1).astype(np.float32)
        pt[f"{tag}_interp_sub"] = io[:, ::2, ::2].astype(np.float32)
        pt[f"{tag}_s1"] = s1o.astype(np.float32)
        pt[f"{tag}_dem"] = demo.astype(np.float32)
        pt[f"{tag}_cloudshad"] = np.packbits(cso > 0)
        pt[f"{tag}_cloudshad_shape"] = np.array(cso.shape)
        pt[f"{tag}_snow"] = np.packbits(np.asarray(snowo) > 0)
        print("process_tile (shapes)", tag, s2o.shape, s1o.shape, demo.shape, do)
    np.savez_compressed(os.path.join(OUT, "process_tile_shapes.npz"), **pt)
    os.chdir(ROOT)
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
