"""Pin the CPU oracle (oracle/restate_numpy.py) against golden vectors captured by
RUNNING the reference (tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import restate_numpy as O
from tests.helpers import e2e_inputs, fake_dsen2, fake_model, golden, synth


def test_codecs():
    g = golden("codecs.npz")
    np.testing.assert_array_equal(O.to_float32(g["u16"]), g["to_float32"])
    np.testing.assert_array_equal(O.to_int16(g["f32"]), g["to_int16"])
    np.testing.assert_allclose(O.convert_to_db(g["db_in"].copy()), g["db_out"], rtol=0, atol=0)
    np.testing.assert_allclose(O.s1_to_db(g["s1_u16"]), g["s1_db"], rtol=0, atol=0)


def test_indices():
    g = golden("indices.npz")
    x = g["x"]
    for name in ("evi", "bi", "msavi2", "grndvi", "make_indices"):
        np.testing.assert_array_equal(getattr(O, name)(x), g[name], err_msg=name)


def test_regrid_matrix_and_data():
    g = golden("regrid.npz")
    for k in range(int(g["n"])):
        dates = g[f"dates_{k}"]
        R = O.regrid_matrix(dates)
        np.testing.assert_allclose(R, g[f"R_{k}"], rtol=0, atol=1e-7, err_msg=f"set {k}")
        np.testing.assert_allclose(R.sum(1), 1.0, atol=1e-6)
        np.testing.assert_allclose(O.regrid(g[f"data_{k}"], dates), g[f"out_{k}"], rtol=0, atol=2e-7)


def test_whittaker():
    g = golden("whittaker.npz")
    # the reference solves in float32 (splu): agreement is bounded by its own rounding
    np.testing.assert_allclose(O.whittaker_interpolate(g["y"]), g["z"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(O.whittaker_interpolate(g["y4"]), g["z4"], rtol=0, atol=5e-5)
    M = O.whittaker_monthly_matrix()
    np.testing.assert_allclose(M.sum(1), 1.0, atol=1e-12)        # constants are preserved


def test_smooth_large_tile():
    g = golden("smooth_large_tile.npz")
    out, dates, interp = O.smooth_large_tile(g["s2"].copy(), g["dates"].copy(), g["interp"].copy())
    np.testing.assert_array_equal(dates, g["dates_out"])
    np.testing.assert_array_equal(interp, g["interp_out"])
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=5e-5)


def test_window_grid():
    g = golden("window_grid.npz")
    for k in range(int(g["n"])):
        H, W, size = g[f"hws_{k}"]
        folder, array = O.window_grid(int(H), int(W), int(size))
        np.testing.assert_array_equal(folder, g[f"folder_{k}"])
        np.testing.assert_array_equal(array, g[f"array_{k}"])


def test_bright_surface():
    g = golden("bright.npz")
    img = synth.synth_bright_window(int(g["seed"]))
    np.testing.assert_allclose(O.identify_bright_bare_surfaces(img), g["out"], atol=1e-7)
    assert g["out"].min() == 0.0 and g["out"].max() == 1.0
    none = O.identify_bright_bare_surfaces(np.full((5, 172, 172, 17), 0.1, np.float32))
    np.testing.assert_array_equal(none, g["out_none"])
    assert np.all(none == 1.0)


def test_normalize():
    g = golden("normalize.npz")
    np.testing.assert_array_equal(O.normalize_subtile(g["x"].copy()), g["y"])


@pytest.mark.parametrize("tag", ["e2e_clear", "e2e_cloudy"])
def test_process_subtiles_and_mosaic_end_to_end(tag):
    """Reference process_subtiles (36 windows, fake session) + load_mosaic_predictions."""
    g = golden(f"{tag}.npz")
    s2, dates, interp, s1, dem = e2e_inputs(g)
    pf = lambda w: O.predict_subtile(w, fake_model, 158)
    wins, feeds = O.process_subtiles(s2, dates, interp, s1, dem, pf, size=158, length=4, return_inputs=True)
    keys = [tuple(k) for k in g["keys"]]
    assert sorted(wins.keys()) == keys
    assert len(feeds) == int(g["n_feeds"])
    # model inputs the reference fed (strided sample), in call order == window order
    order = [k for k in [(int(fy), int(fx)) for fx in sorted({k[1] for k in keys}) for fy in sorted({k[0] for k in keys})]
             if k in feeds]
    got = np.stack([feeds[k][:, ::19, ::19, :] for k in order])
    np.testing.assert_allclose(got, g["feeds_sub"], rtol=0, atol=2e-4)
    stack = np.stack([wins[k] for k in keys])
    ref = g["windows_permille"].astype(np.float64) / 1000.0
    bad = np.abs(stack - ref) > 1.5e-3              # allow one rounding quantum (np.around(.,3))
    assert bad.mean() < 1e-4, bad.mean()
    assert np.array_equal(stack > 1.0, ref > 1.0) or bad.mean() < 1e-4
    # mosaic from the REFERENCE's windows -> must equal the reference's mosaic exactly-ish
    ref_wins = {k: (g["windows_permille"][i] / 1000.0).astype(np.float32) for i, k in enumerate(keys)}
    mos = O.mosaic_predictions(ref_wins, size=158)
    assert mos.shape == g["mosaic"].shape and mos.dtype == np.uint8
    diff = np.abs(mos.astype(int) - g["mosaic"].astype(int))
    assert (diff > 1).mean() < 1e-5 and (diff > 0).mean() < 1e-2, ((diff > 1).mean(), (diff > 0).mean())


def test_mosaic_with_nodata():
    g = golden("mosaic.npz")
    keys = [tuple(k) for k in g["keys"]]
    wins = {k: (g["windows_permille"][i] / 1000.0).astype(np.float32) for i, k in enumerate(keys)}
    mos = O.mosaic_predictions(wins, size=158)
    diff = np.abs(mos.astype(int) - g["mosaic"].astype(int))
    assert (diff > 1).mean() < 1e-5 and (diff > 0).mean() < 1e-2, ((diff > 1).mean(), (diff > 0).mean())
    assert (g["mosaic"] == 255).any()


def test_superresolve_tiling_quirks():
    g = golden("superresolve_tiling.npz")
    arr = np.random.default_rng(int(g["seed"])).random((2, 618, 618, 10)).astype(np.float32)
    inp = arr.copy()
    res = O.superresolve_large_tile(arr, fake_dsen2)
    np.testing.assert_allclose(res[:, ::7, ::7, :], g["out_sub"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res[0, :, :, 4], g["out_band4_full"], rtol=0, atol=1e-6)
    # the dead branch: x in [0,507], y in [550,617] is never refined
    np.testing.assert_array_equal(res[:, :508, 550:, 4:], inp[:, :508, 550:, 4:])
    assert not np.allclose(res[:, :508, :550, 4:], inp[:, :508, :550, 4:])


def test_float_to_int16_matches_reference():
    g = golden("float_to_int16.npz")
    np.testing.assert_array_equal(O.float_to_int16(g["x"]), g["y"])


def test_feature_mosaic_matches_reference():
    g = golden("mosaic_features.npz")
    wins = {tuple(int(v) for v in k): g["windows"][i] for i, k in enumerate(g["keys"])}
    got = O.mosaic_features(wins, size=30, depth=16)
    assert got.shape == g["mosaic"].shape and got.dtype == np.int16
    d = np.abs(got.astype(int) - g["mosaic"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-2, (d.max(), (d > 0).mean())     # float32 summation follows os.listdir order


def test_resize_matches_real_skimage():
    """oracle.resize_bilinear and the package's table builder (resegment._resize) against skimage.transform.resize(order=1) of the REAL
    scikit-image 0.18.3 on the shapes of job.py:741-781 (20 m -> 10 m, float32) and resegment_tiles_wide.py:1190-1236, :1354-1355
    (border-mosaic weight tables, float64, incl. the anti-aliased downsampling).  Float64 tables: 1e-12 (measured <= 1.3e-13).
    Float32 bands: skimage 0.18's warp evaluates coordinates and weights in float32 -- 1 ulp on the x2 grids (measured 6e-8) and
    1.4e-5 on the odd-grid 154 -> 617 branch (x4.006: a float32 coordinate near 150 carries 1.5e-5); scikit-image >= 0.19
    (what the reference's python3.11 Dockerfile installs) evaluates in float64 like the oracle."""
    import importlib
    RS = importlib.import_module("sentinel-tree-cover_amd.resegment")
    g = golden("resize.npz")
    assert str(g["skimage_version"]) == "0.18.3"
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from resize_cases import cases as resize_cases          # the inputs are code (the fixture holds outputs + a checksum per input)
    cases = resize_cases()
    assert len(cases) == int(g["n"])
    worst32, worst64 = 0.0, 0.0
    for k, (a, shape) in enumerate(cases):
        want = g[f"out_{k}"]
        assert tuple(int(v) for v in g[f"shape_{k}"]) == tuple(shape)
        chk = np.array([np.asarray(a, np.float64).sum(), np.abs(np.asarray(a, np.float64)).max()])
        np.testing.assert_allclose(chk, g[f"insum_{k}"], rtol=1e-12, err_msg=f"input {k} is not the one the fixture was generated from")
        got = O.resize_bilinear(a, shape)
        assert got.shape == want.shape
        e = float(np.abs(got - want).max())
        if a.dtype == np.float32:
            x2 = shape[0] == 2 * a.shape[0]
            assert e < (1.5e-7 if x2 else 2e-5), (k, e)
            worst32 = max(worst32, e)
        else:
            assert e < 1e-12, (k, e)
            assert float(np.abs(RS._resize(a, shape) - want).max()) < 1e-12, k
            worst64 = max(worst64, e)
    print(f"resize vs scikit-image 0.18.3: float32 bands {worst32:.2e}, float64 tables {worst64:.2e}")
