"""GPU parity of the process_tile mirror (job.process_tile: raw arrays -> clean stack) against the golden vectors captured
by running the reference's process_tile with its file loader replaced."""
import random

import numpy as np
import pytest

from tests.helpers import golden, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    from ttc import job, weights as Wt
    return job.TTCSession(Wt.synth_weights(0), win_in=44, length=4, dsen2_weights=None)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_process_tile_matches_reference(sess, tag):
    from ttc import job
    g = golden("process_tile.npz")
    seed, T, w20, h20, with_clm = (int(v) for v in g[f"{tag}_cfg"])
    raw = synth.synth_raw_files(seed, T, w20, h20, bool(with_clm))
    random.seed(4)                                           # the gap-fill replays the reference's stdlib RNG stream
    s2, dates, interp, s1, dem, cloudshad, snow = job.process_tile(raw, sess)
    np.testing.assert_array_equal(dates, g[f"{tag}_dates"])
    shp = tuple(int(v) for v in g[f"{tag}_cloudshad_shape"])
    want_cs = np.unpackbits(g[f"{tag}_cloudshad"])[:np.prod(shp)].reshape(shp).astype(bool)
    np.testing.assert_array_equal(cloudshad.cpu().numpy() > 0, want_cs)
    np.testing.assert_array_equal(interp.cpu().numpy()[:, ::2, ::2], g[f"{tag}_interp_sub"])
    np.testing.assert_array_equal(dem.cpu().numpy(), g[f"{tag}_dem"])
    sn = snow.cpu().numpy()
    np.testing.assert_array_equal(sn > 0, np.unpackbits(g[f"{tag}_snow"])[:sn.size].reshape(sn.shape).astype(bool))
    e1 = np.abs(s1.cpu().numpy()[:, ::4, ::4, :] - g[f"{tag}_s1_sub"]).max()
    e2 = np.abs(s2.cpu().numpy()[:, ::3, ::3, :] - g[f"{tag}_s2_sub"])
    print(f"[parity] process_tile {tag}: s1 max|d| = {e1:.2e}, s2 max|d| = {e2.max():.2e} (mean {e2.mean():.2e})")
    assert e1 < 2e-6                                          # log10f vs numpy's log10
    assert e2.max() < 1e-5 and e2.mean() < 1e-7              # measured 7.2e-7 / 6.4e-9 (NNLS from Gram matrices vs scipy's nnls)


@pytest.mark.parametrize("case", ["missing_date", "three_dates", "odd_20m_grid", "no_shadow"])
def test_process_tile_edge_cases_vs_oracle(sess, case):
    """dropped dates (more than half of a date missing), a short stack, an odd 20 m grid (the 40 m bands take the
    reference's odd-grid branches), and make_shadow=False -- against the oracle of the same flow"""
    from oracle import restate_tile as P
    from ttc import job
    if case == "missing_date":
        raw = synth.synth_raw_files(93, 6, 60, 56, False)
        raw["s2_10"][4, : 100] = 0                              # > 50 % of date 4 is missing -> id_missing_px drops it
    elif case == "three_dates":
        raw = synth.synth_raw_files(94, 3, 60, 56, True)
    elif case == "odd_20m_grid":
        raw = synth.synth_raw_files(95, 5, 61, 57, False)
    else:
        raw = synth.synth_raw_files(96, 5, 60, 56, True)
    if case == "no_shadow":
        got = job.process_tile(raw, sess, make_shadow=False)
        s2 = O_upsampled(raw)
        assert np.abs(got[0].cpu().numpy() - np.clip(s2, 0, 1)).max() < 1e-6 and float(got[2].abs().max()) == 0.0
        return
    random.seed(7)
    want = P.process_tile_arrays(raw)
    random.seed(7)
    got = job.process_tile(raw, sess)
    np.testing.assert_array_equal(got[1], want[1])
    assert len(got[1]) < len(raw["dates"]) or case != "missing_date"
    np.testing.assert_array_equal(got[5].cpu().numpy() > 0, want[5] > 0)
    np.testing.assert_array_equal(got[2].cpu().numpy(), want[2])
    e = np.abs(got[0].cpu().numpy() - want[0])
    print(f"[parity] process_tile {case}: dates {list(got[1])}, s2 max|d| = {e.max():.2e}")
    assert e.max() < 1e-5 and e.mean() < 1e-7               # measured <= 3e-7


def O_upsampled(raw):
    from oracle import restate_numpy as R
    return R.upsample_20m(R.to_float32(raw["s2_10"]), R.to_float32(raw["s2_20"]))


def test_adjust_shape_matches_reference_rules():
    from ttc import job
    a = np.arange(2 * 7 * 9 * 3, dtype=np.float32).reshape(2, 7, 9, 3)
    assert job.adjust_shape(a, 6, 8).shape == (2, 6, 8, 3)
    np.testing.assert_array_equal(job.adjust_shape(a, 6, 8), a[:, 1:, 1:])
    assert job.adjust_shape(a, 8, 10).shape == (2, 8, 10, 3)
    np.testing.assert_array_equal(job.adjust_shape(a, 8, 10)[:, 1:, 1:], a)
    np.testing.assert_array_equal(job.adjust_shape(a[0, ..., 0], 3, 5), a[0, 2:-2, 2:-2, 0])


def test_raw_to_raster_end_to_end_vs_oracle():
    """uint16 raw arrays -> uint8 tree-cover raster, every stage on the device, against the chained CPU oracle:
    process_tile -> superresolve_large_tile -> process_subtiles -> load_mosaic_predictions (W = 44 windows on a 160 x 176 tile)."""
    import torch
    from oracle import restate_model as M, restate_numpy as O, restate_tile as P
    from ttc import job, weights as Wt
    W, size, L = 44, 30, 4
    w = Wt.synth_weights(0)
    sess = job.TTCSession(w, win_in=W, length=L)
    raw = synth.synth_raw_files(91, 6, 80, 88, True)
    random.seed(4)
    s2, dates, interp, s1, dem, _, _ = P.process_tile_arrays(raw)
    ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    s2 = O.superresolve_large_tile(s2, ds)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    wins = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), lambda x: O.predict_subtile(x, net, size), size=size, length=L)
    want_u8, want_f = O.mosaic_predictions(wins, size=size, return_float=True)
    random.seed(4)
    g_s2, g_dates, g_interp, g_s1, g_dem, _, _ = job.process_tile(raw, sess)
    sess.ctx.superresolve_tile(g_s2, quirks=True)
    e = np.abs(g_s2.cpu().numpy() - s2)
    print(f"[parity] raw -> clean + super-resolved stack: max|d| = {e.max():.2e}")
    assert e.max() < 1e-5                                   # measured 3e-7
    got_f, got_u8 = job.predict_tile(g_s2, g_dates, g_interp, g_s1, g_dem, sess, size=size)
    assert got_u8.shape == want_u8.shape
    assert np.array_equal(np.isnan(got_f), np.isnan(want_f))
    d = np.abs(got_u8.astype(int) - want_u8.astype(int))
    print(f"[parity] raw -> uint8 raster: > 1 count {np.mean(d > 1):.2e}, == 1 count {np.mean(d == 1):.2e}")
    assert (d > 1).mean() < 1e-4 and (d > 0).mean() < 3e-2


@pytest.mark.parametrize("tag", ["a", "b"])
def test_process_tile_reconciles_shapes_like_reference(sess, tag):
    """the staged mirror on raw arrays a pixel or two off the 20 m grid, against the REFERENCE's process_tile (process_tile_shapes.npz,
    tools/gen_golden_shapes.py): Sentinel-1 scaled and the DEM filtered on their own grids, THEN re-gridded (job.py:699-721) -- on the device
    (ttc_adjust_shape)"""
    from ttc import job
    g = golden("process_tile_shapes.npz")
    seed, T, w20, h20, *d = (int(v) for v in g[f"{tag}_cfg"])
    raw = synth.misshape_raw(synth.synth_raw_files(seed, T, w20, h20, False), d[0:2], d[2:4], d[4:6])
    random.seed(4)
    s2, dates, interp, s1, dem, cloudshad, snow = job.process_tile(dict(raw, clouds=None), sess)
    np.testing.assert_array_equal(dates, g[f"{tag}_dates"])
    shp = tuple(int(v) for v in g[f"{tag}_cloudshad_shape"])
    np.testing.assert_array_equal(cloudshad.cpu().numpy() > 0, np.unpackbits(g[f"{tag}_cloudshad"])[:np.prod(shp)].reshape(shp).astype(bool))
    np.testing.assert_array_equal(interp.cpu().numpy()[:, ::2, ::2], g[f"{tag}_interp_sub"])
    np.testing.assert_array_equal(dem.cpu().numpy(), g[f"{tag}_dem"])
    e1 = np.abs(s1.cpu().numpy() - g[f"{tag}_s1"]).max()
    s2n = s2.cpu().numpy()
    e2 = np.abs(s2n[:, ::3, ::3, :] - g[f"{tag}_s2_sub"])
    edges = np.concatenate([s2n[:, :2].reshape(s2n.shape[0], -1), s2n[:, -2:].reshape(s2n.shape[0], -1),
                            s2n[:, :, :2].reshape(s2n.shape[0], -1), s2n[:, :, -2:].reshape(s2n.shape[0], -1)], 1)
    e3 = np.abs(edges - g[f"{tag}_s2_edges"]).max()
    print(f"[parity] process_tile shapes {tag}: s1 max|d| = {e1:.2e} (whole array), s2 max|d| = {e2.max():.2e}, edge rows / columns {e3:.2e}")
    assert e1 < 2e-6 and e2.max() < 1e-5 and e2.mean() < 1e-7 and e3 < 1e-5


def test_adjust_shape_on_the_device_matches_reference():
    """ttc_adjust_shape against the reference's own adjust_shape outputs (adjust_shape.npz): identical where the rule reaches the requested
    size, TTC_ERR_ARG where it does not"""
    from ttc import _lib
    ctx = _lib.Context(win_in=44, length=1, max_windows=1)
    g = golden("adjust_shape.npz")
    n_ok = n_bad = 0
    for i in range(int(g["n"])):
        a, want = g[f"c{i}_in"], g[f"c{i}_out"]
        w, h = (int(v) for v in g[f"c{i}_want"])
        a4 = a[:, :, :, None] if a.ndim == 3 else (a[None, :, :, None] if a.ndim == 2 else a)
        if all(abs(n - t) in (0, 1) or abs(n - t) % 2 == 0 for n, t in ((a4.shape[1], w), (a4.shape[2], h))):
            np.testing.assert_array_equal(ctx.adjust_shape(a, w, h).cpu().numpy().squeeze(), want)
            n_ok += 1
        else:
            with pytest.raises(RuntimeError, match="adjust_shape"):
                ctx.adjust_shape(a, w, h)
            n_bad += 1
    assert n_ok >= 60 and n_bad >= 6
    ctx.close()
