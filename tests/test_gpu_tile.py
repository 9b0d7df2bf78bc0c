"""GPU parity of the per-tile numeric core (through the C ABI) against
 (a) golden vectors captured from the REFERENCE (tests/golden/, tools/gen_golden.py) and
 (b) the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

from tests.helpers import e2e_inputs, golden, synth

pytestmark = pytest.mark.gpu

_SESS = {}


def _session(W=172, L=4, seed=0, precision="fp32"):
    """one session per geometry for the whole module (workspace is several GB); "fp32+ds16" = an fp32 session whose DSen2 convs run on
    the 16-bit engine (ttc_config.dsen2_precision = fp16 pairs)"""
    from ttc import job, weights as Wt
    key = (W, L, seed, precision)
    if key not in _SESS:
        _SESS.clear()
        _SESS[key] = (job.TTCSession(Wt.synth_weights(seed), win_in=W, length=L, precision=precision.split("+")[0],
                                     dsen2_precision="fp16" if precision.endswith("+ds16") else None), Wt.synth_weights(seed))
    return _SESS[key]


def _report(name, got, ref, atol):
    err = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    print(f"[parity] {name:22s} max|d|={np.nanmax(err):.3e}  (tol {atol})")
    assert np.nanmax(err) <= atol, f"{name}: {np.nanmax(err)} > {atol}"


# --------------------------------------------------------------------------------------------
def test_missing_counts_and_repair():
    from oracle import restate_numpy as O
    sess, _ = _session(44, 4)
    s2, dates, interp, s1, dem = synth.synth_tile(seed=3, T=7, H=60, W=52)
    s2[1, 5:20, 3:9, :] = 0.0
    s2[4, 30:33, 10:40, 2] = 1.0
    s2[2, 0:50, :, :] = 0.0                     # a mostly-missing date
    s2[5, 10:12, 10:12, 3] = np.nan
    s2[6, 10:11, 10:13, 3] = np.nan
    ctx, t = sess.ctx, sess.ctx.torch
    d = t.from_numpy(s2.copy()).cuda()
    ctx.tile_fix_missing(d, do_nan=True, do_zero_one=False)
    ref = O.interpolate_na_vals(s2.copy())
    np.testing.assert_array_equal(d.cpu().numpy(), ref)
    counts = ctx.tile_missing_counts(d)
    m0 = np.sum(ref[..., :10] == 0.0, axis=-1) + np.sum(ref[..., :10] >= 1., axis=-1)
    np.testing.assert_array_equal(counts, np.sum(m0 > 1, axis=(1, 2)))
    ctx.tile_fix_missing(d, do_nan=False, do_zero_one=True)
    arr = ref.copy()
    for bad in (0, 1):                           # job.py:1039-1047 without the date screening
        for i in range(arr.shape[0]):
            a = arr[i]
            a[a == bad] = np.median(arr, axis=0)[a == bad]
    np.testing.assert_array_equal(d.cpu().numpy(), arr)


@pytest.mark.parametrize("L", [4, 12])
def test_small_tile_all_stages_vs_oracle(L):
    """100x100 tile, W=44 windows: temporal operator, medians, S1, window assembly, model, post."""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job
    W, size = 44, 30
    sess, w = _session(W, L)
    s2, dates, interp, s1, dem = synth.synth_tile(seed=21, T=9, H=100, W=100, cloud_frac=0.3)
    s2[2, 3:6, 4:9, :] = 0.0
    s2[5, 10:12, 1:3, 2] = 1.0
    s2[7, :60, :, :] = 0.0                       # dropped by id_missing_px
    net = M.TreeCoverNet(w, dtype=torch.float32)
    pf = lambda win: O.predict_subtile(win, net, size)
    ref_w, ref_in = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), pf, size=size,
                                       length=L, return_inputs=True)
    got, raw = job.process_subtiles(0, 0, s2.copy(), dates, interp, s1, dem, sess, size=size, return_raw=True)
    ctx = sess.ctx
    # temporal stage
    sm_ref, d2, i2 = O.smooth_large_tile(O.interpolate_na_vals(s2.copy()).astype(np.float32), dates.copy(), interp.copy())
    s2q, s1q = O.quarterly(sm_ref, s1, L)
    sm = ctx.debug_fetch("tile_sm", (L, 14, 100, 100))
    _report("smoothed series", sm, np.moveaxis(s2q, -1, 1), 5e-5)
    med = ctx.debug_fetch("tile_med", (14, 100, 100))
    _report("medians", med, np.moveaxis(O.tile_medians(O.interpolate_na_vals(s2.copy()).astype(np.float32)), -1, 0), 1e-6)
    _report("s1 series", ctx.debug_fetch("tile_s1q", (L, 2, 100, 100)), np.moveaxis(s1q, -1, 1), 0)
    _report("s1 median", ctx.debug_fetch("tile_s1med", (2, 100, 100)), np.moveaxis(np.median(s1, axis=0), -1, 0), 0)
    # model inputs (normalised windows) for every window the oracle fed
    fr = ctx.debug_fetch("frames", (36, L + 1, 17, W + 2, W + 2))[:, :, :, 1:-1, 1:-1]
    grid = job.window_grid(100, 100, size)
    worst = 0.0
    for i, (fx, fy) in enumerate(grid):
        if (fy, fx) in ref_in:
            worst = max(worst, np.abs(np.moveaxis(fr[i], 1, -1) - ref_in[(fy, fx)]).max())
    print(f"[parity] model inputs           max|d|={worst:.3e}")
    assert worst <= 2e-4
    assert set(got.keys()) == set(ref_w.keys())
    k = sorted(got.keys())
    g, r = np.stack([got[a] for a in k]), np.stack([ref_w[a] for a in k])
    assert np.array_equal(g > 1.0, r > 1.0), "no-data windows differ"
    _report("windows (rounded)", g, r, 1.1e-3)       # one np.around(.,3) quantum
    assert np.mean(np.abs(g - r) > 1e-6) < 0.02


@pytest.mark.parametrize("T", [1, 2, 3, 8, 17, 32])
def test_tile_date_count_range(T):
    """register-array template boundaries of the per-pixel kernels (T <= 8 / 16 / 32), and the < 2 dates rule"""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job
    W, size, L = 44, 30, 4
    sess, w = _session(W, L)
    s2, dates, interp, s1, dem = synth.synth_tile(seed=40 + T, T=T, H=100, W=100, cloud_frac=0.2)
    if T > 2:
        s2[1, 5:9, 7:11, :] = 0.0
        s2[T - 1, 20:22, 3:6, 4] = 1.0
    net = M.TreeCoverNet(w, dtype=torch.float32)
    pf = lambda win: O.predict_subtile(win, net, size)
    ref_w = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), pf, size=size, length=L)
    got = job.process_subtiles(0, 0, s2.copy(), dates, interp, s1, dem, sess, size=size)
    assert set(got.keys()) == set(ref_w.keys())
    k = sorted(got.keys())
    g, r = np.stack([got[a] for a in k]), np.stack([ref_w[a] for a in k])
    assert np.array_equal(g > 1.0, r > 1.0), "no-data windows differ"
    if T < 2:
        assert np.all(g == 255.0)                    # job.py:1418-1422
    _report(f"windows T={T}", g, r, 1.1e-3)
    assert np.mean(np.abs(g - r) > 1e-6) < 0.02


def test_full_tile_model_inputs_match_reference_feeds():
    """The tensors fed to the model for a 618^2 tile == what the REFERENCE fed its session."""
    from ttc import job
    sess, _ = _session(172, 4)
    for tag in ("e2e_clear", "e2e_cloudy"):
        g = golden(f"{tag}.npz")
        s2, dates, interp, s1, dem = e2e_inputs(g)
        got = job.process_subtiles(0, 0, s2, dates, interp, s1, dem, sess, size=158)
        fr = sess.ctx.debug_fetch("frames", (36, 5, 17, 174, 174))[:, :, :, 1:-1, 1:-1]
        keys = [tuple(k) for k in g["keys"]]
        refw = g["windows_permille"].astype(np.float64) / 1000.0
        grid = job.window_grid(618, 618, 158)
        fed = [i for i, (fx, fy) in enumerate(grid) if not np.all(refw[keys.index((fy, fx))] == 255.0)]
        assert len(fed) == int(g["n_feeds"])
        sub = np.stack([np.moveaxis(fr[i], 1, -1)[:, ::19, ::19, :] for i in fed])
        _report(f"{tag}: model inputs", sub, g["feeds_sub"], 2e-4)
        # no-data decisions (window-level and block mask) agree with the reference
        gw = np.stack([got[k] for k in keys])
        assert np.array_equal(gw > 1.0, refw > 1.0), f"{tag}: no-data mask differs from the reference"


@pytest.mark.parametrize("tag", ["e2e_clear", "e2e_cloudy", "mosaic"])
def test_mosaic_matches_reference(tag):
    from ttc import job
    sess, _ = _session(172, 4)
    g = golden(f"{tag}.npz")
    keys = [tuple(k) for k in g["keys"]]
    wins = {k: (g["windows_permille"][i] / 1000.0).astype(np.float32) for i, k in enumerate(keys)}
    mos = job.load_mosaic_predictions(wins, 1, sess=sess, size=158)
    ref = g["mosaic"]
    assert mos.shape == ref.shape and mos.dtype == np.uint8
    diff = np.abs(mos.astype(int) - ref.astype(int))
    print(f"[parity] mosaic {tag}: >1 count {np.mean(diff > 1):.2e}, ==1 count {np.mean(diff == 1):.2e}")
    assert np.array_equal(mos == 255, ref == 255)
    # exact-integer percentages (1 % of 3-decimal window values) truncate to n or n-1 depending on the float32
    # summation order, which in the reference follows os.listdir order: a 1-count tie band is inherent
    assert (diff > 1).mean() < 1e-5 and (diff > 0).mean() < 3e-2


def test_predict_tile_end_to_end_vs_oracle():
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job
    sess, w = _session(172, 4)
    g = golden("e2e_cloudy.npz")
    s2, dates, interp, s1, dem = e2e_inputs(g)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    ref_w = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(),
                               lambda win: O.predict_subtile(win, net, 158), size=158, length=4)
    ref_u8, ref_f = O.mosaic_predictions(ref_w, size=158, return_float=True)
    f32, u8 = job.predict_tile(s2, dates, interp, s1, dem, sess, size=158)
    assert np.array_equal(np.isnan(f32), np.isnan(ref_f))
    _report("tile percent raster", np.nan_to_num(f32), np.nan_to_num(ref_f), 0.11)   # 1e-3 prob quantum * 100
    d = np.abs(u8.astype(int) - ref_u8.astype(int))
    assert (d > 1).mean() < 1e-5 and (d > 0).mean() < 2e-2


def test_dsen2_and_superresolve_tile():
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import weights as Wt
    sess, _ = _session(172, 4)
    wd = Wt.load_dsen2()
    net = M.DSen2Lite(wd, dtype=torch.float32)
    rng = np.random.default_rng(5)
    x = rng.random((3, 118, 118, 10)).astype(np.float32)
    got = sess.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy()
    _report("DSen2 window", got, net(x, x[..., 4:]), 2e-5)
    # Session.run look-alike with the reference's tensor names
    from ttc import job
    via = sess.run([job.SUPERRESOLVE_LOGITS], feed_dict={job.SUPERRESOLVE_INP: x, job.SUPERRESOLVE_INP_BILINEAR: x[..., 4:]})[0]
    np.testing.assert_array_equal(via, got)
    arr = (rng.random((2, 618, 618, 10)) * 0.6).astype(np.float32)
    ref = O.superresolve_large_tile(arr.copy(), net)
    d = torch.from_numpy(arr.copy()).cuda()
    sess.ctx.superresolve_tile(d, quirks=True)
    out = d.cpu().numpy()
    _report("superresolve tile", out, ref, 5e-5)
    np.testing.assert_array_equal(out[..., :4], arr[..., :4])
    np.testing.assert_array_equal(out[:, :508, 550:, 4:], arr[:, :508, 550:, 4:])     # the never-refined strip
    d2 = torch.from_numpy(arr.copy()).cuda()
    sess.ctx.superresolve_tile(d2, quirks=False)
    assert not np.array_equal(d2.cpu().numpy()[:, :508, 550:, 4:], arr[:, :508, 550:, 4:])


def test_bf16_tile_end_to_end_vs_oracle():
    """precision = "bf16" (bf16 hi + lo pairs, three products, in both graphs): whole tile vs the fp32 oracle, within the
    1e-3 contract of BASELINE.json with margin (raw probabilities 2.5e-4, DSen2 reflectances 1e-4)."""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job, weights as Wt
    sess, w = _session(172, 4, precision="bf16")
    g = golden("e2e_cloudy.npz")
    s2, dates, interp, s1, dem = e2e_inputs(g)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    ref_w = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(),
                               lambda win: O.predict_subtile(win, net, 158), size=158, length=4)
    ref_u8, ref_f = O.mosaic_predictions(ref_w, size=158, return_float=True)
    f32, u8 = job.predict_tile(s2, dates, interp, s1, dem, sess, size=158)
    assert np.array_equal(np.isnan(f32), np.isnan(ref_f))
    _report("tile percent raster (bf16)", np.nan_to_num(f32), np.nan_to_num(ref_f), 0.11)
    d = np.abs(u8.astype(int) - ref_u8.astype(int))
    assert (d > 1).mean() < 1e-5 and (d > 0).mean() < 2e-2
    ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    rng = np.random.default_rng(5)
    x = rng.random((3, 118, 118, 10)).astype(np.float32)
    _report("DSen2 window (bf16)", sess.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy(), ds(x, x[..., 4:]), 1e-4)
    arr = (rng.random((2, 618, 618, 10)) * 0.6).astype(np.float32)
    ref = O.superresolve_large_tile(arr.copy(), ds)
    dd = torch.from_numpy(arr.copy()).cuda()
    sess.ctx.superresolve_tile(dd, quirks=True)
    _report("superresolve tile (bf16)", dd.cpu().numpy(), ref, 2e-4)


def test_feature_mosaic_matches_reference():
    """depth > 1 branch of load_mosaic_predictions (feature export) against the reference's own output"""
    from ttc import job
    sess, _ = _session(44, 2)
    g = golden("mosaic_features.npz")
    wins = {tuple(int(v) for v in k): g["windows"][i] for i, k in enumerate(g["keys"])}
    got = job.load_mosaic_predictions(wins, depth=16, sess=sess, size=30)
    assert got.shape == g["mosaic"].shape and got.dtype == np.int16
    d = np.abs(got.astype(int) - g["mosaic"].astype(int))
    print(f"[parity] feature mosaic: max|d| = {d.max()}, differing = {(d > 0).mean():.2e}")
    assert d.max() <= 1 and (d > 0).mean() < 1e-2


def test_dsen2_precision_option_in_fp32_session():
    """ttc_config.dsen2_precision: DSen2 on fp16 hi + lo pairs (three products) inside an fp32 session -- same oracle, the 16-bit engine's
    tolerances; the model itself still runs the fp32 engine (probabilities identical to the plain fp32 session's on the same feeds)"""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import weights as Wt
    rng = np.random.default_rng(5)
    x = rng.random((3, 118, 118, 10)).astype(np.float32)
    arr = (rng.random((2, 618, 618, 10)) * 0.6).astype(np.float32)
    wins = rng.random((2, 5, 172, 172, 17)).astype(np.float32)
    sess32, _ = _session(172, 4)
    plain = sess32.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy()
    p32 = sess32.ctx.forward_windows(torch.from_numpy(wins).cuda()).cpu().numpy()
    sess, _ = _session(172, 4, precision="fp32+ds16")
    net = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    got = sess.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy()
    _report("DSen2 window (fp32 session, fp16-pair DSen2)", got, net(x, x[..., 4:]), 5e-5)
    assert not np.array_equal(got, plain)                                    # the option took effect
    ref = O.superresolve_large_tile(arr.copy(), net)
    d = torch.from_numpy(arr.copy()).cuda()
    sess.ctx.superresolve_tile(d, quirks=True)
    _report("superresolve tile (fp32 session, fp16-pair DSen2)", d.cpu().numpy(), ref, 1e-4)
    np.testing.assert_array_equal(sess.ctx.forward_windows(torch.from_numpy(wins).cuda()).cpu().numpy(), p32)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("fp16", 5e-5), ("bf16", 1e-4), ("fp32+ds16", 5e-5)])
def test_dsen2_ragged_windows(precision, tol):
    """odd / tiny window sizes: planes whose size is not a multiple of 4 take the conv engines' unaligned staging path"""
    import torch
    from oracle import restate_model as M
    from ttc import weights as Wt
    sess, _ = _session(44, 2, precision=precision)
    net = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    rng = np.random.default_rng(11)
    for n, H, W in [(2, 11, 13), (1, 3, 3), (3, 17, 40), (1, 120, 7)]:
        x = rng.random((n, H, W, 10)).astype(np.float32)
        got = sess.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy()
        _report(f"DSen2 {n}x{H}x{W} ({precision})", got, net(x, x[..., 4:]), tol)


@pytest.mark.parametrize("h,w", [(20, 18), (21, 19), (21, 18), (20, 19)])
def test_upsample_20m(h, w):
    """even grids and the three odd-grid branches of job.py:760-782 (309 x 309 is the production case)"""
    from oracle import restate_numpy as O
    sess, _ = _session(172, 4)
    rng = np.random.default_rng(8)
    s10 = rng.random((2, 2 * h, 2 * w, 4)).astype(np.float32)
    s20 = rng.random((2, h, w, 6)).astype(np.float32)
    got = sess.ctx.upsample_20m(s10, s20).cpu().numpy()
    _report(f"bilinear 20m->10m {h}x{w}", got, O.upsample_20m(s10, s20), 1e-6)


@pytest.mark.parametrize("seed,dates_kind", [(1234, "regular"), (77, "wrap")])
def test_single_call_tile_matches_staged_calls(seed, dates_kind):
    """ttc_predict_tile (one enqueue, temporal operator and date screening formed on the device, speculative gap-fill) against
    the staged calls job.py chains for the same raw tile: identical model inputs (the device-built 12 x T operator vs the host
    mirror temporal.py) and identical rasters; the status word reports no fallback on these tiles.  'wrap': dates that start
    late and end early in the year exercise the year-end wrap-around / mirroring branches of the regrid."""
    import torch
    from ttc import job
    sess, _ = _session(172, 4)
    ctx = sess.ctx
    T, X = 9, 618
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=seed, T=T, H=X, W=X)
    _, _, _, s1, dem = synth.synth_tile(seed=seed, T=2, H=X, W=X)
    if dates_kind == "wrap":
        dates = np.array([100, 118, 131, 150, 178, 200, 221, 240, 262])
    else:
        dates = np.asarray(dates)
    s2[1, 100:400, :, :] = 0.0                      # > 10 % missing pixels on a cloud-free date: the device screening must drop it

    def u16(a):
        return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)
    s2_10, s2_20, s1u = u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), u16(s1)
    # staged path (what bench.py / job.py drive from Python)
    d10 = torch.from_numpy(s2_10.view(np.int16)).cuda(); d20 = torch.from_numpy(s2_20.view(np.int16)).cuda()
    f10, f20, s1db = ctx.to_float32(d10), ctx.to_float32(d20), ctx.s1_to_db(torch.from_numpy(s1u.view(np.int16)).cuda())
    s2d = ctx.upsample_20m(f10, f20)
    dint, _, _ = ctx.remove_cloud_and_shadows(s2d, probs, None, None)
    ctx.superresolve_tile(s2d, quirks=True)
    f32_ref, u8_ref = job.predict_tile(s2d, dates, dint, s1db, dem, sess, size=158)
    frames_ref = ctx.debug_fetch("frames", (36, 5, 17, 174, 174))
    # single call
    u8, f32, frames, status = ctx.predict_tile_raw(s2_10, s2_20, s1u, dem, probs, dates, job.min_all, job.max_all, 158,
                                                   want_float=True, want_inputs=True)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    print(f"[parity] single-call tile {dates_kind}: status {st.tolist()}")
    assert st[0] == 0 and st[2] == 0 and st[1] == T - 1
    d_in = np.abs(frames.cpu().numpy() - frames_ref).max()
    print(f"[parity] single-call tile {dates_kind}: model inputs max|d| = {d_in:.2e}")
    assert d_in <= 2e-6
    got_f = f32.cpu().numpy()
    assert np.array_equal(np.isnan(got_f), np.isnan(f32_ref))
    _report(f"single-call raster ({dates_kind})", np.nan_to_num(got_f), np.nan_to_num(f32_ref), 0.11)
    d = np.abs(u8.cpu().numpy().astype(int) - u8_ref.astype(int))
    assert (d > 1).mean() < 1e-5 and (d > 0).mean() < 2e-2
    # preprocessing only (BASELINE configs[2]): the same model inputs, no model, no raster
    _, _, frames2, _ = ctx.predict_tile_raw(s2_10, s2_20, s1u, dem, probs, dates, job.min_all, job.max_all, 158,
                                            flags=ctx.TILE_INPUTS_ONLY, want_inputs=True)
    np.testing.assert_array_equal(frames2.cpu().numpy(), frames.cpu().numpy())


def test_device_temporal_operator_matches_host_mirror():
    """k_build_wmat (device) against sentinel-tree-cover_amd/temporal.py on date sets that exercise every branch of the regrid
    (utils.py:176-347): regular, late start / early end (wrap-around), mirrored sides, negative days, a duplicate pair (the
    reference raises -> all-zero operator), through the model inputs of a tiny constant-per-date tile."""
    import torch
    from ttc import job, temporal
    sess, _ = _session(44, 4)
    ctx = sess.ctx
    X = 100
    rng = np.random.default_rng(3)
    for dates in ([5, 40, 100, 160, 220, 280, 340], [-20, 12, 33, 95, 130, 171, 200, 244, 290, 301, 350, 380],
                  [100, 130, 160, 200], [200, 230, 260, 300, 330], [10, 25, 40, 80], [15, 15, 60, 200], [-150, -120, 30, 90, 200]):
        T = len(dates)
        levels = rng.uniform(0.05, 0.45, (T, 1, 1, 10)).astype(np.float32)
        s2 = np.broadcast_to(levels, (T, X, X, 10)).copy()
        s2 += rng.normal(0, 0.002, s2.shape).astype(np.float32)
        s2 = np.clip(s2, 0.01, 0.9)
        s2_10 = np.trunc(s2[..., :4] * 65535).astype(np.uint16)
        s2_20 = np.trunc(s2[:, ::2, ::2, 4:] * 65535).astype(np.uint16)
        s1 = np.full((12, X, X, 2), 20000, np.uint16)
        dem = np.zeros((X, X), np.float32)
        mask = np.zeros((T, X, X), np.float32)
        _, _, frames, status = ctx.predict_tile_raw(s2_10, s2_20, s1, dem, mask, dates, job.min_all, job.max_all, 30,
                                                    flags=ctx.TILE_INPUTS_ONLY | ctx.TILE_NO_SUPERRES, want_inputs=True)
        # staged: same arrays through the host mirror
        d10 = torch.from_numpy(s2_10.view(np.int16)).cuda(); d20 = torch.from_numpy(s2_20.view(np.int16)).cuda()
        s2d = ctx.upsample_20m(ctx.to_float32(d10), ctx.to_float32(d20))
        dint, _, _ = ctx.remove_cloud_and_shadows(s2d, mask, None, None)
        s1db = ctx.s1_to_db(torch.from_numpy(s1.view(np.int16)).cuda())
        job._process_subtiles_device(s2d, np.asarray(dates), dint, s1db, dem, sess, 30)
        ref = ctx.debug_fetch("frames", (36, 5, 17, 46, 46))
        d = np.abs(frames.cpu().numpy() - ref).max()
        W = temporal.temporal_operator(np.asarray(dates))
        print(f"[parity] device operator, dates {dates}: model inputs max|d| = {d:.2e} (host operator {'zero' if not W.any() else 'ok'})")
        assert d <= 2e-6


def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 end to end on ONE device: two processes, gloo-staged collectives (TTC_BENCH_BACKEND / TTC_BENCH_DEVICE),
    so the multi-rank control flow -- rendezvous, distinct tile ids per rank, the batched raster gather on the side stream,
    max-over-ranks timing, rank-0-only JSON -- runs before the driver's 8-GPU job ever does"""
    import json
    import subprocess
    import sys
    from tests.helpers import ROOT
    env = dict(os.environ, TTC_BENCH_BACKEND="gloo", TTC_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inflight", "2", "--steps", "2", "--warmup", "1", "--pool", "2",
           "--gather-batch", "2", "--no-cpu-baseline", "--no-dprob", "--no-alt"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert any("rccl_gather_u8" in st for st in out["config"]["stages"])
    assert out["config"]["gathers_timed"] == 2 and out["config"]["tiles_failed"] == 0
    assert out["value"] > 0 and abs(out["value"] - 2 * 2 * 618 * 618 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6
