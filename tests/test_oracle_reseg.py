"""The CPU restatement of the border resegmentation (oracle/restate_reseg.py) against golden vectors captured from the
reference's own functions (tools/gen_golden_reseg.py; src/resegment_tiles_wide.py)."""
import numpy as np
import pytest

from oracle import restate_reseg as RS
from tests.helpers import golden, fake_model, synth_border_strip, synth_reseg_windows


def test_align_dates_and_window_table():
    g = golden("reseg_small.npz")
    for i in range(4):
        ra, rb, left = RS.align_dates(g[f"dates{i}_a"], g[f"dates{i}_b"])
        np.testing.assert_array_equal(np.array(ra, dtype=np.int64), g[f"dates{i}_rm_a"])
        np.testing.assert_array_equal(np.array(rb, dtype=np.int64), g[f"dates{i}_rm_b"])
        assert int(left) == int(g[f"dates{i}_left"])
    for tag in ("real", "small", "odd"):
        n_rows, size, size_y = (int(v) for v in g[f"table_{tag}_cfg"])
        ta, tf = RS.border_window_table(n_rows, size, size_y, tiles_folder_x=n_rows - size // 2)
        np.testing.assert_array_equal(ta, g[f"table_{tag}_array"])
        np.testing.assert_array_equal(tf, g[f"table_{tag}_folder"])


def artifact_cases():
    rng = np.random.default_rng(5)
    for i in range(12):
        base = np.clip(50 + 30 * np.sin(np.arange(618) / (20 + 3 * i))[:, None] + rng.normal(0, 4, (618, 20)), 0, 100).astype(np.float32)
        nb = base[:, ::-1] + np.float32([0, 0.5, 2, 5, 7, 14, 25][i % 7]) * (1 if i < 7 else np.sign(np.sin(np.arange(618) / 15.0))[:, None])
        nb = np.clip(nb, 0, 100).astype(np.float32)
        if i % 3 == 0:
            base[100:140, -4:] = np.nan
            nb[300:320, :2] = np.nan
        yield base, nb


def test_check_if_artifact():
    g = golden("reseg_small.npz")
    got = [RS.check_if_artifact(a.copy(), b.copy()) for a, b in artifact_cases()]
    np.testing.assert_array_equal(got, g["artifact_flags"])
    assert 0 < sum(got) < len(got)


def hist_input(g, tag):
    seed, X, W = (int(v) for v in g[f"{tag}_cfg"])
    s2 = synth_border_strip(seed, X, W, offset=float(g[f"{tag}_off"]))[0]
    return np.median(np.reshape(np.nan_to_num(s2), (4, 3) + s2.shape[1:]), axis=1)


@pytest.mark.parametrize("tag", ["h0", "h1", "h2"])
def test_align_subtile_histograms(tag):
    g = golden("reseg_small.npz")
    arr = hist_input(g, tag)
    out = RS.align_subtile_histograms(arr.copy(), size=90)
    np.testing.assert_array_equal([not np.array_equal(out[t], arr[t]) for t in range(4)], g[f"{tag}_changed"])
    np.testing.assert_allclose(out[:, ::3, ::4, :], g[f"{tag}_out"], rtol=0, atol=1e-6)


def subtile_inputs(g, tag):
    seed, X, size, size_y, align = (int(v) for v in g[f"{tag}_cfg"])
    off = float(g[f"{tag}_off"])
    s2, dates, interp, s1, dem, left_all, right_all, min_clear = synth_border_strip(seed, X, size + 14, offset=off)
    if tag == "c":
        s1[...] = 0; s2[...] = 0; dem[...] = 0
        s2[:, 150:, :, :] = synth_border_strip(seed, X, size + 14, offset=off)[0][:, 150:]
    ta, tf = RS.border_window_table(X, size, size_y, tiles_folder_x=X + 9 - size // 2)
    return dict(s2=s2, dates=dates, interp=interp, s1=s1, dem=dem, left_all=left_all, right_all=right_all, min_clear=min_clear,
                tiles_array=ta, tiles_folder=tf, size=size, size_y=size_y, hist_align=bool(align))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_border_process_subtiles(tag):
    g = golden("reseg_subtiles.npz")
    k = subtile_inputs(g, tag)
    trace = {}
    out = RS.process_border_subtiles(k["s2"].copy(), k["dates"], k["interp"], k["s1"].copy(), k["dem"], fake_model, k["tiles_folder"],
                                     k["tiles_array"], k["right_all"], k["left_all"], k["hist_align"], k["min_clear"],
                                     size=k["size"], size_y=k["size_y"], trace=trace)
    feeds = [trace[t] for t in sorted(trace)]
    assert len(feeds) == int(g[f"{tag}_n_feeds"])
    for i, f in enumerate(feeds):
        np.testing.assert_allclose(f[:, ::5, ::7, :], g[f"{tag}_feed{i}"], rtol=0, atol=2e-6)
        assert abs(f.astype(np.float64).sum() - float(g[f"{tag}_feed{i}_sum"])) < 1e-6 * f.size
    for t, o in enumerate(out):
        assert o["saved"] == bool(g[f"{tag}_saved{t}"])
        if o["saved"]:
            assert [str(o["folder_y"]), str(o["folder_x"])] == list(g[f"{tag}_name{t}"])
            assert o["preds"].shape == g[f"{tag}_preds{t}"].shape
            np.testing.assert_allclose(np.asarray(o["preds"], dtype=np.float32), g[f"{tag}_preds{t}"], rtol=0, atol=2e-6)


def ordered_windows(g, tag):
    seed, Y, X, size, size_y, ud = (int(v) for v in g[f"{tag}_cfg"])
    wins = {(k, x, y): p for k, x, y, p in synth_reseg_windows(seed, (Y, X), size, size_y, bool(ud))}
    order = [("nlrud"[k], x, y) for k, x, y in g[f"{tag}_order"]]
    assert sorted(order) == sorted(wins)
    return [(k, x, y, wins[(k, x, y)]) for k, x, y in order], (Y, X), size


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_recreate_resegmented(tag):
    g = golden("reseg_mosaic.npz")
    wins, shape, size = ordered_windows(g, tag)
    preds, sums = RS.recreate_resegmented(wins, shape, size=size)
    np.testing.assert_allclose(sums[::3, ::3], g[f"{tag}_sums_sub"], rtol=1e-5, atol=1e-7)
    if f"{tag}_preds" in g:
        np.testing.assert_allclose(preds, g[f"{tag}_preds"], rtol=0, atol=2e-4)
    else:
        np.testing.assert_array_equal(np.packbits(preds == 255), g[f"{tag}_nodata"])
        np.testing.assert_allclose(preds[::2, ::2], g[f"{tag}_preds_sub"], rtol=0, atol=2e-4)


def border_case(g, tag):
    from tests.helpers import synth_border_pair
    seed, T, X, Y, size, size_y, same = (int(v) for v in g[f"{tag}_cfg"])
    tile, neighb, tif_t, tif_n = synth_border_pair(seed, T, X, Y, bool(same))
    tt, tn = tif_t.astype(np.float32), tif_n.astype(np.float32)
    tt[tt > 100] = np.nan
    tn[tn > 100] = np.nan
    return tile, neighb, tt, tn, size, size_y


@pytest.mark.parametrize("tag", ["s", "d"])
def test_resegment_border_arrays(tag):
    """resegment_border (:847-1161): shared preprocessing of the strip ("s", dates agree) and per-tile preprocessing with
    histogram alignment ("d"), against the arguments the reference hands to process_subtiles and the windows it saves"""
    import random
    from tests.helpers import fake_dsen2
    g = golden("reseg_border.npz")
    tile, neighb, tt, tn, size, size_y = border_case(g, tag)
    random.seed(11)
    trace = {}
    wins, info = RS.resegment_border_arrays(tile, neighb, tt, tn, fake_model, fake_dsen2, min_dates=2, size=size, size_y=size_y, trace=trace)
    assert info["min_images"] == int(g[f"{tag}_result"][1]) and info["hist_align"] == bool(g[f"{tag}_hist_align"])
    np.testing.assert_array_equal(trace["dates"], g[f"{tag}_dates"])
    np.testing.assert_array_equal(info["tiles_array"], g[f"{tag}_ta"])
    np.testing.assert_array_equal(info["tiles_folder"], g[f"{tag}_tf"])
    np.testing.assert_array_equal(trace["min_clear"], g[f"{tag}_min_clear"])
    np.testing.assert_allclose(trace["interp"][:, ::4, ::4], g[f"{tag}_interp_sub"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(trace["strip"][:, ::9, ::5, :], g[f"{tag}_strip_sub"], rtol=0, atol=1e-4)
    assert abs(trace["strip"].astype(np.float64).sum() - float(g[f"{tag}_strip_sum"])) < 2e-6 * trace["strip"].size
    for t, o in enumerate(wins):
        assert o["saved"] == bool(g[f"{tag}_saved{t}"])
        if o["saved"]:
            np.testing.assert_allclose(np.asarray(o["preds"], dtype=np.float32), g[f"{tag}_preds{t}"], rtol=0, atol=2e-4)


def test_match_s1_steps_like_the_reference():
    """resegment_tiles_wide.py:1071-1097: index lists for 12 / 6 / 4-step Sentinel-1 stacks (host logic of the product mirror)"""
    import ttc  # noqa: F401
    from ttc import resegment as RG
    mk = lambda n: np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1) * np.ones((1, 2, 3, 2), np.float32)      # noqa: E731
    for n, m, want in [(12, 6, [0, 2, 4, 6, 8, 10]), (12, 4, [0, 3, 6, 9]), (6, 4, [0, 1, 3, 5])]:
        a, b = RG.match_s1_steps(mk(n), mk(m))
        assert a.shape[0] == b.shape[0] == m and a[:, 0, 0, 0].tolist() == [float(v) for v in want]
        a, b = RG.match_s1_steps(mk(m), mk(n))
        assert a.shape[0] == b.shape[0] == m and b[:, 0, 0, 0].tolist() == [float(v) for v in want]
    a, b = RG.match_s1_steps(mk(12), mk(12))
    assert a.shape[0] == 12 and b.shape[0] == 12
