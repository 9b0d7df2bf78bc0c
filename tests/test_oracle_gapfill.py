"""Pin the gap-fill oracle (oracle/restate_gapfill.py) against golden vectors captured by running the
reference's cloud_removal.py with random.seed fixed (tools/gen_golden.py).  CPU only."""
import random

import numpy as np

from oracle import restate_gapfill as G
from tests.helpers import golden, synth


def test_gapfill_matches_reference_bit_exactly():
    g = golden("gapfill.npz")
    tiles, dates, probs, pf = synth.synth_gapfill_scene(int(g["seed"]), int(g["T"]), int(g["H"]), int(g["W"]))
    np.testing.assert_array_equal(G.id_areas_to_interp(probs.copy()), g["id_areas"])
    random.seed(int(g["rng_seed"]))
    out, interp, rem, mosaic = G.remove_cloud_and_shadows(tiles.copy(), probs.copy(), pf, return_mosaic=True)
    np.testing.assert_array_equal(interp, g["interp"])
    np.testing.assert_array_equal(mosaic[::2, ::2], g["mosaic_sub"])
    np.testing.assert_array_equal(out[:, ::3, ::3, :], g["tiles_sub"])
    assert abs(out.astype(np.float64).sum() - float(g["tiles_sum"])) < 1e-6
    assert list(rem) == list(g["to_remove"])
    # the stage does something: cloudy pixels moved, clear pixels untouched
    cloudy = interp > 0
    assert np.abs(out - tiles)[cloudy].max() > 0.1
    np.testing.assert_array_equal(out[~(g["interp"] > 0)], tiles[~(g["interp"] > 0)])


def test_closing_window_convention():
    """scipy grey_closing(size=20): dilation window [-9, +10], erosion window [-10, +9], 'reflect' border."""
    rng = np.random.default_rng(0)
    a = rng.random((40, 37))

    def filt(a, axis, lo, hi, fn):
        n = a.shape[axis]
        out = None
        for k in range(lo, hi + 1):
            j = np.arange(n) + k
            j = np.where(j < 0, -j - 1, j)
            j = np.where(j >= n, 2 * n - 1 - j, j)
            v = np.take(a, j, axis=axis)
            out = v if out is None else fn(out, v)
        return out
    d = filt(filt(a, 0, -9, 10, np.maximum), 1, -9, 10, np.maximum)
    e = filt(filt(d, 0, -10, 9, np.minimum), 1, -10, 9, np.minimum)
    from scipy import ndimage as ndi
    np.testing.assert_array_equal(e, ndi.grey_closing(a, size=20))


def test_expected_sampler_is_the_mean_of_the_reference_sampler():
    """sampler = "expected" (restate_gapfill.expected_weights, the product's default): the per-row weight equals the mean
    multiplicity of the reference's stratified draw (CR.py:453-500) over many seeded draws, and the gap-filled stack it
    produces lies inside the spread of individual draws."""
    rng = np.random.default_rng(4)
    evi = np.clip(rng.normal(0.3, 0.4, 4000), -1.5, 1.5).astype(np.float32)
    n = evi.size
    w = G.expected_weights(evi, n)
    n_i = n // 5
    # five quintile strata of ~n/5 rows each survive a cut to n_i almost entirely; the 2 % tails add 10
    assert np.all((w > 0.99) & (w <= 11.0)) and abs(int((w > 10).sum()) - int(round(0.04 * n))) <= 2
    counts = np.zeros(n)
    r = random.Random(3)
    draws = 200
    for _ in range(draws):
        idx = np.concatenate([G.reference_sampler(evi, 10 ** 9, rng=r)])        # no final truncation: the raw multiset
        counts += np.bincount(idx, minlength=n)
    mean = counts / draws
    assert np.abs(mean - w).max() < 0.25 and abs(mean.sum() - w.sum()) / w.sum() < 5e-3
    # strata larger than the cut (n > 90000): survival probability n_i / c < 1, and the final sample[:n_rows] cut scales
    # every expectation by the same factor n_rows / len(sample)
    evi = np.clip(rng.normal(0.3, 0.4, 120000), -1.5, 1.5).astype(np.float32)
    n = evi.size
    w = G.expected_weights(evi, n)
    assert abs(np.median(w) - 18000 / 24000) < 0.01
    counts = np.zeros(n)
    for _ in range(24):
        counts += np.bincount(G.reference_sampler(evi, n, rng=r), minlength=n)
    mean = counts / 24
    scale = mean.sum() / w.sum()
    assert abs(scale - n / float(w.sum())) < 1e-6 and abs(scale - 120000 / 138000) < 0.01
    for lo, hi in ((0.0, 0.9), (9.0, 12.0)):            # body rows and x10 tail rows, each against its own expectation
        m = (w > lo) & (w < hi)
        assert abs(mean[m].mean() / (scale * w[m].mean()) - 1.0) < 0.02
    tiles, dates, probs, pf = synth.synth_gapfill_scene(70, 6, 96, 88)
    det, di, _ = G.remove_cloud_and_shadows(tiles.copy(), probs.copy(), pf.copy(), sampler="expected")
    outs = []
    for seed in (1, 2, 3):
        random.seed(seed)
        o, oi, _ = G.remove_cloud_and_shadows(tiles.copy(), probs.copy(), pf.copy())
        np.testing.assert_array_equal(oi, di)
        outs.append(o)
    spread = max(np.abs(outs[0] - outs[1]).max(), np.abs(outs[0] - outs[2]).max(), np.abs(outs[1] - outs[2]).max())
    dev = max(np.abs(det - o).max() for o in outs)
    print(f"[oracle] expected vs seeded draws: max|d| {dev:.2e}; draw-to-draw spread {spread:.2e}")
    assert dev < 2.0 * spread + 1e-4
