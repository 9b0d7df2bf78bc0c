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
