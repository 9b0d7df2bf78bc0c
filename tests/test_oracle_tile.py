"""Pin oracle/restate_tile.py (process_tile's numeric flow from raw arrays) against the reference's process_tile run with its
file loader replaced (tools/gen_golden.py -> tests/golden/process_tile.npz)."""
import random

import numpy as np
import pytest

from oracle import restate_tile as P
from tests.helpers import golden, synth


@pytest.mark.parametrize("tag", ["a", "b"])
def test_process_tile_matches_reference(tag):
    g = golden("process_tile.npz")
    seed, T, w20, h20, with_clm = (int(v) for v in g[f"{tag}_cfg"])
    raw = synth.synth_raw_files(seed, T, w20, h20, bool(with_clm))
    random.seed(4)
    s2, dates, interp, s1, dem, cloudshad, snow = P.process_tile_arrays(raw)
    np.testing.assert_array_equal(dates, g[f"{tag}_dates"])
    assert len(dates) < T                                       # the scene makes the function drop dates and re-detect
    shp = tuple(int(v) for v in g[f"{tag}_cloudshad_shape"])
    want_cs = np.unpackbits(g[f"{tag}_cloudshad"])[:np.prod(shp)].reshape(shp).astype(bool)
    np.testing.assert_array_equal(cloudshad > 0, want_cs)
    np.testing.assert_array_equal(interp[:, ::2, ::2], g[f"{tag}_interp_sub"])
    # the fixture comes from the reference run with the REAL scikit-image 0.18.3, whose bilinear warp of float32 bands works in
    # float32; the oracle restates the float64 evaluation (what scikit-image >= 0.19 does): one float32 ulp (measured 8.9e-8)
    np.testing.assert_allclose(s2[:, ::3, ::3, :], g[f"{tag}_s2_sub"], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(s1[:, ::4, ::4, :], g[f"{tag}_s1_sub"])
    np.testing.assert_array_equal(dem.astype(np.float32), g[f"{tag}_dem"])
    want_snow = np.unpackbits(g[f"{tag}_snow"])[:snow.size].reshape(snow.shape).astype(bool)
    np.testing.assert_array_equal(np.asarray(snow) > 0, want_snow)
