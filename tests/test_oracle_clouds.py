"""Pin the CPU oracle of the multi-temporal cloud / shadow detector (oracle/restate_clouds.py) against
golden vectors captured by running the imported reference (tools/gen_golden.py: identify_clouds_shadows with the two
raster readers replaced by synthetic masks, or failing as they do without the rasters)."""
import numpy as np
import pytest

from oracle import restate_clouds as C
from tests.helpers import golden, synth


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_detection_matches_reference(tag):
    g = golden("cloud_detection.npz")
    seed, T, H, W, with_masks = (int(v) for v in g[f"{tag}_cfg"])
    img, dem, forest, core, near = synth.synth_detection_scene(seed, T, H, W)
    clouds, fcps = C.identify_clouds_shadows(img.copy(), dem.copy(), forest if with_masks else None,
                                             (core, near) if with_masks else None)
    want_c = np.unpackbits(g[f"{tag}_clouds"])[:T * H * W].reshape(T, H, W).astype(bool)
    want_f = np.unpackbits(g[f"{tag}_fcps"])[:T * H * W].reshape(T, H, W).astype(bool)
    assert float(clouds.max()) == float(g[f"{tag}_clouds_max"])
    np.testing.assert_array_equal(clouds > 0, want_c)
    np.testing.assert_array_equal(np.asarray(fcps, dtype=bool), want_f)
    assert 0.01 < want_c.mean() < 0.99
