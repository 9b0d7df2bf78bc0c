"""write_tif (src/downloading/io.py:229-263) without rasterio: the LZW GeoTIFF the host side of libttc writes is read back
with Pillow (libtiff).  rasterio / GDAL are absent here, so byte-for-byte parity with GDAL's encoder is not claimed: any
conforming reader must see the same pixels and the same georeferencing."""
import numpy as np
import pytest

import ttc  # noqa: F401
from ttc import _lib, job


def _read(path):
    from PIL import Image
    im = Image.open(path)
    return np.array(im), {k: im.tag_v2[k] for k in im.tag_v2}


@pytest.mark.parametrize("shape,kind", [((618, 618), "raster"), ((7, 3), "tiny"), ((300, 517), "noise"), ((64, 4100), "flat")])
def test_geotiff_roundtrip(tmp_path, shape, kind):
    rng = np.random.default_rng(4)
    if kind == "raster":      # smooth 0-100 field with a 255 no-data block, like the product
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        a = np.clip(50 + 40 * np.sin(yy / 37.0) * np.cos(xx / 29.0) + rng.normal(0, 3, shape), 0, 100).astype(np.uint8)
        a[100:180, 200:420] = 255
    elif kind == "noise":     # incompressible: the 12-bit dictionary fills up and is cleared many times
        a = rng.integers(0, 256, shape, dtype=np.uint8)
    elif kind == "flat":      # long runs: the code width grows through 9..12 bits on few symbols
        a = np.full(shape, 17, np.uint8)
    else:
        a = rng.integers(0, 256, shape, dtype=np.uint8)
    path = tmp_path / "t.tif"
    _lib.write_geotiff_u8(path, a, west=-1.25, south=5.0, east=-1.19, north=5.06)
    got, tags = _read(path)
    np.testing.assert_array_equal(got, a)
    assert tags[259] == 5 and tags[258] == (8,) and tags[277] == 1                 # LZW, 8 bits, 1 sample
    np.testing.assert_allclose(tags[33550], ((-1.19 + 1.25) / shape[1], (5.06 - 5.0) / shape[0], 0.0), rtol=1e-15)
    np.testing.assert_allclose(tags[33922], (0, 0, 0, -1.25, 5.06, 0), rtol=0, atol=0)
    assert tuple(tags[34735]) == (1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326)
    if kind in ("raster", "flat"):
        assert path.stat().st_size < a.size


def test_write_tif_mirror(tmp_path):
    """same file name, transposition and bounds convention as the reference's write_tif"""
    arr = (np.arange(20 * 30).reshape(20, 30) % 101).astype(np.float32)
    out = job.write_tif(arr, [10.0, -3.0, 10.06, -2.94], 1234, 567, str(tmp_path) + "/", "_SMOOTH_X")
    assert out.endswith("1234X567Y_SMOOTH_X.tif")
    got, tags = _read(out)
    np.testing.assert_array_equal(got, arr.T.astype(np.uint8))
    np.testing.assert_allclose(tags[33922], (0, 0, 0, 10.0, -2.94, 0))
    with pytest.raises(RuntimeError):
        _lib.write_geotiff_u8(tmp_path / "x.tif", np.zeros((4, 4), np.uint8), 1.0, 0.0, 0.5, 1.0)      # east < west
