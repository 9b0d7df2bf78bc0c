"""CPU: the hickle / HDF5 reader of the C ABI (ttc_read_hkl, csrc/hickle.hip -- hkl.load of src/download_and_predict_job.py:684-714)
against files laid out byte by byte by tools/write_hdf5_fixture.py (tests/golden/hkl/).  Reader and writer are independent
implementations of the HDF5 file-format specification; NO real hickle file exists in the checkout or can be produced here
(no h5py / hickle / libhdf5), so parity with real hickle output is UNPINNED (stated in DESIGN.md)."""
import importlib.util
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, ROOT
import ttc  # noqa: F401
from ttc import _lib

spec = importlib.util.spec_from_file_location("write_hdf5_fixture", os.path.join(ROOT, "tools", "write_hdf5_fixture.py"))
WF = importlib.util.module_from_spec(spec)
spec.loader.exec_module(WF)


@pytest.mark.parametrize("fname", sorted(WF.fixtures().keys()))
def test_reader_returns_the_arrays(fname):
    dname, arr, how = WF.fixtures()[fname]
    got = _lib.read_hkl(os.path.join(GOLDEN, "hkl", fname + ".hkl"))            # hickle's default dataset lookup
    assert got.dtype == arr.dtype and got.shape == arr.shape
    np.testing.assert_array_equal(got, arr)
    np.testing.assert_array_equal(_lib.read_hkl(os.path.join(GOLDEN, "hkl", fname + ".hkl"), dname), arr)


def test_fixture_files_are_what_the_writer_produces(tmp_path):
    """the committed files are deterministic outputs of the committed writer"""
    for fname, (dname, arr, how) in WF.fixtures().items():
        w = WF.Writer()
        oh = w.chunked_dataset(arr, **how) if isinstance(how, dict) else w.contiguous_dataset(arr, continuation=(how == "continuation"))
        links = {dname: oh}
        if fname == "clouds_f32":
            links["aux"] = w.contiguous_dataset(np.arange(4, dtype=np.int32))
        p = tmp_path / (fname + ".hkl")
        w.finish(links, str(p))
        assert p.read_bytes() == open(os.path.join(GOLDEN, "hkl", fname + ".hkl"), "rb").read()


def test_codec_round_trip_of_a_raw_band_file():
    """the job's use: u16 array from the file -> to_float32 (tof_downloading.py:64-72) is x / 65535"""
    from oracle import restate_numpy as O
    arr = _lib.read_hkl(os.path.join(GOLDEN, "hkl", "s2_10_u16.hkl"))
    f = O.to_float32(arr)
    assert f.dtype == np.float32 and f.max() <= 1.0 and np.array_equal(O.to_int16(f), arr)


def test_errors_are_loud(tmp_path):
    with pytest.raises(RuntimeError, match="cannot open"):
        _lib.read_hkl(str(tmp_path / "missing.hkl"))
    bad = tmp_path / "bad.hkl"
    bad.write_bytes(b"not an hdf5 file" * 10)
    with pytest.raises(RuntimeError, match="not an HDF5 file"):
        _lib.read_hkl(str(bad))
    with pytest.raises(RuntimeError, match="dataset not found"):
        _lib.read_hkl(os.path.join(GOLDEN, "hkl", "dates_i64.hkl"), "nope")
    trunc = tmp_path / "trunc.hkl"
    trunc.write_bytes(open(os.path.join(GOLDEN, "hkl", "s2_10_u16.hkl"), "rb").read()[:20000])
    with pytest.raises(RuntimeError):
        _lib.read_hkl(str(trunc))


def test_malformed_chunk_layout_is_rejected(tmp_path):
    """a chunk dimension of 0 (the copy loop would never advance) and a chunk dimensionality that does not match the dataspace
    rank (wrong B-tree key size) are errors, not hangs / out-of-bounds reads"""
    import struct
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 65535, (3, 10, 12, 4)).astype(np.uint16)
    w = WF.Writer()
    w.finish({"data": w.chunked_dataset(arr, chunks=(1, 5, 7, 4))}, str(tmp_path / "ok.hkl"))
    blob = bytearray((tmp_path / "ok.hkl").read_bytes())
    np.testing.assert_array_equal(_lib.read_hkl(str(tmp_path / "ok.hkl")), arr)
    dims = struct.pack("<5I", 1, 5, 7, 4, 2)                      # layout message: chunk dims + element size
    at = bytes(blob).find(dims)
    assert at > 0 and bytes(blob).find(dims, at + 1) < 0
    zero = bytearray(blob); zero[at + 4:at + 8] = struct.pack("<I", 0)
    (tmp_path / "zero.hkl").write_bytes(bytes(zero))
    with pytest.raises(RuntimeError, match="chunk dimension of size 0"):
        _lib.read_hkl(str(tmp_path / "zero.hkl"))
    rank = bytearray(blob); rank[at - 9] = 4                      # dimensionality byte of the layout message (rank + 1 = 5)
    assert blob[at - 9] == 5
    (tmp_path / "rank.hkl").write_bytes(bytes(rank))
    with pytest.raises(RuntimeError, match="does not match the dataspace rank"):
        _lib.read_hkl(str(tmp_path / "rank.hkl"))


def test_load_raw_tile_reads_the_raw_folder(tmp_path):
    """job.load_raw_tile: the file names of job.py:669-683, each through the HDF5 reader"""
    from ttc import job
    rng = np.random.default_rng(1)
    arrays = {"raw/clouds/clouds_12X34Y.hkl": rng.random((3, 8, 8)).astype(np.float32),
              "raw/clouds/cloudmask_12X34Y.hkl": rng.integers(0, 2, (3, 4, 4)).astype(np.float32),
              "raw/s1/12X34Y.hkl": rng.integers(0, 65535, (12, 8, 8, 2)).astype(np.uint16),
              "raw/s2_10/12X34Y.hkl": rng.integers(0, 65535, (3, 8, 8, 4)).astype(np.uint16),
              "raw/s2_20/12X34Y.hkl": rng.integers(0, 65535, (3, 4, 4, 6)).astype(np.uint16),
              "raw/misc/dem_12X34Y.hkl": rng.random((8, 8)).astype(np.float32),
              "raw/misc/s2_dates_12X34Y.hkl": np.array([10, 50, 90], np.int64)}
    for rel, a in arrays.items():
        w = WF.Writer()
        oh = w.chunked_dataset(a, chunks=tuple(max(1, s // 2) for s in a.shape)) if a.ndim > 1 else w.contiguous_dataset(a)
        w.finish({"data": oh}, str(tmp_path / "12" / "34" / rel))
    raw = job.load_raw_tile(12, 34, str(tmp_path) + "/")
    for key, rel in [("clouds", "raw/clouds/clouds_12X34Y.hkl"), ("clm", "raw/clouds/cloudmask_12X34Y.hkl"), ("s1", "raw/s1/12X34Y.hkl"),
                     ("s2_10", "raw/s2_10/12X34Y.hkl"), ("s2_20", "raw/s2_20/12X34Y.hkl"), ("dem", "raw/misc/dem_12X34Y.hkl"),
                     ("dates", "raw/misc/s2_dates_12X34Y.hkl")]:
        np.testing.assert_array_equal(raw[key], arrays[rel])
