"""hickle-shaped HDF5 fixtures written by the REAL h5py / libhdf5 (this container only).

    /opt/conda/bin/python3.9 tools/gen_golden_hkl.py        # writes tests/golden/hkl/*.hkl + expected.npz

The job stores its raw arrays with `hkl.dump(arr, path, mode='w', compression='gzip')`
(src/download_and_predict_job.py:462-463, :592-633) and reads them back with `hkl.load` (:684-714, :599
`list(hkl.load(clean_steps_file))`).  hickle itself is not installed anywhere in this image; it is a thin layer over
`h5py.File(...).create_dataset(name, data=arr, **kwargs)` plus a few attributes, so the files below are laid out the way
hickle 3.4 (`/data_0`, attrs CLASS / VERSION / type), hickle 4 and 5 (`/data`, attrs HICKLE_VERSION / base_type / type) lay
theirs out -- array at the root, list of arrays as a group of `data_i` datasets -- by h5py 3.3.0 on HDF5 1.10.6 with h5py's
defaults (libver earliest: superblock v0, old-style groups, object headers v1, chunk B-trees v1, auto-chunking, gzip level 4
and optionally the shuffle filter).  What ttc_read_hkl must return for each file is stored in expected.npz.
"""
import os
import pickle
import sys

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden")), "hkl")


def smooth_u16(rng, shape):
    """band-like data (compressible, like the job's reflectance * 65535), with a few saturated / zero pixels"""
    base = rng.random(shape[:1] + (shape[1] // 8 + 2, shape[2] // 8 + 2) + shape[3:])
    up = np.kron(base, np.ones((1, 8, 8) + (1,) * (len(shape) - 3)))[:, :shape[1], :shape[2]]
    a = (up * 9000 + rng.integers(0, 40, shape)).astype(np.uint16)
    a[0, :3, :3] = 65535
    a[-1, -2:, :] = 0
    return a


def h3_root(f):
    f.attrs["CLASS"] = b"hickle"
    f.attrs["VERSION"] = b"3.4.9"


def h4_root(f, version):
    f.attrs["HICKLE_VERSION"] = version
    f.attrs["HICKLE_PYTHON_VERSION"] = "3.9.7"


def h4_attrs(d, base_type, cls):
    d.attrs["base_type"] = base_type
    d.attrs["type"] = np.array(pickle.dumps(cls))      # hickle 4/5 store the pickled class as an opaque string


def main():
    os.makedirs(OUT, exist_ok=True)
    for fn in os.listdir(OUT):
        os.remove(os.path.join(OUT, fn))
    rng = np.random.default_rng(20240)
    exp = {}

    def path(name):
        return os.path.join(OUT, name + ".hkl")

    # hickle 3.4: /data_0, gzip, auto-chunked
    a = smooth_u16(rng, (3, 40, 44, 4))
    with h5py.File(path("h3_s2_10_u16"), "w") as f:
        h3_root(f)
        d = f.create_dataset("data_0", data=a, compression="gzip")
        d.attrs["type"] = [b"ndarray"]
    exp["h3_s2_10_u16"] = a

    # hickle 4: /data, gzip + shuffle
    a = smooth_u16(rng, (3, 20, 22, 6))
    with h5py.File(path("h4_s2_20_u16_shuffle"), "w") as f:
        h4_root(f, "4.0.4")
        d = f.create_dataset("data", data=a, compression="gzip", shuffle=True)
        h4_attrs(d, b"ndarray", np.ndarray)
    exp["h4_s2_20_u16_shuffle"] = a

    # hickle 5: /data float32 cloud probabilities, explicit ragged chunks, gzip level 9
    a = rng.random((5, 37, 41)).astype(np.float32)
    with h5py.File(path("h5_clouds_f32"), "w") as f:
        h4_root(f, "5.0.2")
        d = f.create_dataset("data", data=a, compression="gzip", compression_opts=9, chunks=(2, 16, 16))
        h4_attrs(d, b"ndarray", np.ndarray)
        f.create_dataset("aux", data=np.arange(4, dtype=np.int32))        # a sibling the default lookup must not pick
    exp["h5_clouds_f32"] = a

    # Sentinel-1: many small chunks -> a chunk B-tree with more than one level
    a = smooth_u16(rng, (12, 64, 64, 2))
    with h5py.File(path("h4_s1_u16_many_chunks"), "w") as f:
        h4_root(f, "4.0.4")
        d = f.create_dataset("data", data=a, compression="gzip", chunks=(1, 8, 8, 2))
        h4_attrs(d, b"ndarray", np.ndarray)
    exp["h4_s1_u16_many_chunks"] = a

    # DEM: contiguous float32 (no compression requested)
    a = (rng.random((46, 52)) * 900).astype(np.float32)
    with h5py.File(path("h4_dem_f32_contig"), "w") as f:
        h4_root(f, "4.0.4")
        d = f.create_dataset("data", data=a)
        h4_attrs(d, b"ndarray", np.ndarray)
    exp["h4_dem_f32_contig"] = a

    # image dates: a python list of ints -> one int64 dataset (hickle 4: base_type list, contiguous; hickle 3: data_0, gzip)
    dates = np.array([-20, 12, 33, 95, 130, 171, 200, 244, 290, 301, 350, 380], np.int64)
    with h5py.File(path("h4_dates_list_i64"), "w") as f:
        h4_root(f, "4.0.4")
        d = f.create_dataset("data", data=dates)
        h4_attrs(d, b"list", list)
    exp["h4_dates_list_i64"] = dates
    with h5py.File(path("h3_dates_i64_gzip"), "w") as f:
        h3_root(f)
        d = f.create_dataset("data_0", data=dates, compression="gzip")
        d.attrs["type"] = [b"list"]
    exp["h3_dates_i64_gzip"] = dates

    # float64 array (np.save'd intermediates re-dumped through hickle)
    a = rng.standard_normal((7, 9, 3))
    with h5py.File(path("h5_f64"), "w") as f:
        h4_root(f, "5.0.2")
        d = f.create_dataset("data", data=a, compression="gzip", shuffle=True)
        h4_attrs(d, b"ndarray", np.ndarray)
    exp["h5_f64"] = a

    # containers: hickle 4/5 `data` GROUP with data_0, data_1 ...; hickle 3 `data_0` group with data_0 ...
    a0, a1 = smooth_u16(rng, (2, 24, 24, 4)), rng.random((2, 24, 24)).astype(np.float32)
    with h5py.File(path("h4_nested_list"), "w") as f:
        h4_root(f, "4.0.4")
        g = f.create_group("data")
        h4_attrs(g, b"list", list)
        for i, x in enumerate((a0, a1)):
            d = g.create_dataset(f"data_{i}", data=x, compression="gzip")
            h4_attrs(d, b"ndarray", np.ndarray)
    exp["h4_nested_list"] = a0
    exp["h4_nested_list:data/data_1"] = a1
    with h5py.File(path("h3_nested_list"), "w") as f:
        h3_root(f)
        g = f.create_group("data_0")
        g.attrs["type"] = [b"list"]
        for i, x in enumerate((a1, a0)):
            d = g.create_dataset(f"data_{i}", data=x, compression="gzip", shuffle=True)
            d.attrs["type"] = [b"ndarray"]
    exp["h3_nested_list"] = a1
    exp["h3_nested_list:data_0/data_1"] = a0

    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    with open(os.path.join(OUT, "GENERATOR.txt"), "w") as fh:
        fh.write(f"tools/gen_golden_hkl.py: python {sys.version.split()[0]}, h5py {h5py.__version__}, "
                 f"HDF5 {h5py.version.hdf5_version}, numpy {np.__version__}\n")
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
