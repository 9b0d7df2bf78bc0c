"""Golden vectors for the bilinear resizes of the path, from the REAL scikit-image (this container only).

    /opt/conda/bin/python3.9 tools/gen_golden_resize.py      # writes tests/golden/resize.npz

Calls `skimage.transform.resize(img, shape, order=1)` exactly as the reference does
(src/download_and_predict_job.py:741-743, :759-781 for the 20 m -> 10 m bands;
src/resegment_tiles_wide.py:1190-1236 and :1354-1355 for the border-mosaic weight tables) on the shapes
those call sites produce (tools/resize_cases.py builds the inputs; the test rebuilds them), and stores the outputs + a checksum per input.  Fails if scikit-image is not importable: this
fixture exists to pin oracle.restate_numpy.resize_bilinear / resegment._resize against the real thing.
"""
import os
import sys

import numpy as np
import skimage
from skimage.transform import resize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from resize_cases import cases as resize_cases  # noqa: E402


def main():
    out = {}
    cases = resize_cases()
    for k, (img, shape) in enumerate(cases):
        out[f"insum_{k}"] = np.array([np.asarray(img, np.float64).sum(), np.abs(np.asarray(img, np.float64)).max()])   # the input itself is code: tools/resize_cases.py
        out[f"shape_{k}"] = np.array(shape)
        out[f"out_{k}"] = resize(img, shape, order=1)
    out["n"] = len(cases)
    out["skimage_version"] = np.array(skimage.__version__)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "resize.npz"), **out)
    print("resize.npz:", len(cases), "cases, scikit-image", skimage.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
