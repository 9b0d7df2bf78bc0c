"""Golden vectors for the bilinear resizes of the path, from the REAL scikit-image (this container only).

    /opt/conda/bin/python3.9 tools/gen_golden_resize.py      # writes tests/golden/resize.npz

Calls `skimage.transform.resize(img, shape, order=1)` exactly as the reference does
(src/download_and_predict_job.py:741-743, :759-781 for the 20 m -> 10 m bands;
src/resegment_tiles_wide.py:1190-1236 and :1354-1355 for the border-mosaic weight tables) on the shapes
those call sites produce, and stores inputs + outputs.  Fails if scikit-image is not importable: this
fixture exists to pin oracle.restate_numpy.resize_bilinear / resegment._resize against the real thing.
"""
import os
import sys

import numpy as np
import skimage
from skimage.transform import resize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("TTC_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))


def fspecial_gauss(size, sigma):
    # the same Gaussian window the reference builds for its blend weights (an input here, not the thing under test)
    x, y = np.mgrid[-size // 2 + 1:size // 2 + 1, -size // 2 + 1:size // 2 + 1]
    return np.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))


def main():
    rng = np.random.default_rng(404)
    out = {}
    cases = []
    # a3: 20 m -> 10 m, float32 reflectance, even and odd grids (309 is what a 618 tile has)
    for (h, w) in [(20, 18), (21, 19), (155, 155), (309, 309)]:
        cases.append((rng.random((h, w)).astype(np.float32), (2 * h, 2 * w)))
    # odd-grid branch: the 40 m mean of mid[1:, 1:] resized to (width - 1, height - 1)
    cases.append((rng.random((154, 154)).astype(np.float32), (617, 617)))
    cases.append((rng.random((10, 9)).astype(np.float32), (41, 37)))
    # f2 window weights: half of a 670 / 684 / 620 / 588 Gaussian squeezed to the half-window
    cases.append((fspecial_gauss(670, 150)[335:, :], (335, 206)))
    cases.append((fspecial_gauss(684, 150)[:342, :], (342, 220)))
    cases.append((fspecial_gauss(620, 150)[:, 310:], (206, 310)))
    cases.append((fspecial_gauss(412, 95)[206:, :], (206, 220)))
    cases.append((fspecial_gauss(216, 44)[:108, :], (108, 168)))
    # f2 stack ramps: a square Gaussian to a non-square mosaic, identity resizes, transposed ramps (one axis shrinks, one grows)
    cases.append((fspecial_gauss(300, 300 / 5.25), (300, 618)))
    cases.append((fspecial_gauss(618, 618 / 5.25), (618, 320)))
    lin = (np.ones((84, 320)) * (np.arange(84) / 84)[:, None]) ** 1.2
    cases.append((lin, (84, 320)))
    cases.append((np.concatenate([lin, np.zeros((216, 320))], axis=0).T, (300, 320)))
    cases.append((np.flipud(np.concatenate([np.zeros((534, 320)), lin], axis=0).T), (618, 320)))
    # generic random: both axes shrink by non-integer factors; one shrinks / one grows
    cases.append((rng.random((300, 40)), (150, 684)))
    cases.append((rng.random((97, 131)), (41, 50)))
    for k, (img, shape) in enumerate(cases):
        out[f"in_{k}"] = img
        out[f"shape_{k}"] = np.array(shape)
        out[f"out_{k}"] = resize(img, shape, order=1)
    out["n"] = len(cases)
    out["skimage_version"] = np.array(skimage.__version__)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "resize.npz"), **out)
    print("resize.npz:", len(cases), "cases, scikit-image", skimage.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
