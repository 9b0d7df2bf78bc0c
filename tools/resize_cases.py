"""The input images of tests/golden/resize.npz, as code: tools/gen_golden_resize.py (real scikit-image, /opt/conda python3.9) and
tests/test_oracle_golden.py build them from here, so the fixture stores the expected OUTPUTS only (and a checksum per input)."""
import numpy as np


def fspecial_gauss(size, sigma):
    # the same Gaussian window the reference builds for its blend weights (an input here, not the thing under test)
    x, y = np.mgrid[-size // 2 + 1:size // 2 + 1, -size // 2 + 1:size // 2 + 1]
    return np.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))


def cases():
    """-> list of (image, output shape): the shapes the reference's call sites produce (job.py:741-781, resegment_tiles_wide.py:1190-1236, :1354-1355)"""
    rng = np.random.default_rng(404)
    out = []
    # a3: 20 m -> 10 m, float32 reflectance, even and odd grids (309 is what a 618 tile has)
    for (h, w) in [(20, 18), (21, 19), (155, 155), (309, 309)]:
        out.append((rng.random((h, w)).astype(np.float32), (2 * h, 2 * w)))
    # odd-grid branch: the 40 m mean of mid[1:, 1:] resized to (width - 1, height - 1)
    out.append((rng.random((154, 154)).astype(np.float32), (617, 617)))
    out.append((rng.random((10, 9)).astype(np.float32), (41, 37)))
    # f2 window weights: half of a 670 / 684 / 620 / 588 Gaussian squeezed to the half-window
    out.append((fspecial_gauss(670, 150)[335:, :], (335, 206)))
    out.append((fspecial_gauss(684, 150)[:342, :], (342, 220)))
    out.append((fspecial_gauss(620, 150)[:, 310:], (206, 310)))
    out.append((fspecial_gauss(412, 95)[206:, :], (206, 220)))
    out.append((fspecial_gauss(216, 44)[:108, :], (108, 168)))
    # f2 stack ramps: a square Gaussian to a non-square mosaic, identity resizes, transposed ramps (one axis shrinks, one grows)
    out.append((fspecial_gauss(300, 300 / 5.25), (300, 618)))
    out.append((fspecial_gauss(618, 618 / 5.25), (618, 320)))
    lin = (np.ones((84, 320)) * (np.arange(84) / 84)[:, None]) ** 1.2
    out.append((lin, (84, 320)))
    out.append((np.concatenate([lin, np.zeros((216, 320))], axis=0).T, (300, 320)))
    out.append((np.flipud(np.concatenate([np.zeros((534, 320)), lin], axis=0).T), (618, 320)))
    # generic random: both axes shrink by non-integer factors; one shrinks / one grows
    out.append((rng.random((300, 40)), (150, 684)))
    out.append((rng.random((97, 131)), (41, 50)))
    return out
