"""Host-side construction of the 12 x T temporal operator applied by the HIP kernel
`k_tile_temporal`.

The reference regrids irregular acquisition dates onto a fixed 24-step (15-day) grid by a
<=4-image distance-weighted blend (calculate_and_save_best_images,
src/downloading/utils.py:176-347), runs a 2nd-order Whittaker smoother (lambda = 100) on the
24 steps (Smoother, src/preprocessing/whittaker_smoother.py:10-47) and averages consecutive pairs
to 12 months (:64-67).  All three are linear in the images for fixed dates, so the whole stage is
one matrix  W = P . (I + lambda D2'D2)^-1 . R(dates)  of shape [12, T]; only R depends on the tile.
W is built here in float64 (microseconds) and shipped to the GPU as float32.
"""
from __future__ import annotations

import numpy as np

GRID = np.arange(0, 360, 15)          # 24 ideal acquisition days


def _side_weights(near_first, dist, closest):
    """Weights of one side (prior or after).  `dist` is ordered as the reference keeps it
    (prior: far..near, after: near..far); `near_first` tells which end is the near image."""
    w = np.abs(1.0 - dist / closest)
    if len(w) == 2:
        near, far = (0, 1) if near_first else (1, 0)
        w[far] = abs((dist[near] / dist[far]) * w[near])
    return w


def regrid_matrix(image_dates) -> np.ndarray:
    """R [24, T] float32 such that regridded = R @ images (utils.py:176-347).

    Raises ValueError when the reference itself would raise (duplicate dates make its
    ratio / index lists disagree); the caller then mirrors job.py:1073-1080 (all-zero series)."""
    dates = np.asarray(image_dates).astype(np.int64).copy()
    far_neg = dates < -100
    dates[far_neg] = dates[far_neg] % 365                    # utils.py:190
    T = len(dates)
    R = np.zeros((24, T), dtype=np.float32)
    dmin, dmax = dates.min(), dates.max()
    for row, g in enumerate(GRID):
        d = dates - g
        prior = d[d < 5][-2:]                                 # <= 2 images before (or within 5 days after) g
        if prior.size:
            prior = prior[prior > prior.max() - 100]
        after = d[d >= -5][:2]
        if after.size:
            after = after[after < after.min() + 100]
        p_shift = a_shift = 0
        if prior.size == 0:                                   # wrap to last year's final image, or mirror
            if dmin >= 90:
                prior, p_shift = d[-1:], 365
            else:
                prior = after
        if after.size == 0:
            if dmax <= 270:
                after, a_shift = d[:1], 365
            else:
                after = prior
        pd = np.maximum(np.abs(prior - p_shift).astype(np.float64), 1.0)
        ad = np.maximum(np.abs(after + a_shift).astype(np.float64), 1.0)
        closest = max(pd[-1] + ad[0], 2.0)
        pw = _side_weights(False, pd, closest)
        aw = _side_weights(True, ad, closest)
        total = pw.sum() + aw.sum()
        pw, aw = (pw / total).astype(np.float32), (aw / total).astype(np.float32)
        p_idx = np.flatnonzero(np.isin(dates, g + prior))[:2]
        a_idx = np.flatnonzero(np.isin(dates, g + after))[-2:]
        if len(p_idx) != len(pw) or len(a_idx) != len(aw):
            raise ValueError("regrid_matrix: ambiguous (duplicate) image dates")
        np.add.at(R[row], p_idx, pw)
        np.add.at(R[row], a_idx, aw)
    return R


def whittaker_monthly_matrix(n=24, lmbd=100.0, out=12) -> np.ndarray:
    """M [out, n] float64: Whittaker smoothing (I + lmbd D'D)^-1 followed by the mean of each
    n/out consecutive steps (whittaker_smoother.py:25-36, :64-67)."""
    D = np.diff(np.eye(n), n=2, axis=0)                      # (n-2) x n second differences
    A = np.eye(n) + lmbd * D.T @ D
    P = np.kron(np.eye(out), np.full((1, n // out), 1.0 / (n // out)))
    return P @ np.linalg.inv(A)


_M = None


def temporal_operator(image_dates) -> np.ndarray:
    """W [12, T] float32.  All-zero when the regrid is undefined (job.py:1073-1080)."""
    global _M
    if _M is None:
        _M = whittaker_monthly_matrix()
    try:
        R = regrid_matrix(image_dates).astype(np.float64)
    except (ValueError, IndexError):
        return np.zeros((12, len(image_dates)), dtype=np.float32)
    return (_M @ R).astype(np.float32)
