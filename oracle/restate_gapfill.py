"""CPU oracle for the cloud / shadow gap-fill.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, from
src/preprocessing/cloud_removal.py (`CR.py` below):
  id_areas_to_interp            CR.py:774-798
  remove_cloud_and_shadows      CR.py:888-973
  make_aligned_mosaic           CR.py:578-699
  align_interp_array_randomforest  CR.py:316-575  (per-date non-negative least squares fit)
  calculate_clouds_in_mosaic    CR.py:703-732
Pinned by tests/test_oracle_gapfill.py against golden vectors captured from the imported
reference with `random.seed` fixed (the reference samples with the stdlib global RNG).
The sampling is injected (`sampler`) so that the product's deterministic GPU sampler can be
checked against the same arithmetic.
"""
from __future__ import annotations

import random

import numpy as np
from scipy import ndimage as ndi
from scipy.optimize import nnls


# ------------------------------------------------------------------------------ a6 feather weights
def feather(mask, closing):
    """CR.py:913-921 (closing 20) / :785-794 (closing 15) for one date; float64 result."""
    d = ndi.distance_transform_edt(1 - mask)
    d[d > 12] = 12
    b = 1 - d / 12
    b[b < 0.2] = 0.
    return ndi.grey_closing(b, size=closing)


def feather_stack(probs, closing, clip=False):
    out = np.copy(probs).astype(np.float32)
    if clip:
        out = np.clip(out, 0, 1)
    for t in range(out.shape[0]):
        if np.sum(out[t]) > 0:
            out[t] = feather(out[t], closing)
    return out.astype(np.float32)


def id_areas_to_interp(probs):
    """CR.py:774-798."""
    return feather_stack(probs, 15, clip=True)


# ------------------------------------------------------------------------------ a7 aligned mosaic
def _ndwi(a):
    with np.errstate(all='ignore'):
        return (a[..., 1] - a[..., 3]) / (a[..., 1] + a[..., 3])


def make_aligned_mosaic(arr, interp):
    """CR.py:578-699 (randomforest=False).  NOTE: sets interp[i] = 1 in place for dates that
    cannot be aligned (CR.py:679-680)."""
    water = np.median(_ndwi(arr), axis=0) > 0
    water = ndi.binary_dilation(1 - water, iterations=2)
    water = ndi.binary_dilation(1 - water, iterations=5)
    T, H, W, B = arr.shape
    mosaic = np.zeros((H, W, B), dtype=np.float32)
    divisor = (np.sum(1 - interp, axis=0))[..., np.newaxis]
    for i in range(T):
        m_i = np.logical_and(interp[i] < 0.25, water == 0)
        ref = np.zeros((H, W, B), dtype=np.float32)
        cnt = np.zeros((H, W, B), dtype=np.float32)
        for b in range(T):
            if b != i:
                m = np.logical_and(np.logical_and(interp[i] < 0.25, interp[b] < 1), water == 0)
                ref[m * m_i] += arr[b][m * m_i]
                cnt[m * m_i] += 1
        with np.errstate(all='ignore'):
            ref = ref / cnt
        m_i[cnt[..., 0] == 0] = 0.
        src = arr[i][m_i]
        ref = ref.reshape(H * W, B)
        ref = ref[~np.isnan(ref).any(axis=1)]
        if src.shape[0] > 1000 and ref.shape[0] > 1000:
            src = src[:ref.shape[0]]
            ref = ref[:src.shape[0]]
            med_ref, std_ref = np.nanmedian(ref, axis=0), np.nanstd(ref, axis=0)
            med_src, std_src = np.nanmedian(src, axis=0), np.nanstd(src, axis=0)
            k = std_ref / std_src
            add = med_ref - med_src * k
            a_i = np.copy(arr[i])
            a_i[water == 0] = a_i[water == 0] * k + add
            mosaic = mosaic + (1 - interp[i][..., np.newaxis]) * a_i
        elif np.mean(water < 0.9):
            interp[i] = 1.
    divisor[divisor < 0] = 0.
    with np.errstate(all='ignore'):
        mosaic = mosaic / divisor
    mosaic[np.isnan(mosaic)] = np.percentile(arr, 10, axis=0)[np.isnan(mosaic)]
    mosaic = np.maximum(mosaic, np.min(arr, axis=0))
    mosaic = np.minimum(mosaic, np.max(arr, axis=0))
    return mosaic


# ------------------------------------------------------------------------------ a8 per-date NNLS alignment
def snow_prob(arr):
    """CR.py:348-370 (returns the float probability, unlike process_tile's boolean variant)."""
    with np.errstate(all='ignore'):
        ndsi = (arr[..., 1] - arr[..., 8]) / (arr[..., 1] + arr[..., 8])
        ndsi[ndsi < 0.10] = 0.
        ndsi[ndsi > 0.42] = 0.42
        p = (ndsi - 0.1) / 0.32
        p[arr[..., 3] < 0.10] = 0.
        p[np.logical_and(arr[..., 3] > 0.35, p > 0)] = 1.
        p[arr[..., 0] < 0.10] = 0.
        p[np.logical_and(arr[..., 0] > 0.22, p > 0)] = 1.
        p[(arr[..., 0] / arr[..., 2]) < 0.75] = 0.
    return p


def evi_unclipped(x):
    """CR.py:332-345."""
    with np.errstate(all='ignore'):
        e = 2.5 * ((x[..., 3] - x[..., 2]) / (x[..., 3] + (6 * x[..., 2]) - (7.5 * x[..., 0]) + 1))
    return np.clip(e, -1.5, 1.5)


def reference_sampler(evi_vals, n_rows, rng=random):
    """CR.py:453-500: EVI-stratified sample with the stdlib global RNG (tails x10, five quintile
    strata of n//5, shuffled).  Returns the row indices (with repeats)."""
    n_samples = np.minimum(90000, n_rows)
    n_i = n_samples // 5
    b2, b20, b40, b60, b80, b98 = (np.percentile(evi_vals, q) for q in (2, 20, 40, 60, 80, 98))
    p2 = np.argwhere(evi_vals < b2).squeeze()
    p20 = np.argwhere(evi_vals < b20).squeeze()
    p40 = np.argwhere(np.logical_and(evi_vals >= b20, evi_vals < b40)).squeeze()
    p60 = np.argwhere(np.logical_and(evi_vals >= b40, evi_vals < b60)).squeeze()
    p80 = np.argwhere(np.logical_and(evi_vals >= b60, evi_vals < b80)).squeeze()
    p100 = np.argwhere(evi_vals >= b80).squeeze()
    p98 = np.argwhere(evi_vals >= b98).squeeze()
    p98 = np.repeat(p98, 10)
    p2 = np.repeat(p2, 10)
    for p in (p2, p98, p20, p40, p60, p80, p100):
        rng.shuffle(p)
    sample = np.concatenate([p2, p20[:n_i], p40[:n_i], p60[:n_i], p80[:n_i], p100[:n_i], p98])
    rng.shuffle(sample)
    return sample[:n_rows]


def expected_weights(evi_vals, n_rows):
    """sampler = "expected" (the product's default, gapfill.hip k_row_weights_all): instead of DRAWING reference_sampler's
    sample, every candidate row enters the fit with its EXPECTED multiplicity under that scheme -- a quintile stratum of
    c rows is cut to n_i = min(90000, n) // 5 after a shuffle, so each of its rows survives with probability
    min(1, n_i / c); the 2 % tails are appended ten times each.  (The final `sample[:n_rows]` cut after the last shuffle
    scales every expectation by the same factor, which leaves a least-squares solution unchanged.)  Thresholds are the
    reference's np.percentile values on the float32 EVI column."""
    n_i = np.minimum(90000, n_rows) // 5
    b2, b20, b40, b60, b80, b98 = (np.percentile(evi_vals, q) for q in (2, 20, 40, 60, 80, 98))
    stratum = ((evi_vals >= b20).astype(int) + (evi_vals >= b40) + (evi_vals >= b60) + (evi_vals >= b80))
    cnt = np.bincount(stratum, minlength=5)
    w = np.minimum(1.0, n_i / np.maximum(cnt, 1).astype(np.float64))[stratum]
    w[cnt[stratum] == 0] = 0.0
    w = w.astype(np.float32)
    w[evi_vals < b2] += 10.0
    w[evi_vals >= b98] += 10.0
    return w


def align_date(fill, array, date, interp, mosaic, water_mask, sampler=reference_sampler):
    """CR.py:316-575 for one date: returns (prediction [H,W,10] to blend in, flagged).
    sampler: a callable (evi, n_rows) -> row indices (the reference draws them with the stdlib RNG), or the string
    "expected": weighted least squares with expected_weights() above, no RNG."""
    T, H, W, B = array.shape
    snow = np.mean(snow_prob(array), axis=0)[..., np.newaxis]
    if not (np.sum(interp[date] > 0) > 0 and np.sum(interp[date] == 0) > 0):
        return fill, []
    if not np.mean(np.logical_and(interp[date] < 1, water_mask <= 1)) > 0.01:
        return fill, []
    n_cur = np.sum(np.logical_and(interp[date] == 0, water_mask <= 1))
    if n_cur > 40000:
        t0, t1 = max(date, 0), date + 1
    else:
        t0 = max(date - 2, 0) if date == T - 1 else max(date - 1, 0)
        t1 = min(date + 2, T)
    ys, xs = [], []
    for t in range(t0, t1):
        req = np.logical_and(interp[t] == 0, water_mask < 1)
        ys.append(np.concatenate([array[t], snow], axis=-1)[req])
        xs.append(np.concatenate([mosaic, snow], axis=-1)[req])
    if n_cur > 40000:
        X, Y = xs[0], ys[0]
    else:
        X, Y = np.concatenate(xs, axis=0), np.concatenate(ys, axis=0)
    if isinstance(sampler, str):
        assert sampler == "expected", sampler
        sw = np.sqrt(expected_weights(evi_unclipped(Y), X.shape[0]).astype(np.float64))[:, np.newaxis]
        X, Y = np.copy(X), np.copy(Y)
    else:
        idx = sampler(evi_unclipped(Y), X.shape[0])
        X, Y = X[idx], Y[idx]
        sw = None
    out = np.copy(fill)
    full = np.concatenate([fill, snow], axis=-1).reshape(H * W, B + 1)
    sel = np.logical_and(interp[date] > 0, water_mask <= 1)
    for band in range(10):
        train_x = np.copy(X)
        X[..., band] = np.clip(X[..., band], 0.005, 1)        # AFTER the copy (CR.py:550): band b is fitted on
        if sw is None:
            beta, _ = nnls(train_x.astype(np.float64), Y[..., band].astype(np.float64))   # cols < b clipped, >= b raw
        else:                                                                             # min sum_i w_i (y_i - x_i . beta)^2
            beta, _ = nnls(train_x.astype(np.float64) * sw, Y[..., band].astype(np.float64) * sw[:, 0])
        pred = (full.astype(np.float64) @ beta).reshape(H, W)
        out[sel, band] = pred[sel]
    return out, []


# ------------------------------------------------------------------------------ a9 clouds left in the mosaic
def calculate_clouds_in_mosaic(mosaic, interp, pfcps):
    """CR.py:703-732."""
    only1 = np.sum(1 - (interp > 0), axis=0).squeeze() < 2
    if len(pfcps.shape) == 3 and pfcps.shape[0] > 1:
        pfcps = pfcps[0]
    pfcps = ndi.binary_dilation(pfcps, iterations=10)
    only1 = np.maximum(only1, pfcps.squeeze())
    if np.sum(only1) == np.prod(only1.shape):
        return np.zeros_like(only1)
    ref_blue = np.percentile(mosaic[..., 0][~only1], 99)
    ref_red = np.percentile(mosaic[..., 2][~only1], 99)
    c = ((mosaic[..., 0] > ref_blue) * (mosaic[..., 2] > ref_red) * only1 * (np.sum(mosaic[..., :3], axis=-1) < 1))
    c[pfcps.squeeze() > 0] = 0.
    c = ndi.binary_dilation(1 - c, iterations=3)
    c = ndi.binary_dilation(1 - c, iterations=8)
    return c


def remove_cloud_and_shadows(tiles, probs, pfcps, sampler=reference_sampler, return_mosaic=False):
    """CR.py:888-973.  tiles [T,H,W,10] is modified in place; returns (tiles, interp, to_remove)."""
    interp = feather_stack(probs, 20)
    mosaic = make_aligned_mosaic(tiles, interp)
    water_mask = _ndwi(np.median(tiles, axis=0)) > 0.0
    to_remove = []
    for date in range(tiles.shape[0]):
        fill = np.zeros_like(tiles[date])
        fill[interp[date] > 0] = mosaic[interp[date] > 0]
        pred, rem = align_date(fill, tiles, date, interp, mosaic, water_mask, sampler)
        w = interp[date][..., np.newaxis]
        tiles[date] = tiles[date] * (1 - w) + pred * w
        if len(rem) > 0:
            to_remove.append(date)
        if np.mean(interp[date] == 1) == 1:
            to_remove.append(date)
    interp = interp + calculate_clouds_in_mosaic(mosaic, interp, pfcps)[np.newaxis]
    interp[interp > 1] = 1.
    if return_mosaic:
        return tiles, interp, to_remove, mosaic
    return tiles, interp, to_remove
