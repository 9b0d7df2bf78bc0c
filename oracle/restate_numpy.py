"""CPU oracle (numpy/scipy) for the numeric stages of the tree-cover hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line it restates; `job.py` = src/download_and_predict_job.py.
Pinned by tests/test_oracle_golden.py against fixtures captured from the
imported reference by tools/gen_golden.py.

Array conventions are the reference's: time-first, channel-last
([T, X, Y, C]), float32 unless noted.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

# --------------------------------------------------------------------------- constants
# job.py:1829-1842
MIN_ALL = np.array([0.006576638437476157, 0.0162050812542916, 0.010040436408026246,
                    0.013351644159609368, 0.01965362020294499, 0.014229037918669413,
                    0.015289539940489814, 0.011993591210803388, 0.008239871824216068,
                    0.006546120393682765, 0.0, 0.0, 0.0, -0.1409399364817101,
                    -0.4973397113668104, -0.09731556326714398, -0.7193834232943873])
MAX_ALL = np.array([0.2691233691920348, 0.3740291447318227, 0.5171435111009385,
                    0.6027466239414053, 0.5650263218127718, 0.5747005416952773,
                    0.5933928435187305, 0.6034943160143434, 0.7472037842374304,
                    0.7000076295109483, 0.4, 0.948334642387533, 0.6729257769285485,
                    0.8177635298774327, 0.35768999002433816, 0.7545951919107605,
                    0.7602693339366691])


# --------------------------------------------------------------------------- shape reconciliation
def adjust_shape(arr, width, height):
    """job.py:260-310: bring axes 1 / 2 of [T, X, Y, C] (3-D: [T, X, Y]; 2-D: [X, Y]) to width x height.  One pixel short: the first
    row / column once more (np.pad 'edge' with (1, 0)); an even shortfall: half at either end; one pixel long: the first row / column
    goes; an even excess: half at either end.  Written per axis as a gather -- index i of the result reads clip(i + off, 0, n - 1).
    The reference's remaining branches (odd differences of 3 or more) leave the axis at the WRONG length (pads 2*(d//2), crops only
    (d//2) in total) and process_tile raises on its next assignment: here that is a ValueError.  Squeezes like the reference."""
    arr = np.asarray(arr)
    arr = arr[:, :, :, np.newaxis] if arr.ndim == 3 else arr
    arr = arr[np.newaxis, :, :, np.newaxis] if arr.ndim == 2 else arr
    for ax, want in ((1, int(width)), (2, int(height))):
        n = arr.shape[ax]
        if n == want:
            continue
        d = abs(want - n)
        if d // 2 != 0 and d % 2 != 0:
            raise ValueError(f"adjust_shape: axis {ax} is {n}, wanted {want}: the reference does not reconcile an odd difference of {d}")
        off = (1 if n > want else -1) * max(d // 2, 1)
        arr = np.take(arr, np.clip(np.arange(want) + off, 0, n - 1), axis=ax)
    return arr.squeeze()


# --------------------------------------------------------------------------- codecs
def to_float32(a):
    """src/tof/tof_downloading.py:64-72."""
    if not np.issubdtype(a.dtype, np.floating):
        a = np.float32(a) / np.float32(65535.)
    return a.astype(np.float32)


def to_int16(a):
    """src/tof/tof_downloading.py:51-61 (really uint16)."""
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


def float_to_int16(arr, precision=1000):
    """job.py:174-180."""
    arr = np.array(arr, dtype=np.float32)
    arr[np.isnan(arr)] = -32768
    arr = np.clip(arr, (-32768 / precision), (32767 / precision))
    arr = arr * precision
    return np.int16(arr)


def convert_to_db(x, min_db=22):
    """job.py:74-89."""
    x = 10 * np.log10(x + 1 / 65535)
    x[x < -min_db] = -min_db
    x = (x + min_db) / min_db
    return np.clip(x, 0, 1)


def s1_to_db(s1_u16):
    """job.py:699-708: u16 -> [0,1], saturated pixels -> per-image median, dB on both pols."""
    s1 = np.float32(s1_u16) / 65535
    for i in range(s1.shape[0]):
        s1_i = s1[i]
        s1_i[s1_i == 1] = np.median(s1_i[s1_i < 65535], axis=0)
        s1[i] = s1_i
    s1[..., -1] = convert_to_db(s1[..., -1], 22)
    s1[..., -2] = convert_to_db(s1[..., -2], 22)
    return s1.astype(np.float32)


# --------------------------------------------------------------------------- indices
def _c01(x):
    return np.clip(x, 0, 1)


def evi(x):
    """src/preprocessing/indices.py:15-27."""
    b, r, n = _c01(x[..., 0]), _c01(x[..., 2]), _c01(x[..., 3])
    return np.clip(2.5 * ((n - r) / (n + (6 * r) - (7.5 * b) + 1)), -1.5, 1.5)


def bi(x):
    """src/preprocessing/indices.py:47-54."""
    b11, b4, b8, b2 = _c01(x[..., 8]), _c01(x[..., 2]), _c01(x[..., 3]), _c01(x[..., 0])
    return np.clip(((b11 + b4) - (b8 + b2)) / (((b11 + b4) + (b8 + b2)) + 1e-5), -1, 1)


def msavi2(x):
    """src/preprocessing/indices.py:30-44."""
    r, n = _c01(x[..., 2]), _c01(x[..., 3])
    s = (2 * n + 1) ** 2 - 8 * (n - r)
    s[s < 0] = 0.
    return np.clip((2 * n + 1 - np.sqrt(s)) / 2, -1, 1)


def grndvi(x):
    """src/preprocessing/indices.py:4-12."""
    n, g, r = _c01(x[..., 3]), _c01(x[..., 1]), _c01(x[..., 2])
    return (n - (g + r)) / ((n + (g + r)) + 1e-5)


def make_indices(arr):
    """job.py:998-1006; channel order evi, bi, msavi2, grndvi."""
    out = np.zeros(arr.shape[:3] + (4,), dtype=np.float32)
    out[..., 0] = evi(arr)
    out[..., 1] = bi(arr)
    out[..., 2] = msavi2(arr)
    out[..., 3] = grndvi(arr)
    return out


# --------------------------------------------------------------------------- date regrid
def regrid_matrix(image_dates) -> np.ndarray:
    """src/downloading/utils.py:176-302 as a 24 x T float32 matrix.

    The reference blends, for each 15-day grid step, up to two prior and two
    following images with distance weights.  The blend is linear in the
    images, so the whole function is `R @ images`; this builds R.  Raises
    ValueError where the reference would raise (ratio/index count mismatch
    from duplicate dates) -- callers mirror job.py:1073-1080 (zeros).
    """
    dates = np.array(image_dates).copy()
    dates[dates < -100] = dates[dates < -100] % 365
    T = len(dates)
    R = np.zeros((24, T), dtype=np.float32)
    for row, g in enumerate(range(0, 360, 15)):
        d = np.array([(x - g) for x in dates])
        prior = d[np.where(d < 5)][-2:]
        if prior.shape[0] > 0:
            prior = np.array(prior[prior > (-100 + np.max(prior))]).flatten()
        after = d[np.where(d >= -5)][:2]
        if after.shape[0] > 0:
            after = np.array(after[after < (100 + np.min(after))])
        after_flag = prior_flag = 0
        if len(prior) == 0:
            if np.min(dates) >= 90:
                prior, prior_flag = d[-1:], 365
            else:
                prior = after
        if len(after) == 0:
            if np.max(dates) <= 270:
                after, after_flag = d[:1], 365
            else:
                after = prior
        pc = np.maximum(abs(prior - prior_flag), 1.)
        ac = np.maximum(abs(after + after_flag), 1.)
        closest = np.maximum(abs(pc[-1]) + abs(ac[0]), 2)
        pm = abs(1 - (pc / closest))
        am = abs(1 - (ac / closest))
        if len(pm) == 2:
            pm[0] = abs((pc[1] / pc[0]) * pm[1])
        if len(am) == 2:
            am[1] = abs((ac[0] / ac[1]) * am[0])
        div = np.sum(np.concatenate([abs(pm), abs(am)]))
        pr, ar = (pm / div).astype(np.float32), (am / div).astype(np.float32)
        p_dates, a_dates = g + prior, g + after
        p_idx = sorted(set(i for i, v in enumerate(dates) if v in p_dates))
        a_idx = sorted(set(i for i, v in enumerate(dates) if v in a_dates))
        if len(a_idx) > 2:
            a_idx = a_idx[-2:]
        if len(p_idx) > 2:
            p_idx = p_idx[:2]
        if len(p_idx) != len(pr) or len(a_idx) != len(ar):
            raise ValueError("regrid: duplicate dates (reference raises here too)")
        for i, w in zip(p_idx, pr):
            R[row, i] += w
        for i, w in zip(a_idx, ar):
            R[row, i] += w
    return R


def regrid(img, image_dates):
    """src/downloading/utils.py:304-347: apply the blend in float32 -> [24, X, Y, C]."""
    R = regrid_matrix(image_dates)
    return np.einsum('gt,txyc->gxyc', R, img.astype(np.float32)).astype(np.float32)


# --------------------------------------------------------------------------- Whittaker
def whittaker_system(n=24, lmbd=100.0) -> np.ndarray:
    """src/preprocessing/whittaker_smoother.py:25-36: A = I + lmbd * D2' D2 (float64)."""
    D = np.zeros((n - 2, n))
    for i in range(n - 2):
        D[i, i:i + 3] = [1., -2., 1.]
    return np.eye(n) + lmbd * D.T @ D


def whittaker_monthly_matrix(n=24, lmbd=100.0, out=12) -> np.ndarray:
    """M = P @ inv(A): smooth (whittaker_smoother.py:38-47) then mean of each n/out
    consecutive steps (:64-67).  float64, [out, n]."""
    A = whittaker_system(n, lmbd)
    P = np.zeros((out, n))
    k = n // out
    for m in range(out):
        P[m, m * k:(m + 1) * k] = 1.0 / k
    return P @ np.linalg.inv(A)


def whittaker_interpolate(x24):
    """Smoother.interpolate_array (whittaker_smoother.py:44-69) on [24, X, Y, C]."""
    M = whittaker_monthly_matrix(x24.shape[0])
    return np.einsum('mg,gxyc->mxyc', M, x24.astype(np.float64)).astype(np.float32)


def temporal_operator(image_dates) -> np.ndarray:
    """W = M @ R  (12 x T, float64): regrid followed by Whittaker + monthly mean."""
    return whittaker_monthly_matrix() @ regrid_matrix(image_dates).astype(np.float64)


# --------------------------------------------------------------------------- missing data
def interpolate_na_vals(s2):
    """src/preprocessing/interpolation.py:42-56 (bn.median == NaN-propagating median)."""
    if np.sum(np.isnan(s2)) > 0:
        med = np.median(s2, axis=0).astype(np.float32)
        med[np.isnan(med)] = 0.
        for t in range(s2.shape[0]):
            nanv = np.isnan(s2[t])
            s2[t, nanv] = med[nanv]
    return s2


def id_missing_px(s2, thresh=11):
    """src/preprocessing/interpolation.py:5-23."""
    m0 = np.sum(s2[..., :10] == 0.0, axis=-1)
    mp = np.sum(s2[..., :10] >= 1., axis=-1)
    cnt = np.sum((m0 + mp) > 1., axis=(1, 2))
    return np.argwhere(cnt >= (s2.shape[1] ** 2) / thresh).flatten()


def interpolate_missing_vals(s2):
    """src/preprocessing/interpolation.py:26-39.  NOTE the reference's guard
    `logical_and(s2 >= 1, s2 == 0)` can never be true, so this is the identity."""
    return s2


def deal_w_missing_px(arr, dates, interp):
    """job.py:1031-1054 (the running median is recomputed after each in-place date fix)."""
    missing = id_missing_px(arr, 10)
    if len(missing) > 0:
        dates = np.delete(dates, missing)
        arr = np.delete(arr, missing, 0)
        interp = np.delete(interp, missing, 0)
    if np.sum(arr == 0) > 0:
        for i in range(arr.shape[0]):
            a = arr[i]
            a[a == 0] = np.median(arr, axis=0)[a == 0]
    if np.sum(arr == 1) > 0:
        for i in range(arr.shape[0]):
            a = arr[i]
            a[a == 1] = np.median(arr, axis=0)[a == 1]
    rm = np.argwhere(np.sum(np.isnan(arr), axis=(1, 2, 3)) > 0).flatten()
    if len(rm) > 0:
        dates = np.delete(dates, rm)
        arr = np.delete(arr, rm, 0)
        interp = np.delete(interp, rm, 0)
    return arr, dates, interp


def smooth_large_tile(arr, dates, interp):
    """job.py:1057-1096 (+ make_and_smooth_indices :1009-1028) -> [12, X, Y, 14]."""
    arr, dates, interp = deal_w_missing_px(arr, dates, interp)
    shp = (24, arr.shape[1], arr.shape[2])
    try:
        ind = regrid(make_indices(arr), dates)
    except Exception:
        ind = np.zeros(shp + (4,), dtype=np.float32)
    ind = whittaker_interpolate(ind)
    try:
        bands = regrid(arr, dates)
    except Exception:
        bands = np.zeros(shp + (arr.shape[-1],), dtype=np.float32)
    bands = whittaker_interpolate(bands)
    out = np.zeros(bands.shape[:3] + (14,), dtype=np.float32)
    out[..., :10] = bands
    out[..., 10:] = ind
    return out, dates, interp


# --------------------------------------------------------------------------- windows
def window_grid(H, W, size, n_rows=6, diff=7):
    """job.py:1295-1317 + src/tof/tof_downloading.py:498-524.

    Returns (folder[n,4], array[n,4]) int arrays (x, y, sx, sy): `folder` are the
    output windows, `array` the input windows grown by `diff` on interior sides.
    """
    gap_x = int(np.ceil((H - size) / (n_rows - 1)))
    gap_y = int(np.ceil((W - size) / (n_rows - 1)))
    xs = np.hstack([np.arange(0, H - size, gap_x), np.array(H - size)])
    ys = np.hstack([np.arange(0, W - size, gap_y), np.array(W - size)])
    folder = np.array([[x, y, size, size] for x in xs for y in ys], dtype=np.int64)
    nx, ny = len(xs), len(ys)
    arr = folder.copy()
    for i in range(len(arr)):
        ix, iy = i // ny, i % ny
        first_x, last_x = ix == 0, ix == nx - 1
        first_y, last_y = iy == 0, iy == ny - 1
        arr[i, 2] += diff if (first_x or last_x) else 2 * diff
        arr[i, 3] += diff if (first_y or last_y) else 2 * diff
        if not first_x:
            arr[i, 0] -= diff
        arr[i, 1] -= diff
    arr[arr < 0] = 0
    return folder, arr


def normalize_subtile(subtile, min_all=MIN_ALL, max_all=MAX_ALL):
    """job.py:316-325 (in place).  min/max are Python floats in the reference, i.e. "weak"
    scalars: the arithmetic is float32 with float32-rounded constants."""
    for b in range(subtile.shape[-1]):
        mn, mx = float(min_all[b]), float(max_all[b])
        subtile[..., b] = np.clip(subtile[..., b], mn, mx)
        subtile[..., b] = (subtile[..., b] - (mx + mn) / 2) / ((mx - mn) / 2)
    return subtile


def identify_bright_bare_surfaces(img):
    """job.py:1099-1122 on the un-normalised [L+1, W, W, 17] window."""
    r = (img[..., 3] / (img[..., 8] + 0.01)) < 0.9
    r = r * (np.mean(img[..., :3], axis=-1) > 0.2)
    r = r * (evi(img) < 0.3)
    bright = np.sum(r, axis=0) > 1
    bright = ndi.binary_dilation(1 - bright, iterations=2)
    bright = ndi.binary_dilation(1 - bright, iterations=1)
    blurred = ndi.distance_transform_edt(1 - bright)
    blurred[blurred > 3] = 3
    blurred = blurred / 3
    return blurred[7:-7, 7:-7]


def no_image_mask(min_clear, size):
    """job.py:1451-1472: [size+14]^2 clear-image count -> bool [size, size] or None."""
    mc = min_clear[6:-6, 6:-6]
    no = mc < 1
    s2 = ndi.generate_binary_structure(2, 2)
    no = 1 - ndi.binary_dilation(1 - no, structure=s2, iterations=6)
    no = ndi.binary_dilation(no, structure=s2, iterations=6)
    if size == 158:
        nb, bs, thr = 4, 40, 0.25
    elif size == 142:
        nb, bs, thr = 9, 16, 0.75
    else:
        return None
    no = np.reshape(no, (nb, bs, nb, bs))
    no = np.sum(no, axis=(1, 3)) > (bs * bs) * thr
    no = no.repeat(bs, axis=0).repeat(bs, axis=1)
    return no[1:-1, 1:-1]


def quarterly(s2_12, s1_12, length):
    """job.py:1274-1283."""
    if length == 4:
        s2 = np.median(np.reshape(s2_12, (4, 3) + s2_12.shape[1:]), axis=1)
        s1 = np.median(np.reshape(s1_12, (4, 3) + s1_12.shape[1:]), axis=1)
        return s2, s1
    if length == 1:
        s2 = np.repeat(np.median(s2_12, axis=0)[np.newaxis], 4, axis=0)
        s1 = np.repeat(np.median(s1_12, axis=0)[np.newaxis], 4, axis=0)
        return s2, s1
    return s2_12, s1_12


def tile_medians(s2):
    """job.py:1152-1160: [X, Y, 14] medians over dates of bands and of per-date indices."""
    med = np.median(s2, axis=0).astype(np.float32)
    for f in (evi, bi, msavi2, grndvi):
        med = np.concatenate([med, np.median(f(s2), axis=0)[..., np.newaxis]], axis=-1)
    return med


def process_subtiles(s2, dates, interp, s1, dem, predict_fn, size=158, length=4,
                     min_all=MIN_ALL, max_all=MAX_ALL, return_inputs=False):
    """job.py:1125-1483 numeric core.  Returns {(folder_y, folder_x): preds[size,size] f32}
    -- the arrays the reference would np.save to processed/{folder_y}/{folder_x}.npy.

    predict_fn(window[L+1, size+14, size+14, 17] normalised) -> [size, size] is the
    stand-in for predict_subtile(subtile, sess, op, size) (job.py:328-369).
    """
    s2 = interpolate_na_vals(s2)
    s2 = np.float32(s2)
    s2_median = tile_medians(s2)
    s2, dates, interp = smooth_large_tile(s2, dates, interp)
    s2_median = s2_median[np.newaxis]
    s1_median = np.median(s1, axis=0)[np.newaxis].astype(np.float32)
    s2, s1 = quarterly(s2, s1, length)
    L = s2.shape[0]
    folder, array = window_grid(s1.shape[1], s1.shape[2], size)
    out, feeds = {}, {}
    pad_u = pad_d = pad_l = pad_r = 0       # stale across iterations, as in the reference
    for t in range(len(folder)):
        sx, sy = array[t, 0], array[t, 1]
        fx, fy = folder[t, 0], folder[t, 1]
        ex, ey = sx + array[t, 2], sy + array[t, 3]
        sub = np.copy(s2[:, sx:ex, sy:ey, :])
        med2 = np.copy(s2_median[:, sx:ex, sy:ey, :])
        med1 = np.copy(s1_median[:, sx:ex, sy:ey, :])
        itile = interp[:, sx:ex, sy:ey]
        dsub = dem[np.newaxis, sx:ex, sy:ey]
        s1sub = np.copy(s1[:, sx:ex, sy:ey, :])
        min_clear = np.sum(itile < 0.33, axis=0)
        no_images = np.percentile(min_clear, 50) < 1
        if sub.shape[2] == size + 7:
            pad_u = 7 if sy == 0 else 0
            pad_d = 7 if sy != 0 else 0
            p4 = ((0, 0), (0, 0), (pad_u, pad_d), (0, 0))
            sub, s1sub = np.pad(sub, p4, 'reflect'), np.pad(s1sub, p4, 'reflect')
            med2, med1 = np.pad(med2, p4, 'reflect'), np.pad(med1, p4, 'reflect')
            dsub = np.pad(dsub, p4[:3], 'reflect')
            min_clear = np.pad(min_clear, ((0, 0), (pad_u, pad_d)), 'reflect')
        if sub.shape[1] == size + 7:
            pad_l = 7 if sx == 0 else 0
            pad_r = 7 if sx != 0 else 0
            p4 = ((0, 0), (pad_l, pad_r), (0, 0), (0, 0))
            sub, s1sub = np.pad(sub, p4, 'reflect'), np.pad(s1sub, p4, 'reflect')
            med2, med1 = np.pad(med2, p4, 'reflect'), np.pad(med1, p4, 'reflect')
            dsub = np.pad(dsub, p4[:3], 'reflect')
            # job.py:1395 pads with the (possibly stale) y-axis amounts -- replicated
            min_clear = np.pad(min_clear, ((pad_u, pad_d), (0, 0)), 'reflect')
        allb = np.zeros((L + 1, size + 14, size + 14, 17), dtype=np.float32)
        allb[:-1, ..., :10] = sub[..., :10]
        allb[:-1, ..., 11:13] = s1sub
        allb[:-1, ..., 13:] = sub[..., 10:]
        allb[:, ..., 10] = dsub.repeat(L + 1, axis=0)
        allb[-1, ..., :10] = med2[..., :10]
        allb[-1, ..., 11:13] = med1
        allb[-1, ..., 13:] = med2[..., 10:]
        bright = identify_bright_bare_surfaces(allb)
        if len(dates) < 2:
            no_images = True
        if no_images:
            preds = np.full((size, size), 255)
        else:
            allb = normalize_subtile(allb, min_all, max_all)
            if return_inputs:
                feeds[(int(fy), int(fx))] = allb.copy()
            preds = predict_fn(allb)
        mask = no_image_mask(min_clear, size)
        if mask is not None:
            preds[mask] = 255.
        preds = np.around(preds * bright, 3).astype(np.float32)
        out[(int(fy), int(fx))] = preds
    return (out, feeds) if return_inputs else out


def predict_subtile(subtile, model_fn, size):
    """job.py:328-369 with `model_fn(batch[1, L+1, W, W, 17]) -> [1, W-14, W-14, 1]`
    standing in for sess.run(op, feed_dict)."""
    if np.sum(subtile) != 0:
        if not isinstance(subtile.flat[0], np.floating):
            subtile = subtile / 65535.
        preds = model_fn(subtile[np.newaxis].astype(np.float32)).squeeze()
        clip = (preds.shape[0] - size) // 2
        if clip > 0:
            preds = preds[clip:-clip, clip:-clip]
        return np.float32(preds)
    return np.full((size, size), 255)


# --------------------------------------------------------------------------- mosaic
def fspecial_gauss(size, sigma):
    """job.py:1489-1501."""
    x, y = np.mgrid[-size // 2 + 1:size // 2 + 1, -size // 2 + 1:size // 2 + 1]
    return np.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))


def _calc_overlap(idx, tile, size):
    """job.py:1503-1512."""
    sub = tile[..., idx]
    others = np.delete(tile, idx, -1)
    others = others[~np.isnan(sub)].reshape((size, size, tile.shape[-1] - 1))
    rm = np.argwhere(np.sum(np.isnan(others), axis=(0, 1)) == (size * size)).flatten()
    with np.errstate(all='ignore'):
        others = np.nanmean(np.delete(others, rm, -1), axis=-1)
        sub = sub[~np.isnan(sub)].reshape((size, size))
        return np.nanmean(abs(others - sub))


def mosaic_predictions(windows: dict, size=158, sigma=36, return_float=False):
    """job.py:1515-1641 (depth == 1) from {(folder_y, folder_x): preds}.

    Returns uint8 [max_y_extent, max_x_extent] -- i.e. the raster TRANSPOSED relative
    to the [X, Y] tile arrays (job.py:1578).  With return_float also returns the
    float32 blend before quantisation (NaN = no data).
    """
    keys = sorted(windows.keys())
    xt = sorted(set(k[0] for k in keys))
    max_x = max(xt) + size
    max_y = max(k[1] for k in keys) + size
    n = len(keys)
    preds = np.full((max_x, max_y, n), np.nan, dtype=np.float32)
    mults = np.zeros((max_x, max_y, n), dtype=np.float32)
    for i, (a, b) in enumerate(keys):
        p = np.array(windows[(a, b)], copy=True)
        p[p < 255] = p[p < 255] * 100
        if np.sum(p) < size * size * 255:
            p = p.T.astype(np.float32)
            preds[a:a + size, b:b + size, i] = p
            g = fspecial_gauss(size, sigma)
            g[p > 100] = 0.
            mults[a:a + size, b:b + size, i] = g
    mults[np.isnan(preds)] = 0.
    try:
        ratios = np.zeros(n, dtype=np.float32)
        for i in range(n):
            ratios[i] = _calc_overlap(i, preds, size)
        with np.errstate(all='ignore'):
            multipliers = np.median(ratios) / ratios
        multipliers[multipliers > 1.5] = 1.5
        for i in range(n):
            mults[..., i] *= multipliers[i]
    except Exception:
        pass
    preds[preds > 100] = np.nan
    with np.errstate(all='ignore'):
        mults = mults / np.sum(mults, axis=-1)[..., np.newaxis]
    nan_cnt = np.sum(np.isnan(preds), axis=2)
    blend = np.nansum(preds * mults, axis=-1)
    blend[nan_cnt == n] = np.nan
    blend_f = blend.copy()
    blend[np.isnan(blend)] = 255.
    with np.errstate(all='ignore'):
        out = blend.astype(np.uint8)
    out[out <= 15] = 0
    out[out > 100] = 255
    no = ndi.binary_dilation(out == 255, structure=ndi.generate_binary_structure(2, 2),
                             iterations=10)
    out[no] = 255
    return (out, blend_f) if return_float else out


# --------------------------------------------------------------------------- 20 m -> 10 m
def mosaic_features(windows: dict, size, depth):
    """job.py:1515-1592, depth > 1 (feature export): {(outer, inner): int16 [size, size, depth]} -> int16
    [depth, max_outer + size, max_inner + size].  Gaussian weights without the no-data handling of depth 1."""
    keys = sorted(windows.keys())
    max_x = max(k[0] for k in keys) + size
    max_y = max(k[1] for k in keys) + size
    n = len(keys)
    out = np.zeros((depth, max_x, max_y), dtype=np.int16)
    for start in range(0, depth, 8):
        end = min(start + 8, depth)
        preds = np.full((end - start, max_x, max_y, n), np.nan, dtype=np.float32)
        mults = np.zeros((1, max_x, max_y, n), dtype=np.float32)
        for i, (a, b) in enumerate(keys):
            p = windows[(a, b)][..., start:end].T.astype(np.float32)
            preds[:, a:a + size, b:b + size, i] = p
            mults[:, a:a + size, b:b + size, i] = fspecial_gauss(size, 36)
        with np.errstate(all='ignore'):
            mults = mults / np.sum(mults, axis=-1)[..., np.newaxis]
            out[start:end] = np.int16(np.nansum(preds * mults, axis=-1))
    return out


def resize_bilinear(img, shape):
    """skimage.transform.resize(img, shape, order=1) with scikit-image's defaults (mode='reflect',
    anti_aliasing=True), as the reference calls it (job.py:741-743, :759-781; resegment_tiles_wide.py:1190-1236,
    :1354-1355):
      1. anti-aliasing: along every axis that SHRINKS (factor f = n_in / n_out > 1) the image is first smoothed with a
         Gaussian of sigma = (f - 1) / 2 (scipy.ndimage.gaussian_filter, truncate 4, numpy-pad 'reflect' == ndimage
         'mirror' borders), in the dtype of the input; axes that grow or keep their size get sigma 0 (no filter);
      2. bilinear sampling at pixel-centre aligned coordinates f * (i + 0.5) - 0.5 with mirrored edge samples
         == scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) in float64.
    PINNED (round 4) against the real scikit-image 0.18.3 on the reference's own shapes: tests/golden/resize.npz
    (tools/gen_golden_resize.py, run with /opt/conda/bin/python3.9)."""
    img = np.asarray(img)
    if img.dtype.kind != 'f':
        img = img.astype(np.float64)
    factors = [i / o for o, i in zip(shape, img.shape)]
    sigma = [max(0.0, (f - 1.0) / 2.0) for f in factors]
    if any(s > 0 for s in sigma):
        img = ndi.gaussian_filter(img, sigma, mode='mirror')
    zoom = [o / i for o, i in zip(shape, img.shape)]
    return ndi.zoom(img.astype(np.float64), zoom, order=1, mode='mirror', grid_mode=True)


def upsample_20m(s2_10, s2_20):
    """job.py:734-782: [T,2h,2w,4] + [T,h,w,6] -> [T,2h,2w,10].  The 40 m bands (20 m indices 4, 5)
    are 2x2-averaged first; on odd 20 m grids (309 for a 618 tile!) the first row / column is set
    aside and written back nearest-replicated (:760-782)."""
    T, w, h = s2_10.shape[0], s2_20.shape[1] * 2, s2_20.shape[2] * 2
    out = np.zeros((T, w, h, 10), np.float32)
    out[..., :4] = s2_10
    for band in range(4):
        for t in range(T):
            out[t, ..., band + 4] = resize_bilinear(s2_20[t, ..., band], (w, h))

    def mean4(m):
        return np.mean(m.reshape(m.shape[0] // 2, 2, m.shape[1] // 2, 2), axis=(1, 3))

    for band in range(4, 6):
        for t in range(T):
            mid = s2_20[t, ..., band]
            ox, oy = mid.shape[0] % 2, mid.shape[1] % 2
            row0, col0 = mid[0, :], mid[:, 0]
            out[t, ox:, oy:, band + 4] = resize_bilinear(mean4(mid[ox:, oy:]), (w - ox, h - oy))
            if ox:
                out[t, 0, :, band + 4] = row0.repeat(2)
            if oy:
                out[t, :, 0, band + 4] = col0.repeat(2)
    return out


def superresolve_large_tile(arr, dsen2_fn, wsize=110):
    """job.py:95-147 driver.  dsen2_fn(padded[T,118,118,10], bilinear[T,118,118,6]) ->
    [T,118,118,6] stands in for sess.run(superresolve_logits).  Replicates the
    unreachable third branch (windows y==last, x!=last are never refined) and the
    in-place aliasing of `x_end`.
    """
    def worker(a):
        pad = np.pad(a, ((0, 0), (4, 4), (4, 4), (0, 0)), 'reflect')
        res = dsen2_fn(pad, pad[..., 4:])
        a[..., 4:] = res[:, 4:-4, 4:-4, :]
        return a

    xr = [x for x in range(0, arr.shape[1] - wsize, wsize)] + [arr.shape[1] - wsize]
    yr = [y for y in range(0, arr.shape[2] - wsize, wsize)] + [arr.shape[2] - wsize]
    x_end = np.copy(arr[:, xr[-1]:, ...])
    for x in xr:
        for y in yr:
            if x != xr[-1] and y != yr[-1]:
                arr[:, x:x + wsize, y:y + wsize, ...] = worker(arr[:, x:x + wsize, y:y + wsize, ...])
            elif x == xr[-1]:
                arr[:, x:x + wsize, y:y + wsize, ...] = worker(x_end[:, :, y:y + wsize, ...])
    return arr
