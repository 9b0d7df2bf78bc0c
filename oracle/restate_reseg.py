"""CPU oracle for the tile-border resegmentation (src/resegment_tiles_wide.py): the numeric core of `resegment_border`
(:847-1161) from the arrays `process_tile` returns for a tile and its right-hand neighbour, the border re-prediction
`process_subtiles` (:360-616) and the border-aware mosaic `recreate_resegmented_tifs` / `mosaic_subtiles` (:1169-1549).
S3 / hickle / GeoTIFF IO, the tile database and the ARD update (`update_ard_tiles`, :793-844) are out of scope.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/test_oracle_reseg.py against golden vectors captured by
running the reference's own functions (tools/gen_golden.py, reseg_*.npz).  `skimage.transform.resize` is absent from the
survey image, so the weight maps of the border mosaic carry the same "parity unpinned" caveat as row a3:
both the golden generator and this file use ndimage.zoom(order=1, mode='mirror', grid_mode=True) for it."""
from __future__ import annotations

import numpy as np

from oracle import restate_clouds as C
from oracle import restate_gapfill as G
from oracle import restate_numpy as R

SIZE = 670        # resegment_tiles_wide.py:1598
SIZE_Y = 206      # :1599
LEN = 4           # :41

# :1664-1685 -- float32 vectors here (the main job keeps Python floats, job.py:1829-1842)
MIN_ALL32 = np.asarray(R.MIN_ALL, dtype=np.float64).astype(np.float32)
MAX_ALL32 = np.array(R.MAX_ALL, dtype=np.float64, copy=True)
MAX_ALL32[10] = 0.509269855802243     # :1675 (the DEM maximum differs from job.py:1839's 0.4)
MAX_ALL32 = MAX_ALL32.astype(np.float32)


def align_dates(tile_date, neighb_date):
    """:238-257 -- dates further than 1 day from every date of the other tile, and repeated dates, are dropped."""
    tile_date, neighb_date = np.asarray(tile_date), np.asarray(neighb_date)
    d_t = [np.min(abs(a - neighb_date)) for a in tile_date]
    d_n = [np.min(abs(a - tile_date)) for a in neighb_date]
    dup_t = np.argwhere(np.diff(tile_date, prepend=0) == 0).flatten()
    dup_n = np.argwhere(np.diff(neighb_date, prepend=0) == 0).flatten()
    rm_t = [i for i, d in enumerate(d_t) if d > 1] + list(dup_t)
    rm_n = [i for i, d in enumerate(d_n) if d > 1] + list(dup_n)
    left = np.minimum(len(tile_date) - len(rm_t), len(neighb_date) - len(rm_n))
    return rm_t, rm_n, left


def split_fn(item, form, size=SIZE):
    """:84-93 -- the tile keeps its last size//2 + 7 columns (axis 2), the neighbour its first size//2 + 7."""
    tiles_x = None
    if form == 'tile':
        overlap_left = item.shape[2] - (size // 2) - 7
        tiles_x = overlap_left + 7
        item = item[:, :, overlap_left:]
    if form == 'neighbor':
        item = item[:, :, :(size // 2) + 7]
    return item, tiles_x


def split_to_border(s2, interp, s1, dem, fname, size=SIZE):
    """:105-115 (edge == 'right').  The 5th value is the `_` of the last split: for the tile, the column at which its
    half of the strip starts (callers use it as tiles_folder_x); None for the neighbour."""
    s1, _ = split_fn(s1, fname, size)
    interp, _ = split_fn(interp, fname, size)
    s2, _ = split_fn(s2, fname, size)
    dem, last = split_fn(dem[np.newaxis], fname, size)
    return s2, interp, s1, dem.squeeze(), last


def _cartesian(*arrays):
    mesh = np.meshgrid(*arrays)
    return np.reshape(np.concatenate(mesh).ravel(), (len(mesh), mesh[0].size)).T


def make_tiles_right_neighb(tiles_folder_x, tiles_folder_y, size=SIZE, size_y=SIZE_Y):
    """:267-281"""
    windows = _cartesian(tiles_folder_x, tiles_folder_y)
    tiles_folder = np.hstack([windows, np.full_like(windows, size + 7)])
    tiles_folder = np.sort(tiles_folder, axis=0)
    uy = np.unique(tiles_folder[:, 1])
    tiles_folder[:, 1] = np.tile(uy, int(len(tiles_folder[:, 1]) / len(uy)))
    tiles_array = np.copy(tiles_folder)
    tiles_array[1:, 1] -= 7
    tiles_array[:, 0] = 0.
    tiles_array[:, 2] = size + 14.
    tiles_array[:, 3] = size_y + 7.
    tiles_array[1:-1, 3] += 7
    return tiles_array, tiles_folder


def border_window_table(n_rows, size=SIZE, size_y=SIZE_Y, tiles_folder_x=0):
    """:1135-1138 -- four windows stacked along axis 1 of the strip."""
    gap_y = int(np.ceil((n_rows - size_y) / 3))
    tiles_folder_y = np.hstack([np.arange(0, n_rows - size_y, gap_y), np.array(n_rows - size_y)])
    return make_tiles_right_neighb(tiles_folder_x, tiles_folder_y, size, size_y)


def align_subtile_histograms(array, size=SIZE):
    """:284-343, in place.  Per date: band means / stds of the non-water pixels of the two halves (split at column
    (size+14)//2) are moved onto their average; kept only when the seam step at that column shrinks.  Note the halves'
    names are crossed in the reference (columns [half:] are `left`): columns [:half] get the statistics of [half:]."""
    half = (size + 14) // 2

    def ndwi(a):
        with np.errstate(all='ignore'):
            return (a[..., 1] - a[..., 3]) / (a[..., 1] + a[..., 3])

    left_water = ndwi(np.median(array[:, :, half:], axis=0)) >= 0.1
    right_water = ndwi(np.median(array[:, :, :half], axis=0)) >= 0.1
    for t in range(array.shape[0]):
        left, right = array[t, :, half:], array[t, :, :half]
        with np.errstate(all='ignore'):
            std_right = np.nanstd(right[~right_water], axis=0)
            std_left = np.nanstd(left[~left_water], axis=0)
            std_ref = (std_right + std_left) / 2
            mean_right = np.nanmean(right[~right_water], axis=0)
            mean_left = np.nanmean(left[~left_water], axis=0)
            mean_ref = (mean_right + mean_left) / 2
            mult_l = std_left / std_ref
            add_l = mean_left - mean_ref * mult_l
            mult_r = std_right / std_ref
            add_r = mean_right - mean_ref * mult_r
            before = abs(np.roll(array[t], 1, axis=1) - array[t])
            before = np.mean(before[:, (size // 2) + 7, :], axis=(0, 1))
            cand = np.copy(array[t])
            cand[:, :half, :] = cand[:, :half, :] * mult_l + add_l
            cand[:, half:, :] = cand[:, half:, :] * mult_r + add_r
            after = abs(np.roll(cand, 1, axis=1) - cand)
            after = np.mean(after[:, (size // 2) + 7, :], axis=(0, 1))
        if after < before:
            array[t, :, :half, :] = array[t, :, :half, :] * mult_l + add_l
            array[t, :, half:, :] = array[t, :, half:, :] * mult_r + add_r
    return array


def predict_subtile(subtile, model_fn, size=SIZE, size_y=SIZE_Y):
    """:182-222 -- float32 mid-range / range normalisation; all-zero window -> 255 fill ((SIZE, SIZE) as coded)."""
    if np.sum(subtile) != 0:
        if not isinstance(subtile.flat[0], np.floating):
            subtile = subtile / 65535.
        mn, mx = MIN_ALL32.reshape(1, 1, 1, 17), MAX_ALL32.reshape(1, 1, 1, 17)
        mid = ((mx + mn) / 2).astype(np.float32)
        rng = (mx - mn).astype(np.float32)
        x = np.clip(subtile, mn, mx)
        x = (x - mid) / (rng / 2)
        return np.asarray(model_fn(x[np.newaxis])).squeeze(), x
    return np.full((size, size), 255), None


def seam_adjust(preds, size=SIZE):
    """:518-531 -- halves pulled together when the 4-column means either side of the seam differ by > 0.15"""
    left_mean = np.mean(preds[:, (size - 8) // 2: size // 2])
    right_mean = np.mean(preds[:, size // 2: (size + 8) // 2])
    if abs(left_mean - right_mean) > 0.15:
        left, right = preds[:, :size // 2], preds[:, size // 2:]
        left_mean = np.mean(left[left > 0.05])
        right_mean = np.mean(right[right > 0.05])
        adj = (right_mean - left_mean) / 2
        left[left > 0.05] += adj
        right[right > 0.05] -= adj
        preds = np.clip(preds, 0, 1)
    return preds


def keep_decision(preds, left_all, right_all, start_y, size=SIZE, size_y=SIZE_Y):
    """:534-613 -- whether the window is written (all but the final `else`)."""
    if not np.max(preds) < 255:
        return True
    with np.errstate(all='ignore'):
        left_source = np.nanmean(left_all[start_y:start_y + size_y, :100])
        right_source = np.nanmean(right_all[start_y:start_y + size_y, -100:])
        lo, hi = np.minimum(left_source, right_source), np.maximum(left_source, right_source)
        src = 100 * np.nanmean(preds)
    if src <= lo - 15 or src >= hi + 15:
        return True
    if src <= hi + 15 and src >= lo - 15:
        return True
    return bool(np.isnan(hi) and np.isnan(lo))


def process_border_subtiles(s2, dates, interp, s1, dem, model_fn, tiles_folder, tiles_array, right_all, left_all,
                            hist_align, min_clear, size=SIZE, size_y=SIZE_Y, length=LEN, trace=None):
    """:360-616 -> list of dicts {folder_x, folder_y, preds, saved}; `trace[t]` receives the normalised model feed.
    s2 [12, X, size+14, 14] (10 bands + 4 indices), s1 [12, X, size+14, 2], dem [X, size+14]."""
    s2 = np.float32(R.interpolate_na_vals(s2))
    s2_median = np.median(s2, axis=0)[np.newaxis]
    s1_median = np.median(s1, axis=0)[np.newaxis]
    if length == 4:
        s2 = np.median(np.reshape(s2, (4, 3) + s2.shape[1:]), axis=1)
        s1 = np.median(np.reshape(s1, (4, 3) + s1.shape[1:]), axis=1)
    out = []
    for t in range(len(tiles_folder)):
        start_x, start_y = int(tiles_array[t][0]), int(tiles_array[t][1])
        folder_x, folder_y = tiles_folder[t][1], tiles_folder[t][0]
        end_x, end_y = start_x + int(tiles_array[t][2]), start_y + int(tiles_array[t][3])
        subset = np.copy(s2[:, start_y:end_y, start_x:end_x, :])
        med_s2 = np.copy(s2_median[:, start_y:end_y, start_x:end_x, :])
        med_s1 = s1_median[:, start_y:end_y, start_x:end_x, :]
        dates_tile = np.copy(dates)
        dem_sub = dem[np.newaxis, start_y:end_y, start_x:end_x]
        s1_sub = s1[:, start_y:end_y, start_x:end_x, :]
        to_remove = np.argwhere(np.sum(np.isnan(subset), axis=(1, 2, 3)) > 0).flatten()
        if len(to_remove) > 0:
            dates_tile = np.delete(dates_tile, to_remove)
            subset = np.delete(subset, to_remove, 0)
        subtile = subset
        if hist_align:
            subset = align_subtile_histograms(subset, size)
            med_s2 = align_subtile_histograms(med_s2, size)
        if subtile.shape[2] == size + 7:
            pu, pd = (7 if start_y != 0 else 0), (7 if start_y == 0 else 0)
            pad4, pad3 = ((0, 0), (0, 0), (pu, pd), (0, 0)), ((0, 0), (0, 0), (pu, pd))
            subtile, s1_sub = np.pad(subtile, pad4, 'reflect'), np.pad(s1_sub, pad4, 'reflect')
            dem_sub = np.pad(dem_sub, pad3, 'reflect')
            med_s2, med_s1 = np.pad(med_s2, pad4, 'reflect'), np.pad(med_s1, pad4, 'reflect')
        if subtile.shape[1] == size_y + 7:
            pl, pr = (7 if start_y == 0 else 0), (7 if start_y != 0 else 0)
            pad4, pad3 = ((0, 0), (pl, pr), (0, 0), (0, 0)), ((0, 0), (pl, pr), (0, 0))
            subtile, s1_sub = np.pad(subtile, pad4, 'reflect'), np.pad(s1_sub, pad4, 'reflect')
            dem_sub = np.pad(dem_sub, pad3, 'reflect')
            med_s2, med_s1 = np.pad(med_s2, pad4, 'reflect'), np.pad(med_s1, pad4, 'reflect')
        full = np.empty((length + 1, size_y + 14, size + 14, 17), dtype=np.float32)
        full[:-1, ..., :10] = subtile[..., :10]
        full[:-1, ..., 11:13] = s1_sub
        full[:-1, ..., 13:] = subtile[..., 10:]
        full[:, ..., 10] = dem_sub.repeat(length + 1, axis=0)
        full[-1, ..., :10] = med_s2[..., :10]
        full[-1, ..., 11:13] = med_s1
        full[-1, ..., 13:] = med_s2[..., 10:]
        if len(dates_tile) < 2:
            preds = np.full((size_y, size), 255)
        else:
            preds, feed = predict_subtile(full, model_fn, size, size_y)
            if trace is not None and feed is not None:
                trace[t] = feed
        preds = seam_adjust(preds, size)
        saved = keep_decision(preds, left_all, right_all, start_y, size, size_y)
        out.append(dict(folder_x=int(folder_x), folder_y=int(folder_y), preds=preds, saved=saved))
    return out


def check_if_artifact(tile, neighb):
    """:675-710 -- is there a visible seam between the last column of `tile` and the first of `neighb` (0-100, NaN)?"""
    with np.errstate(all='ignore'):
        right_mean = np.nanmean(neighb[:, :3])
        left_mean = np.nanmean(tile[:, -3:])

        def blocks(v):      # all-NaN blocks give NaN (and numpy's "mean of empty slice" warning), as in the reference
            v = np.pad(v, (10 - (v.shape[0] % 10)) // 2, constant_values=np.nan)
            return np.nanmean(np.reshape(v, (v.shape[0] // 10, 10)), axis=1)
        right, left = blocks(neighb[:, 0]), blocks(tile[:, -1])
        f20 = np.nanmean(abs(right - left) > 20)
        f125 = np.nanmean(abs(right - left) > 12.5)
        f_l = np.nanmean(abs(right[:15] - left[:15]) > 17.5)
        f_r = np.nanmean(abs(right[-15:] - left[-15:]) > 17.5)
    lr = abs(right_mean - left_mean)
    other0 = lr > 6
    other = np.logical_and(f125 > 0.5, lr > 1)
    other2 = np.logical_and((f20 > 0.3) or (f_l > 0.5) or (f_r > 0.5), lr > 1)
    return 1 if (other0 or other or other2) else 0


def adjust_resegment(res, mults, n):
    """:1164-1166"""
    return res * np.maximum(np.sum(mults[..., :n], axis=-1), 1.)


def mosaic_subtiles(preds, mults, na, kind, left, right, up, down, size=SIZE):
    """:1169-1237 -- weighted mean over the stack, and the ramp that weights this stack against the others."""
    na = np.tile(na, (1, 1, preds.shape[-1]))
    preds[na > 0] = np.nan
    mults[np.isnan(preds)] = 0.
    with np.errstate(all='ignore'):
        mults = mults / np.sum(mults, axis=-1)[..., np.newaxis]
        preds = np.nansum(preds * mults, axis=-1)
    m = (np.arange(0, 300, 1) / 300) ** 1.33
    border = np.tile(m, (size // 2, 1))
    mult_arr = (np.arange(0, size // 2, 1) / (size // 2))[:, np.newaxis]
    mult_arr = (np.ones((size // 2, preds.shape[1])) * mult_arr) ** 1.2
    left, right, up, down = (v is not None for v in (left, right, up, down))
    zero_arr = np.zeros((preds.shape[0] - (size // 2), preds.shape[1]))
    rs = R.resize_bilinear
    if kind == 'n':
        m = R.fspecial_gauss(preds.shape[0], preds.shape[0] / 5.25)
        m = rs(m, (preds.shape[0], preds.shape[1]))
    if kind == 'r':
        m = np.copy(mult_arr)
        if up:
            m[:, :300] *= border
        if down:
            m[:, -300:] *= np.fliplr(border)
        m = rs(m, (size // 2, preds.shape[1]))
        m = np.concatenate([zero_arr, m], axis=0)
        m = rs(m, (preds.shape[0], preds.shape[1]))
    if kind == 'l':
        m = np.flipud(np.copy(mult_arr))
        if up:
            m[:, :300] *= border
        if down:
            m[:, -300:] *= np.fliplr(border)
        m = rs(m, (size // 2, preds.shape[1]))
        m = np.concatenate([m, zero_arr], axis=0)
        m = rs(m, (preds.shape[0], preds.shape[1]))
    if kind == 'u':
        m = np.flipud(np.copy(mult_arr))
        if left:
            m[:, :300] *= border
        if right:
            m[:, -300:] *= np.fliplr(border)
        m = rs(m, (size // 2, preds.shape[1]))
        m = np.concatenate([m, zero_arr], axis=0)
        m = m.T
        m = rs(m, (preds.shape[0], preds.shape[1]))
    if kind == 'd':
        m = np.copy(mult_arr)
        if right:
            m[:, :300] *= border
        if left:
            m[:, -300:] *= np.fliplr(border)
        m = rs(m, (size // 2, preds.shape[1]))
        m = np.concatenate([zero_arr, m], axis=0)
        m = np.flipud(m.T)
        m = rs(m, (preds.shape[0], preds.shape[1]))
    m[np.isnan(preds)] = 0.
    return preds, m


def _fspecial_size(subtile_size, border):
    """:1303-1313 (normal windows) and :1338-1347 (border windows)"""
    if subtile_size in (208, 216):
        return 44
    if subtile_size == 348:
        return 85
    if subtile_size == 412:
        return 95
    if not border and subtile_size == 168:
        return 38
    if border and (subtile_size == 588 or subtile_size >= 620):
        return 150
    return 28


def recreate_resegmented(windows, shape, size=SIZE):
    """:1240-1549 with the directory listing replaced by `windows`: a list of (kind, x_tile, y_tile, prediction) in
    the listing order, kind in {'n', 'l', 'r', 'u', 'd'} = plain `{x}/{y}.npy`, `{x}/left{y}.npy`, `right{x}/{y}.npy`,
    `{x}/up{y}.npy`, `{x}/down{y}.npy`.  -> (preds float [shape[1], shape[0]], 255 = no data; sums)."""
    by = {k: [w for w in windows if w[0] == k] for k in 'nlrud'}
    X, Y = shape[1], shape[0]
    pn = np.full((X, Y, len(by['n'])), np.nan, dtype=np.float32)
    mn = np.full((X, Y, len(by['n'])), 0, dtype=np.float32)
    sum_normal, sum_normal_na = np.zeros((X, Y)), np.zeros((X, Y))
    sum_reseg, sum_reseg_na = np.zeros((X, Y)), np.zeros((X, Y))
    i = 0
    for _, xt, yt, pred in by['n']:
        sy, sx = pred.shape
        sub = np.maximum(sx, sy)
        if np.sum(pred) < sx * sy * 255:
            p = (pred * 100).T.astype(np.float32)
            if (xt + sx - 1) < X and (yt + sy - 1) < Y:
                pn[xt:xt + sx, yt:yt + sy, i] = p
                f = R.fspecial_gauss(sub, _fspecial_size(sub, False))
                f[p > 100] = 0.
                mn[xt:xt + sx, yt:yt + sy, i] = f
                cnt = np.ones_like(p)
                sum_normal[xt:xt + sx, yt:yt + sy] += cnt
                cnt[p <= 100] = 0.
                sum_normal_na[xt:xt + sx, yt:yt + sy] += cnt
            i += 1

    def border_block(kind):
        lst = by[kind]
        if not lst:
            return None, None
        P = np.full((X, Y, len(lst)), np.nan, dtype=np.float32)
        M = np.full((X, Y, len(lst)), 0, dtype=np.float32)
        for j, (_, xt, yt, pred) in enumerate(lst):
            if kind in 'lr':
                sy, sx = pred.shape[0], pred.shape[1] // 2
                sub = np.maximum(sx * 2, sy)
            else:
                sy, sx = pred.shape[0] // 2, pred.shape[1]
                sub = np.maximum(sx, sy * 2)
            f = R.fspecial_gauss(sub, _fspecial_size(sub, True))
            f = {'l': f[sx:, :], 'r': f[:sx, :], 'u': f[:, sy:], 'd': f[:, :sy]}[kind]
            f = R.resize_bilinear(f, (sx, sy))
            if np.sum(pred) < sx * sy * 255:
                p = (pred * 100).T.astype(np.float32)
                p = {'l': p[sx:, :], 'r': p[:sx, :], 'u': p[:, sy:], 'd': p[:, :sy]}[kind]
                P[xt:xt + sx, yt:yt + sy, j] = p
                adj = P[xt:xt + sx, yt:yt + sy, :]
                adj[p > 100] = 255.
                f[p > 100] = 0.
                M[xt:xt + sx, yt:yt + sy, j] = f
                cnt = np.ones_like(p)
                sum_reseg[xt:xt + sx, yt:yt + sy] += cnt
                cnt[p <= 100] = 0.
                sum_reseg_na[xt:xt + sx, yt:yt + sy] += cnt
        return P, M

    pl, ml = border_block('l')
    pr, mr = border_block('r')
    pu, mu = border_block('u')
    pd, md = border_block('d')
    isnan = np.zeros_like(sum_reseg)
    isnan[(sum_reseg == 0) * ((sum_normal - sum_normal_na) == 0)] = 1.
    isnan[(sum_reseg > 0) * (np.logical_or(((sum_normal - sum_normal_na) == 0), (sum_reseg_na > 0)))] = 1.
    isnan = isnan[..., np.newaxis]
    preds_n, mults_n = mosaic_subtiles(pn, mn, isnan, 'n', pl, pr, pu, pd, size)
    parts = []
    for kind, P, M in (('r', pr, mr), ('l', pl, ml), ('u', pu, mu), ('d', pd, md)):
        if P is not None:
            p_k, m_k = mosaic_subtiles(P, M, isnan, kind, pl, pr, pu, pd, size)     # NaNs `P` in place where na > 0
            m_k[np.sum(~np.isnan(P), axis=-1) == 0] = 0.
        else:
            p_k, m_k = np.zeros_like(preds_n), np.zeros_like(preds_n)
        parts.append((p_k, m_k))
    (p_r, m_r), (p_l, m_l), (p_u, m_u), (p_d, m_d) = parts
    with np.errstate(all='ignore'):
        sums = m_l + m_r + m_u + m_d + mults_n
        preds = (p_l * (m_l / sums)) + (p_d * (m_d / sums))
        preds = preds + (p_r * (m_r / sums)) + (p_u * (m_u / sums))
        preds = preds + (preds_n * (mults_n / sums))
    preds[np.isnan(preds)] = 255.
    preds[isnan.squeeze() == 1.] = 255.
    return preds, sums


# ---- strip preparation: resegment_border, :847-1161 ---------------------------------------------------------------------
def preprocess_tile(arr, dates, interp, clm, dem, sampler=G.reference_sampler):
    """:619-672 -> (arr, interp, dates).  `interp` is ignored, as in the reference (it is recomputed)."""
    missing = R.id_missing_px(arr, 20)
    if len(missing) > 0:
        dates = np.delete(dates, missing)
        arr = np.delete(arr, missing, 0)
    cld, fcps = C.identify_clouds_shadows(arr, dem)
    if clm is not None:
        if len(missing) > 0:
            clm = np.delete(clm, missing, 0)
        try:
            clm[fcps] = 0.
            cld = np.maximum(clm, cld)
        except Exception:           # date mismatch between the Sen2Cor mask and the stack: the reference carries on
            pass
    interp = G.id_areas_to_interp(cld)
    to_remove = np.argwhere(np.mean(interp == 1, axis=(1, 2)) > 0.95)
    if len(to_remove) > 0:
        cld = np.delete(cld, to_remove, axis=0)
        dates = np.delete(dates, to_remove)
        arr = np.delete(arr, to_remove, axis=0)
        cld, fcps = C.identify_clouds_shadows(arr, dem)
    arr, interp2, _ = G.remove_cloud_and_shadows(arr, cld, fcps, sampler=sampler)
    return arr, interp2, dates


def regularize_and_smooth(arr, dates):
    """:772-790 -- date regrid + Whittaker per band pair -> 12 monthly steps"""
    n = arr.shape[0]
    if n < 12:
        arr = np.concatenate([arr, np.zeros((12 - n,) + arr.shape[1:], dtype=np.float32)], axis=0)
    for j in range(0, 10, 2):
        arr[:12, ..., j:j + 2] = R.whittaker_interpolate(R.regrid(arr[:n, ..., j:j + 2], dates))
    return arr[:12]


def make_and_smooth_indices(arr, dates):
    """job.py:1009-1028"""
    try:
        ind = R.regrid(R.make_indices(arr), dates)
    except Exception:
        ind = np.zeros((24, arr.shape[1], arr.shape[2], 4), dtype=np.float32)
    return R.whittaker_interpolate(ind)


def resegment_border_arrays(tile, neighb, tile_tif, neighbor_tif, model_fn, dsen2_fn, min_dates=2, size=SIZE, size_y=SIZE_Y,
                            sampler=G.reference_sampler, trace=None):
    """The array flow of resegment_border (:847-1161, edge == "right") once both tiles are known to be processed and an
    artifact was found: `tile` / `neighb` = dicts with the outputs of process_tile (s2 [T, X, Y, 10], dates, interp, s1
    [12, X, Y, 2], dem [X, Y]) and optionally clm (Sen2Cor mask at 10 m, [T, X, Y]).  tile_tif / neighbor_tif: the existing
    rasters (float, NaN above 100).  -> (windows of process_border_subtiles, info dict)"""
    cp = lambda d: {k: (np.array(v, copy=True) if v is not None else None) for k, v in d.items()}        # noqa: E731
    a, b = cp(tile), cp(neighb)
    _, _, min_images = align_dates(a["dates"], b["dates"])
    if min_images >= 3:
        s2, interp, s1, dem, tiles_x = split_to_border(a["s2"], a["interp"], a["s1"], a["dem"], "tile", size)
        s2n, interp_n, s1n, dem_n, _ = split_to_border(b["s2"], b["interp"], b["s1"], b["dem"], "neighbor", size)
        dates, dates_n = a["dates"], b["dates"]
        clm = split_fn(a["clm"], 'tile', size)[0] if a.get("clm") is not None else None
        clm_n = split_fn(b["clm"], 'neighbor', size)[0] if b.get("clm") is not None else None
        rm_t, rm_n, _ = align_dates(dates, dates_n)
        if len(rm_t) > 0:
            s2, dates = np.delete(s2, rm_t, 0), np.delete(dates, rm_t)
            if clm is not None and clm.shape[0] > 0:
                clm = np.delete(clm, rm_t, 0)
        if len(rm_n) > 0:
            s2n, dates_n = np.delete(s2n, rm_n, 0), np.delete(dates_n, rm_n)
            if clm_n is not None and clm_n.shape[0] > 0:
                clm_n = np.delete(clm_n, rm_n, 0)
        if clm is not None and clm_n is not None:
            try:
                clm = np.float32(np.concatenate([clm_n, clm], axis=2))      # neighbour first, as coded (:971)
                clm[np.isnan(clm)] = 0.
            except Exception:
                clm = None
        else:
            clm = None
        s2 = np.concatenate([s2, s2n], axis=2).astype(np.float32)
        dem = np.concatenate([dem, dem_n], axis=1)
        s2, interp, dates = preprocess_tile(s2, dates, None, clm, dem, sampler)
        s2, dates, interp = R.deal_w_missing_px(s2, dates, interp)
        dates_n = dates
        indices = make_and_smooth_indices(s2, dates)
        s2 = regularize_and_smooth(s2, dates)
        min_clear = np.sum(interp != 1, axis=0)
    else:
        s2, interp, dates = preprocess_tile(a["s2"], a["dates"], a["interp"], a.get("clm"), a["dem"], sampler)
        s2, interp, s1, dem, tiles_x = split_to_border(s2, interp, a["s1"], a["dem"], "tile", size)
        s2n, interp_n, dates_n = preprocess_tile(b["s2"], b["dates"], b["interp"], b.get("clm"), b["dem"], sampler)
        s2n, interp_n, s1n, dem_n, _ = split_to_border(s2n, interp_n, b["s1"], b["dem"], "neighbor", size)
        rm_t, rm_n, min_images = align_dates(dates, dates_n)
        half = (size + 14) // 2
        min_clear = np.concatenate([np.sum(interp[..., -half:] != 1, axis=0), np.sum(interp_n[..., :half] != 1, axis=0)], axis=1)
        if min_images >= min_dates:
            if len(rm_t) > 0:
                s2, interp, dates = np.delete(s2, rm_t, 0), np.delete(interp, rm_t, 0), np.delete(dates, rm_t)
            if len(rm_n) > 0:
                s2n, interp_n, dates_n = np.delete(s2n, rm_n, 0), np.delete(interp_n, rm_n, 0), np.delete(dates_n, rm_n)
        s2, dates, interp = R.deal_w_missing_px(s2, dates, interp)
        ind = make_and_smooth_indices(s2, dates)
        s2 = regularize_and_smooth(s2, dates)
        s2n, dates_n, interp_n = R.deal_w_missing_px(s2n, dates_n, interp_n)
        ind_n = make_and_smooth_indices(s2n, dates_n)
        s2n = regularize_and_smooth(s2n, dates_n)
        s2 = np.concatenate([s2, s2n], axis=2).astype(np.float32)
        indices = np.concatenate([ind, ind_n], axis=2)
        dem = np.concatenate([dem, dem_n], axis=1)
        interp = np.concatenate([interp[:interp_n.shape[0]], interp_n[:interp.shape[0]]], axis=2)
    s1 = np.concatenate([s1, s1n], axis=2)
    s2 = R.superresolve_large_tile(s2, dsen2_fn, wsize=125)                   # :144-179
    out = np.empty(s2.shape[:3] + (14,), dtype=np.float32)
    out[..., :10] = s2
    out[..., 10:] = indices
    ta, tf = border_window_table(s1.shape[1], size, size_y, tiles_folder_x=tiles_x)
    hist_align = not np.array_equal(np.array(dates), np.array(dates_n))
    right_all = neighbor_tif[:, :size // 2]
    left_all = tile_tif[:, -(size // 2):]
    if trace is not None:
        trace.update(strip=out.copy(), dates=np.array(dates), dates_n=np.array(dates_n), hist_align=hist_align, interp=interp.copy(),
                     min_clear=min_clear.copy(), s1=s1.copy(), dem=dem.copy())
    wins = process_border_subtiles(out, dates, interp, s1, dem, model_fn, tf, ta, right_all, left_all, hist_align, min_clear,
                                   size=size, size_y=size_y)
    return wins, dict(min_images=int(min_images), hist_align=hist_align, tiles_array=ta, tiles_folder=tf)
