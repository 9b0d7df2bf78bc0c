"""CPU oracle for the multi-temporal cloud / shadow DETECTION (the step just before the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, from src/preprocessing/cloud_removal.py (`CR.py` below):
  identify_clouds_shadows   CR.py:1215-1677
  detect_pfcp               CR.py:1109-1212
The two ESA-WorldCover rasters the reference reads with rasterio (CR.py:735-771) are inputs here:
  forest   [H, W]  == adjust_cloudmask_in_forests(...)            (None -> zeros, the reference's except branch)
  urban    (core [H, W], near [H, W]) == the two resized masks of mask_nonurban_areas (None -> no urban pixels)
Pinned by tests/test_oracle_clouds.py against golden vectors captured from the imported reference.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi
from scipy import signal

_S8 = ndi.generate_binary_structure(2, 2)


def _dil(x, n, structure=None):
    return ndi.binary_dilation(x, iterations=n, structure=structure)


def _ero_then_dil(x, n_ero, n_dil):
    """1 - dilate(x == 0, n_ero), then dilate n_dil: the reference's opening idiom (4-connected)."""
    return _dil(1 - _dil(x == 0, n_ero), n_dil)


def _ndwi(a):
    with np.errstate(all='ignore'):
        return (a[..., 1] - a[..., 3]) / (a[..., 1] + a[..., 3])


def _ndvi(a):
    with np.errstate(all='ignore'):
        return (a[..., 3] - a[..., 2]) / (a[..., 3] + a[..., 2])


def _ndbi(a):
    with np.errstate(all='ignore'):
        return (a[..., 8] - a[..., 3]) / (a[..., 8] + a[..., 3])


def _winsum3(x):
    """3x3 moving sum with reflect padding (CR.py:1247-1252)."""
    p = np.pad(x, 1, mode='reflect')
    p[3:] -= p[:-3]
    p[:, 3:] -= p[:, :-3]
    return p.cumsum(0)[2:].cumsum(1)[:, 2:]


# ----------------------------------------------------------------------------------------------- windows over dates
def shadow_window(T, t):
    """dates used as the shadow reference of date t (CR.py:1268-1275)"""
    lo, hi = max(0, t - 4), min(T, t + 3)
    if hi - lo == 3:
        if hi == T:
            lo = max(lo - 1, 0)
        if lo == 0:
            hi = min(hi + 1, T)
    return np.arange(lo, hi)


def cloud_windows(T, t):
    """(others, close) of date t for the cloud candidates (CR.py:1341-1363)"""
    lo, hi = max(0, t - 2), min(T, t + 3)
    if hi - lo == 3:
        if hi == T:
            lo = max(lo - 2, 0)
        if lo == 0:
            hi = min(hi + 2, T)
    close = [max(0, t - 1), min(T - 1, t + 1)]
    if close[1] - close[0] < 2:
        close = [close[0] + 1, close[1] + 1] if close[0] == 0 else [close[0] - 1, close[1] - 1]
    if close[-1] >= T - 2 and T > 3:
        close = [close[0] - 1] + close
    return np.arange(lo, hi), np.array(close)


# ----------------------------------------------------------------------------------------------- false-positive helper
def detect_pfcp(img, dem, urban=None):
    """CR.py:1109-1212 -> (fcps [T,H,W] float32, pfps [T,H,W] float)"""
    T, H, W = img.shape[:3]
    ndwi_med = np.median(_ndwi(img), axis=0)
    with np.errstate(all='ignore'):
        built = np.logical_and(_ndbi(img) > 0, _ndbi(img) > _ndvi(img))
    pf = np.median(built, axis=0) * (ndwi_med < 0)
    if urban is None:
        pf = np.zeros_like(dem)                              # the reference's except branch (no raster available)
    else:
        core, near = urban
        pf[core == 1] = 1.
        pf[near == 0] = 0.
    pf[(dem / 90) > 0.10] = 0.
    pfps = np.tile(pf[np.newaxis], (T, 1, 1))

    def half(b):
        return np.mean(np.reshape(b, (b.shape[0] // 2, 2, b.shape[1] // 2, 2)), axis=(1, 3))

    box = np.ones((7, 7)) / 49

    def local_var(r):
        return (signal.convolve2d(r ** 2, box, mode='same', boundary='symm') -
                signal.convolve2d(r, box, mode='same', boundary='symm') ** 2)

    cdis = np.zeros((T, H, W), dtype=np.float32)
    for t in range(T):
        b8 = half(ndi.gaussian_filter(np.copy(img[t, ..., 3]), sigma=0.5, truncate=3))
        b8a, b7 = half(np.copy(img[t, ..., 7])), half(np.copy(img[t, ..., 6]))
        with np.errstate(all='ignore'):
            va, vb = local_var(b8 / b8a), local_var(b7 / b8a)
            cdi = (vb - va) / (vb + va)
        hit = (cdi >= -0.4).repeat(2, axis=0).repeat(2, axis=1)
        cdis[t] = hit * (_ndvi(img[t]) < 0.4)
    for t in range(T):
        cdis[t] = _dil(cdis[t], 6, _S8)
        pfps[t] = _dil(pfps[t], 6, _S8)
    return pfps * cdis, pfps


# ----------------------------------------------------------------------------------------------- main
def identify_clouds_shadows(img, dem, forest=None, urban=None, trace=None):
    """CR.py:1215-1677 -> (clouds [T,H,W] float32 in {0,1}, fcps [T,H,W] bool).
    trace: optional dict receiving the state after each stage (numbered like the stages of the HIP driver)."""
    def keep(name, x):
        if trace is not None:
            trace[name] = np.array(x, dtype=np.float32, copy=True)
    T = img.shape[0]
    vis = img[..., :3]
    water = np.nanmedian(_ndwi(img), axis=0)
    forest = np.zeros_like(dem) if forest is None else forest

    # coarse single-date cloud mask (Hollstein et al. 2016, Fig. 6), opened
    with np.errstate(all='ignore'):
        clm = (img[..., 7] > 0.166) * (img[..., 1] > 0.28) * (img[..., 5] / img[..., 8] < 4.292)
    for t in range(T):
        clm[t] = _ero_then_dil(clm[t], 2, 10)
    keep("1_clm", clm)

    # ---- shadows: darker than the cloud-free temporal reference in B8A / B11 / blue ---------------------------------
    ref4 = img[..., [0, 1, 7, 8]]
    all_ref = np.copy(ref4)
    all_ref[clm > 0] = np.nan
    with np.errstate(all='ignore'):
        all_med = np.nanmedian(all_ref, axis=0)
    all_med[np.isnan(all_med)] = np.median(ref4, axis=0)[np.isnan(all_med)]
    shadows = np.zeros(img.shape[:3], dtype=np.float32)
    for t in range(T):
        w = shadow_window(T, t)
        loc = np.copy(ref4)[w]
        loc[clm[w] > 0] = np.nan
        with np.errstate(all='ignore'):
            loc_max, loc_med = np.nanmax(loc, axis=0), np.nanmedian(loc, axis=0)
        loc_med[np.isnan(loc_med)] = np.min(ref4, axis=0)[np.isnan(loc_med)]
        b, g, a8, s11 = img[t, ..., 0], img[t, ..., 1], img[t, ..., 7], img[t, ..., 8]
        with np.errstate(all='ignore'):
            d8a_max, d11_max = (a8 - loc_max[..., 2]) < -0.04, (s11 - loc_max[..., 3]) < -0.04
            s = ((s11 - loc_med[..., 3]) < -0.04) * ((a8 - loc_med[..., 2]) < -0.04) * (b < 0.09) * \
                ((b - loc_med[..., 0]) < -0.02) * (a8 < 0.17)
            dark = d11_max * d8a_max * (b < 0.03) * (a8 < 0.18)
        dark[water > 0] = 0.
        s = np.maximum(s, dark)
        s[water > 0] = 0.
        with np.errstate(all='ignore'):
            slope = d8a_max * d11_max * (b < 0.07) * (a8 < 0.18)
            slope = slope * (np.sum(img[t, ..., :3], axis=-1) < 0.28)
        slope[water > 0] = 0.
        slope = slope * (dem >= 25)
        s = np.maximum(s, slope)
        with np.errstate(all='ignore'):
            wet = ((b - all_med[..., 0]) < -0.05) * ((g - all_med[..., 1]) < -0.05) * (a8 < 0.03) * \
                  ((all_med[..., 1] - g) > 0.02) * (water > 0)
        shadows[t] = s + wet
    keep("2_shadow_candidates", shadows)
    for t in range(T):
        s = _ero_then_dil(shadows[t], 2, 3)
        d = ndi.distance_transform_edt(1 - s)
        shadows[t] = 1 - (d > 5)
    keep("3_shadows", shadows)

    # ---- clouds: brighter than the darkest shadow-free neighbours ---------------------------------------------------
    clouds = np.zeros_like(shadows)
    dark_ref = np.copy(vis)
    if T > 2:
        dark_ref[shadows > 0] = np.nan
    for t in range(T):
        others, close = cloud_windows(T, t)
        if T > 2:
            with np.errstate(all='ignore'):
                upper = np.nanmin(dark_ref[others], axis=0)
                near = np.nanmin(dark_ref[close], axis=0).astype(np.float32)
            gap = np.isnan(upper[..., 0])
            for c in range(3):
                upper[..., c][gap] = np.percentile(img[..., c], 25, axis=0)[gap]
            lo_i, hi_i = close[0], close[-1]
            for _ in range(10):
                if np.sum(np.isnan(near) > 0):
                    lo_i, hi_i = max(lo_i - 1, 0), min(hi_i + 1, T)
                    wider = np.array([x for x in np.arange(lo_i, hi_i) if x != t])
                    with np.errstate(all='ignore'):
                        fill = np.nanmin(dark_ref[wider], axis=0).astype(np.float32)
                    near[np.isnan(near)] = fill[np.isnan(near)]
            if np.sum(np.isnan(near) > 0):
                near[np.isnan(near)] = np.min(vis, axis=0)[np.isnan(near)]
        else:
            near = np.min(dark_ref, axis=0).astype(np.float32)
            upper = near
        thr = np.maximum(np.minimum((near[..., 0] / 0.02 / 100) + 0.005, 0.10), 0.05)
        thr[forest == 1] -= 0.02
        thr = np.maximum(thr, 0.04)
        frac_i, frac_c, extra = 0., 1., 0.
        while (frac_c - frac_i) > 0.075:
            with np.errstate(all='ignore'):
                far = ((img[t, ..., 0] - upper[..., 0]) > 0.08) * ((img[t, ..., 1] - upper[..., 1]) > 0.08) * \
                      ((img[t, ..., 2] - upper[..., 2]) > 0.07)
                nearc = ((img[t, ..., 0] - near[..., 0]) > (thr + extra + 0.01)) * \
                        ((img[t, ..., 1] - near[..., 1]) > (thr + extra + 0.01)) * \
                        ((img[t, ..., 2] - near[..., 2]) > (thr + extra))
            frac_i, frac_c = np.mean(far > 0), np.mean(nearc > 0)
            extra += 0.0025
        nearc = nearc * (np.sum(img[t, ..., :3], axis=-1) < 0.75)
        eroded = 1 - _dil(nearc == 0, 2)
        nearc[forest == 0] = eroded[forest == 0]
        clouds[t] = np.maximum(far, nearc)
    keep("4_cloud_candidates", clouds)

    # ---- per-image brightness outliers (z-score of brightness / median brightness) ----------------------------------
    bsum = np.sum(vis, axis=-1)
    masked = np.copy(bsum)
    masked[np.logical_or(clouds > 0, shadows > 0)] = np.nan
    with np.errstate(all='ignore'):
        med_b = np.nanmedian(masked, axis=(1, 2))
    bright = np.zeros_like(clouds, dtype=np.float32)
    for t in range(T):
        with np.errstate(all='ignore'):
            ratio = np.sum(img[t, ..., :3], axis=-1) / med_b[t]
        ratio[water > 0] = 1.
        with np.errstate(all='ignore'):
            if np.sum(clouds[t] < 0.90):
                sel = ratio[clouds[t] == 0]
                z = (ratio - np.nanmean(sel)) / np.nanstd(sel)
            else:
                z = (ratio - np.nanmean(ratio)) / np.nanstd(ratio)
        bright[t][z > 3.5] = 1.
        bright[t] *= (water < 0)
    repeats = np.sum((bright - clouds) > 0, axis=0)
    for t in range(T):
        bright[t][repeats > 1] = 0.
    clouds = np.maximum(clouds, bright)
    keep("5_brightness", clouds)

    # clouds are white: drop coloured bright surfaces
    for t in range(T):
        mean_b = np.mean(img[t, ..., :3], axis=-1)
        rng = np.max(img[t, ..., :3], axis=-1) - np.min(img[t, ..., :3], axis=-1)
        with np.errstate(all='ignore'):
            coloured = (mean_b < 0.4) * ((rng / mean_b) > 0.5)
        clouds[t] = clouds[t] * (1 - coloured)
    keep("6_white", clouds)

    # urban false positives (Fmask 4.0 parallax + built-up index)
    fcps, pfcps = detect_pfcp(img, dem, urban)
    keep("7_fcps", fcps); keep("7_pfps", pfcps)

    def not_much_brighter(t):
        lo, hi = max(t - 1, 0), min(t + 2, T)
        floor = np.min(img[lo:hi, ..., :3], axis=(0, 3))
        return (np.mean(img[t, ..., :3], axis=-1) - floor) < 0.4

    for t in range(T):
        drop = np.logical_and(fcps[t] > 0, not_much_brighter(t))
        clouds[t][drop] = 0.
        shadows[t][drop] = 0.
    # bright bare surfaces: NIR / SWIR1 < 0.75
    with np.errstate(all='ignore'):
        nsr = (img[..., 3] / (img[..., 8] + 0.01)) < 0.75
    nsr = ndi.binary_dilation(nsr, iterations=3)               # 3-D cross: also spreads to the neighbouring dates
    for t in range(T):
        nsr[t][water < 0] = 0.
        clouds[t][np.logical_and(nsr[t] > 0, not_much_brighter(t))] = 0.
    # water false positives, lone pixels, dark pixels
    for t in range(T):
        clouds[t][_dil((water > 0) * (img[t, ..., 8] < 0.11), 10)] = 0.
    for t in range(T):
        clouds[t][_winsum3(clouds[t]) < 5] = 0.
    for t in range(T):
        dark = _dil(np.sum(img[t, ..., :3], axis=-1) < 0.21, 3) * (1 - forest)
        clouds[t][dark.astype(np.uint8)] = 0.
    keep("8_false_positives", clouds); keep("8_shadows", shadows); keep("8_nsr", nsr)

    # ---- shape clean-up: erode / dilate urban and non-urban clouds differently --------------------------------------
    for t in range(T):
        clouds[t] = 1 - _dil(clouds[t] == 0, 1)
        pfcps[t] = _dil(pfcps[t], 5)
        urban_c = 1 - _dil((clouds[t] * pfcps[t]) == 0, 3)
        rest = clouds[t] * (1 - pfcps[t])
        n9 = _winsum3(rest)
        big, small = np.copy(rest), np.copy(rest)
        big[n9 < 6] = 0.
        small[n9 >= 6] = 0.
        rest = np.maximum(_dil(big, 5), _dil(small, 1))
        rest = 1 - (ndi.distance_transform_edt(1 - rest) > 3)
        clouds[t] = rest + urban_c
    keep("9_shape", clouds)
    # implausible shadow amounts: keep only shadows near clouds (or on high ground)
    for t in range(T):
        with np.errstate(all='ignore'):
            ms, mc = np.mean(shadows[t]), np.mean(clouds[t])
            if ms > mc + 0.3 and mc < 0.3:
                shadows[t] = shadows[t] * np.logical_or(_dil(np.copy(clouds[t]), 50), dem >= 30)
            if np.mean(clouds[t]) < 0.05 and (np.mean(shadows[t]) / np.mean(clouds[t])) > 3:
                shadows[t] = shadows[t] * np.logical_or(_dil(np.copy(clouds[t]), 50), dem >= 30)
    keep("10_shadows", shadows)
    clouds = np.maximum(clouds, shadows)
    fcps = ndi.binary_dilation(np.maximum(fcps, nsr), iterations=2)

    # false-negative shadows from the per-image blue statistics
    for t in range(T):
        if np.mean(clouds[t]) < 0.9:
            with np.errstate(all='ignore'):
                inv = 1 / np.copy(img[t, ..., 0])[clouds[t] == 0]
                level = np.mean(inv) + 2 * np.std(inv)
                extra_s = (1 / img[t, ..., 0] > level) * (img[t, ..., 7] < 0.17)
            extra_s = _ero_then_dil(extra_s, 2, 2)
            extra_s[water > 0] = 0.
            clouds[t] = np.maximum(clouds[t], extra_s)
    clouds[clouds > 1] = 1.
    keep("11_extra_shadows", clouds)

    # haze: whole images that are bright, flat and white
    mean_b = np.mean(vis, axis=-1)
    mb, sb, sw = [], [], []
    for t in range(T):
        if np.mean(clouds[t]) < 1:
            clear = clouds[t] == 0
            mb.append(np.mean(mean_b[t][clear]))
            sb.append(np.std(mean_b[t][clear]))
            sw.append(np.std(np.ptp(img[t, ..., :3][clear], axis=1)))
    hb, hs, hw = mb / np.median(mb), sb / np.median(sb), sw / np.median(sw)
    haze = np.logical_or((hb >= 1.5) * (hs <= 0.67) * (hw < 1), (hb >= 1.3) * (hs <= 0.5))
    for t in range(len(haze)):
        if haze[t]:
            clouds[t] = 1.
    return clouds, fcps
