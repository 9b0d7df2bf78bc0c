"""CPU oracle for process_tile (src/download_and_predict_job.py:641-995) from the arrays it loads (file IO excluded):
Sen2Cor mask clean-up, Sentinel-1 scaling, DEM median filter, 20 m -> 10 m, missing-data screening, snow map,
cloud / shadow detection with the re-detection rounds after heavily clouded dates are dropped, gap-fill, final clip.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/test_oracle_tile.py against golden vectors captured by
running the reference's process_tile with its file loader replaced (tools/gen_golden.py)."""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

from oracle import restate_clouds as C
from oracle import restate_gapfill as G
from oracle import restate_numpy as R


def clean_sen2cor_mask(clm20):
    """job.py:688-697: 20 m mask -> 10 m; two consecutive dates flagged at a pixel are both dropped (sequentially)."""
    clm = clm20.repeat(2, axis=1).repeat(2, axis=2)
    for i in range(clm.shape[0]):
        lo, hi = max(i - 1, 0), min(i + 1, clm.shape[0])
        both = np.sum(clm[lo:hi], axis=0) == 2
        clm[lo:hi, both] = 0.
    return clm


def snow_flags(s2):
    """job.py:799-817 (boolean variant of the snow probability)."""
    with np.errstate(all='ignore'):
        ndsi = (s2[..., 1] - s2[..., 8]) / (s2[..., 1] + s2[..., 8])
        ndsi[ndsi < 0.10] = 0.
        ndsi[ndsi > 0.42] = 0.42
        p = (ndsi - 0.1) / 0.32
        p[s2[..., 3] < 0.10] = 0.
        p[np.logical_and(s2[..., 3] > 0.35, p > 0)] = 1.
        p[s2[..., 0] < 0.10] = 0.
        p[np.logical_and(s2[..., 0] > 0.22, p > 0)] = 1.
        p[(s2[..., 0] / s2[..., 2]) < 0.75] = 0.
    return p > 0


def process_tile_arrays(raw, forest=None, urban=None, sampler=G.reference_sampler, cloudshad=None):
    """raw: dict with s2_10 / s2_20 / s1 (uint16), dem (metres), dates, clouds, clm (20 m mask or None).
    -> (sentinel2, dates, interp, s1, dem / 90, cloudshad, snow) like process_tile (make_shadow=True).
    cloudshad [T, X, Y] (optional): a GIVEN cloud + shadow mask stands in for identify_clouds_shadows (job.py:839 and the
    re-detections after dates are dropped return its surviving dates; no false-positive mask) -- the flow of the single-call
    tile entry's staged fall-back."""
    given = None if cloudshad is None else np.array(cloudshad, dtype=np.float32, copy=True)
    clm = clean_sen2cor_mask(np.array(raw["clm"], copy=True)) if raw.get("clm") is not None else None
    s1 = R.s1_to_db(raw["s1"])
    s2_10, s2_20 = R.to_float32(raw["s2_10"]), R.to_float32(raw["s2_20"])
    dem = ndi.median_filter(np.array(raw["dem"], copy=True), size=5)
    dates = np.array(raw["dates"], copy=True)
    # job.py:716-721: the 20 m stack decides the tile's grid; S1 (already dB-scaled), the 10 m bands and the filtered DEM are brought onto it
    width, height = s2_20.shape[1] * 2, s2_20.shape[2] * 2
    s1, s2_10, dem = (R.adjust_shape(a, width, height) for a in (s1, s2_10, dem))
    if s2_10.ndim == 3:                                                # :724-727, a single image
        s2_10, s2_20 = s2_10[np.newaxis], (s2_20[np.newaxis] if s2_20.ndim == 3 else s2_20)
    clouds = np.array(raw["clouds"], copy=True) if raw.get("clouds") is not None else np.zeros((len(dates), 1, 1), np.float32)
    s2 = R.upsample_20m(s2_10, s2_20)

    def drop(idx):
        nonlocal clouds, dates, s2, clm, given
        if given is not None:
            given = np.delete(given, idx, axis=0)
        if clouds.shape[0] == len(dates):
            clouds = np.delete(clouds, idx, axis=0)
        dates = np.delete(dates, idx)
        s2 = np.delete(s2, idx, axis=0)
        if clm is not None:
            clm = np.delete(clm, idx, axis=0)

    missing = R.id_missing_px(s2, 2)
    if len(missing) > 0:
        drop(missing)
    flags = snow_flags(s2)
    per_img = np.mean(flags, axis=(1, 2))
    snow = 1 - ndi.binary_dilation(np.mean(flags, axis=0) < 0.7, iterations=2)
    snowy = np.argwhere(per_img > 0.25).flatten()
    if len(snowy) > 10:
        drop(snowy)
    s2 = R.interpolate_missing_vals(s2)

    def detect(first):
        if given is not None:
            cs, fc = given.copy(), None
        else:
            cs, fc = C.identify_clouds_shadows(s2, dem, forest, urban)
        if clm is not None:
            if first and fc is not None:
                clm[fc] = 0.
            cs = np.maximum(cs, clm)
        return cs, fc

    cloudshad, fcps = detect(True)
    interp = G.id_areas_to_interp(cloudshad)
    for rnd in range(3):
        heavy = np.argwhere(np.mean(interp > 0, axis=(1, 2)) > 0.9).flatten()
        if len(heavy) > 0:
            drop(heavy)
            interp = np.delete(interp, heavy, axis=0)
            cloudshad, fcps = detect(False)
            if rnd < 2:
                interp = G.id_areas_to_interp(cloudshad)
    interp = G.id_areas_to_interp(cloudshad)
    s2, interp, to_remove = G.remove_cloud_and_shadows(s2, cloudshad, fcps if fcps is not None else np.zeros(s2.shape[1:3], bool), sampler)
    if len(to_remove) > 0:
        drop(to_remove)
        interp = np.delete(interp, to_remove, axis=0)
        cloudshad, fcps = detect(False)
        interp = G.id_areas_to_interp(cloudshad)
    return np.clip(s2, 0, 1), dates, interp, s1, dem / 90, cloudshad, snow
