"""Seeded synthetic inputs shaped like the reference's per-tile arrays.

Used by bench.py, the tests and tools/gen_golden.py (so that fixtures only need
to store seeds + expected outputs).  numpy only; no device work here.

Raw-tile layout (BASELINE.json `[12, 15, 618, 618]`), planar
float32 [T, 15, H, W]:
  0-3   S2 10 m bands (B2,B3,B4,B8)          4-9  S2 20/40 m bands, nearest-replicated
  10-11 S1 VV/VH already in [0,1] dB scale   12   DEM / 90 (constant over T)
  13    cloud probability                    14   binary cloud+shadow mask
"""
from __future__ import annotations

import numpy as np


def _smooth_field(rng, h, w, scale=32):
    """Cheap smooth random field in [0,1]: bilinear-upsampled coarse noise."""
    ch, cw = h // scale + 2, w // scale + 2
    c = rng.random((ch, cw))
    y = np.linspace(0, ch - 1.001, h)
    x = np.linspace(0, cw - 1.001, w)
    y0, x0 = y.astype(int), x.astype(int)
    fy, fx = (y - y0)[:, None], (x - x0)[None, :]
    a = c[y0][:, x0] * (1 - fy) * (1 - fx) + c[y0 + 1][:, x0] * fy * (1 - fx)
    b = c[y0][:, x0 + 1] * (1 - fy) * fx + c[y0 + 1][:, x0 + 1] * fy * fx
    return a + b


def synth_dates(rng, T):
    return np.sort(rng.choice(np.arange(0, 365), size=T, replace=False)).astype(np.int64)


def synth_tile(seed=1234, T=12, H=618, W=618, cloud_frac=0.0):
    """Reference-layout per-tile arrays, i.e. what process_tile (job.py:641-995) returns:
    s2 [T,H,W,10] f32 in (0,1), dates [T], interp [T,H,W] f32, s1 [12,H,W,2] f32, dem [H,W] f32."""
    rng = np.random.default_rng(seed)
    dates = synth_dates(rng, T)
    veg = _smooth_field(rng, H, W, 48).astype(np.float32)            # "tree-ness"
    tex = rng.random((H, W), dtype=np.float32) * 0.04
    season = (0.5 + 0.5 * np.sin(2 * np.pi * (dates[:, None, None] - 100) / 365.0)).astype(np.float32)
    base = np.array([0.05, 0.08, 0.07, 0.30, 0.12, 0.22, 0.27, 0.30, 0.20, 0.12], dtype=np.float32)
    vegd = np.array([-0.02, -0.02, -0.04, 0.15, 0.0, 0.08, 0.10, 0.12, -0.08, -0.06], dtype=np.float32)
    s2 = (base[None, None, None, :]
          + vegd[None, None, None, :] * (veg[None, :, :, None] * (0.6 + 0.4 * season[..., None]))
          + tex[None, :, :, None]
          + rng.random((T, H, W, 10), dtype=np.float32) * 0.02)
    s2 = np.clip(s2, 0.002, 0.95).astype(np.float32)
    s1 = np.clip(0.55 + 0.2 * veg[None, :, :, None] + rng.random((12, H, W, 2), dtype=np.float32) * 0.1
                 - np.array([0.0, 0.2], dtype=np.float32), 0, 1).astype(np.float32)
    dem = (_smooth_field(rng, H, W, 96) * 3.0).astype(np.float32)     # already / 90
    interp = np.zeros((T, H, W), dtype=np.float32)
    if cloud_frac > 0:
        yy, xx = np.mgrid[0:H, 0:W]
        for t in range(T):
            for _ in range(rng.integers(0, 3)):
                cy, cx = rng.integers(0, H), rng.integers(0, W)
                ry, rx = rng.integers(20, int(60 + 300 * cloud_frac)), rng.integers(20, int(60 + 300 * cloud_frac))
                d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
                interp[t] = np.maximum(interp[t], np.clip(1.5 - d, 0, 1).astype(np.float32))
    return s2, dates, interp, s1, dem


def synth_raw_tile(seed=1234, T=12, H=618, W=618):
    """BASELINE raw stack [T, 15, H, W] planar f32 + dates (see module docstring)."""
    s2, dates, interp, s1, dem = synth_tile(seed, T, H, W, cloud_frac=0.15)
    rng = np.random.default_rng(seed + 7)
    raw = np.empty((T, 15, H, W), dtype=np.float32)
    raw[:, 0:4] = np.moveaxis(s2[..., 0:4], -1, 1)
    lo = s2[:, ::2, ::2, 4:10]                                        # 20 m grid
    lo = np.repeat(np.repeat(lo, 2, axis=1), 2, axis=2)[:, :H, :W]
    raw[:, 4:10] = np.moveaxis(lo, -1, 1)
    s1t = s1[np.minimum(np.arange(T), 11)]
    raw[:, 10:12] = np.moveaxis(s1t, -1, 1)
    raw[:, 12] = dem[None]
    mask = (interp > 0.5).astype(np.float32)
    raw[:, 13] = np.clip(interp + rng.random((T, H, W), dtype=np.float32) * 0.1, 0, 1)
    raw[:, 14] = mask
    bright = 0.35 * mask[..., None]                                    # clouds are bright
    raw[:, 0:10] += np.moveaxis(bright * np.ones(10, dtype=np.float32), -1, 1)
    return np.clip(raw, 0, 1), dates


def synth_windows(seed=0, N=1, L=4, W=172, C=17):
    """U(-1,1) model inputs [N, L+1, W, W, C] (NHWC per frame, as fed to predict/Placeholder:0)."""
    rng = np.random.default_rng(seed)
    return (rng.random((N, L + 1, W, W, C), dtype=np.float32) * 2 - 1).astype(np.float32)


def synth_bright_window(seed=21, L=4, W=172):
    """Un-normalised [L+1, W, W, 17] window with two bright-bare patches (job.py:1099-1122 test input)."""
    rng = np.random.default_rng(seed)
    img = (rng.random((L + 1, W, W, 17)) * 0.5).astype(np.float32)
    for (a, b, c, d) in [(40, 70, 50, 90), (120, 124, 10, 13), (0, 12, 160, 172)]:
        img[:, a:b, c:d, 8] = 0.6       # high SWIR
        img[:, a:b, c:d, :3] = 0.3      # bright visible
        img[:, a:b, c:d, 3] = 0.31      # NIR/SWIR < 0.9
    return img


def synth_gapfill_scene(seed=31, T=6, H=224, W=224):
    """Cloudy Sentinel-2 stack for the gap-fill stage (cloud_removal.py:888-973 input):
    tiles [T,H,W,10] f32 with per-date gain/offset drift, a lake (NDWI > 0), bright cloud blobs and
    darker shadow blobs; probs [T,H,W] f32 binary cloud+shadow mask; pfcps [H,W] bool (none)."""
    s2, dates, _, _, _ = synth_tile(seed=seed, T=T, H=H, W=W)
    rng = np.random.default_rng(seed + 1)
    gain = (1.0 + 0.15 * (rng.random((T, 1, 1, 10), dtype=np.float32) - 0.5)).astype(np.float32)
    off = (0.02 * (rng.random((T, 1, 1, 10), dtype=np.float32) - 0.5)).astype(np.float32)
    tiles = (s2 * gain + off).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    lake = ((yy - 0.7 * H) / (0.12 * H)) ** 2 + ((xx - 0.25 * W) / (0.18 * W)) ** 2 < 1.0
    tiles[:, lake, 3] = 0.03                     # NIR dark over water -> NDWI > 0
    tiles[:, lake, 1] = 0.08
    probs = np.zeros((T, H, W), dtype=np.float32)
    for t in range(T):
        nblob = [2, 0, 1, 3, 1, 2][t % 6]
        for b in range(nblob):
            cy, cx = rng.integers(0, H), rng.integers(0, W)
            ry, rx = rng.integers(H // 12, H // 4), rng.integers(W // 12, W // 4)
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
            probs[t][m] = 1.0
            tiles[t][m] += (0.35 if b % 2 == 0 else -0.04)          # cloud / shadow
    big = min(3, T - 1)
    probs[big, : (6 * H) // 10, :] = 1.0          # one date mostly covered (< 40 000 clear px -> multi-date fit)
    tiles[big, : (6 * H) // 10, :, :] += 0.3
    tiles = np.clip(tiles, 0.001, 0.98).astype(np.float32)
    return tiles, dates, probs, np.zeros((H, W), dtype=bool)


def synth_detection_scene(seed=77, T=7, H=120, W=112):
    """Raw (not yet gap-filled) Sentinel-2 stack for the multi-temporal cloud / shadow detector
    (cloud_removal.py:1215-1677): bright white cloud blobs with displaced dark shadows, a lake, a built-up patch
    (high SWIR, low NIR), a sand patch (NIR/SWIR < 0.75), a hazy date when T > 5.
    -> (img [T,H,W,10] f32, dem [H,W] f32 metres, forest [H,W] f32 in {0,1}, urban_core [H,W] u8, urban_near [H,W] u8)"""
    s2, dates, _, _, dem = synth_tile(seed=seed, T=T, H=H, W=W)
    rng = np.random.default_rng(seed + 7)
    img = s2.copy()
    yy, xx = np.mgrid[0:H, 0:W]
    lake = ((yy - 0.75 * H) / (0.13 * H)) ** 2 + ((xx - 0.2 * W) / (0.15 * W)) ** 2 < 1.0
    img[:, lake, 3] = 0.03; img[:, lake, 1] = 0.08; img[:, lake, 8] = 0.02; img[:, lake, 7] = 0.025
    town = (np.abs(yy - 0.3 * H) < 0.12 * H) & (np.abs(xx - 0.7 * W) < 0.14 * W)
    img[:, town, 8] = 0.32; img[:, town, 3] = 0.22; img[:, town, 2] = 0.2
    sand = ((yy - 0.15 * H) / (0.08 * H)) ** 2 + ((xx - 0.2 * W) / (0.1 * W)) ** 2 < 1.0
    img[:, sand, 3] = 0.25; img[:, sand, 8] = 0.42; img[:, sand, 0] = 0.2; img[:, sand, 1] = 0.24; img[:, sand, 2] = 0.3
    for t in range(T):
        for b in range([2, 0, 1, 3, 1, 2, 0][t % 7]):
            cy, cx = rng.integers(0, H), rng.integers(0, W)
            ry, rx = rng.integers(H // 14, H // 5), rng.integers(W // 14, W // 5)
            cloud = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
            shadow = ((yy - cy - ry // 2 - 3) / ry) ** 2 + ((xx - cx - rx // 2 - 3) / rx) ** 2 < 1.0
            shadow &= ~cloud
            img[t][shadow] *= 0.35
            img[t][cloud, :3] = 0.55 + 0.1 * rng.random(int(cloud.sum()))[:, None]
            img[t][cloud, 3:] = 0.5 + 0.05 * rng.random((int(cloud.sum()), 7))
    if T > 5:
        img[5, ..., :3] = img[5, ..., :3] * 0.3 + 0.33          # haze: bright, flat, white
    img = np.clip(img + rng.normal(0, 0.003, img.shape), 0.001, 0.98).astype(np.float32)
    demm = (dem * 12.0).astype(np.float32)                        # 0 .. 36 m: both sides of the 9 / 25 / 30 m rules
    forest = (_smooth_field(np.random.default_rng(seed + 9), H, W, 25) > 0.55).astype(np.float32)
    core = town.astype(np.uint8)
    near = (np.abs(yy - 0.3 * H) < 0.25 * H) & (np.abs(xx - 0.7 * W) < 0.3 * W)
    return img, demm, forest, core, near.astype(np.uint8)


def synth_raw_files(seed=91, T=6, w20=40, h20=44, with_clm=True):
    """The arrays process_tile (job.py:641-995) loads from temp/raw/*: uint16 Sentinel-2 at 10 m [T, 2*w20, 2*h20, 4] and
    20 m [T, w20, h20, 6], uint16 Sentinel-1 [12, X, Y, 2], DEM in metres [X, Y], day-of-year dates, the s2cloudless
    probabilities [T, X, Y] (only sliced / deleted there) and, optionally, the 20 m Sen2Cor cloud mask [T, w20, h20]."""
    X, Y = 2 * w20, 2 * h20
    img, dem, _, _, _ = synth_detection_scene(seed, T, X, Y)
    rng = np.random.default_rng(seed + 3)
    if T > 3:
        img[2, : (19 * X) // 20] = np.clip(img[2, : (19 * X) // 20] * 0.2 + 0.55, 0, 0.98)      # one date almost fully clouded
    q = lambda a: np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)
    s2_10 = q(img[..., :4])
    lo = img[..., 4:].reshape(T, w20, 2, h20, 2, 6).mean(axis=(2, 4))
    s2_20 = q(lo)
    s2_10[0, 3:5, 4:9, :] = 0                                            # a few missing samples
    _, dates, _, s1, _ = synth_tile(seed=seed, T=T, H=X, W=Y)
    s1u = q(s1)
    s1u[1, :4, :5] = 65535
    clouds = rng.random((T, X, Y)).astype(np.float32)
    clm = None
    if with_clm:
        clm = np.zeros((T, w20, h20), dtype=np.float32)
        clm[1, 5:12, 6:15] = 1.0; clm[2, 5:12, 6:15] = 1.0                # "two in a row" -> dropped (job.py:691-697)
        clm[min(4, T - 1), 20:30, 10:22] = 1.0
    return {"s2_10": s2_10, "s2_20": s2_20, "s1": s1u, "dem": dem.astype(np.float32), "dates": np.asarray(dates), "clouds": clouds, "clm": clm}


def misshape_raw(raw, d10=(1, -1), ds1=(2, 1), ddem=(-1, 2)):
    """`raw` (synth_raw_files) with the 10 m bands, Sentinel-1 and the DEM a pixel or two OFF the grid of the 20 m stack -- the case
    adjust_shape (job.py:260-310) exists for: d* = (rows, cols) to add (reflect-padded at both ends) or to take away (cropped at both
    ends).  Saturated Sentinel-1 samples are planted in rows that a crop removes, so that the order "per-image median over the image as
    stored, THEN adjust_shape" (job.py:699-718) shows in the result."""
    def grow(a, ax, d):
        if d == 0:
            return a
        if d > 0:
            pad = [(0, 0)] * a.ndim
            pad[ax] = (d - d // 2, d // 2)
            return np.pad(a, pad, mode="reflect")
        d = -d
        sl = [slice(None)] * a.ndim
        sl[ax] = slice(d // 2, a.shape[ax] - (d - d // 2))
        return a[tuple(sl)]
    out = dict(raw)
    out["s2_10"] = np.ascontiguousarray(grow(grow(raw["s2_10"], 1, d10[0]), 2, d10[1]))
    s1 = np.ascontiguousarray(grow(grow(raw["s1"], 1, ds1[0]), 2, ds1[1]))
    s1[3, 0, :, :] = 65535                       # first row: removed by a 1-pixel or an even crop
    s1[5, :, 0, 0] = 65535
    out["s1"] = s1
    out["dem"] = np.ascontiguousarray(grow(grow(raw["dem"], 0, ddem[0]), 1, ddem[1]))
    return out
