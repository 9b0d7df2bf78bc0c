"""Host-side mirror of the reference's per-tile functions (src/download_and_predict_job.py).

Same names, argument meaning and numpy-in / numpy-out contracts as the reference, so the
download / mosaic notebooks and the job script can swap them in; the
arithmetic runs in libttc_hip.so on an MI355X.  There is no CPU fallback.

  TTCSession.run(fetch, feed_dict)          <->  tf.Session.run on the frozen graphs (job.py:1788-1824)
  normalize_subtile(subtile)                <->  job.py:316-325
  predict_subtile(subtile, sess, op, size)  <->  job.py:328-369
  process_subtiles(...)                     <->  job.py:1125-1483 (returns the windows instead of np.save)
  load_mosaic_predictions(windows, depth=1) <->  job.py:1515-1641 (takes the windows instead of a folder)
  predict_tile(...)                         ==   process_subtiles + load_mosaic_predictions in one call
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib, temporal, weights as _weights

SIZE = 172 - 14          # job.py:60
LEN = 4                  # job.py:61 / args.length default (job.py:1715)

# job.py:1829-1842
min_all = [0.006576638437476157, 0.0162050812542916, 0.010040436408026246, 0.013351644159609368,
           0.01965362020294499, 0.014229037918669413, 0.015289539940489814, 0.011993591210803388,
           0.008239871824216068, 0.006546120393682765, 0.0, 0.0, 0.0, -0.1409399364817101,
           -0.4973397113668104, -0.09731556326714398, -0.7193834232943873]
max_all = [0.2691233691920348, 0.3740291447318227, 0.5171435111009385, 0.6027466239414053,
           0.5650263218127718, 0.5747005416952773, 0.5933928435187305, 0.6034943160143434,
           0.7472037842374304, 0.7000076295109483, 0.4, 0.948334642387533, 0.6729257769285485,
           0.8177635298774327, 0.35768999002433816, 0.7545951919107605, 0.7602693339366691]

# tensor names of the reference graphs (job.py:1794-1796, :1807-1824)
PREDICT_INP = "predict/Placeholder:0"
PREDICT_LENGTH = "predict/PlaceholderWithDefault:0"
PREDICT_LOGITS = ("predict/conv2d/Sigmoid:0", "predict/conv2d_13/Sigmoid:0")
PREDICT_LATEFEATS = "predict/csse_out_mul/mul:0"                        # job.py:1808
PREDICT_EARLYFEATS = "predict/gru_drop/drop_block2d/cond/Merge:0"       # job.py:1809
SUPERRESOLVE_INP = "superresolve/Placeholder:0"
SUPERRESOLVE_INP_BILINEAR = "superresolve/Placeholder_1:0"
SUPERRESOLVE_LOGITS = "superresolve/Add_2:0"


def _name(t):
    return t if isinstance(t, str) else getattr(t, "name", str(t))


class TTCSession:
    """Stands in for the `tf.Session` objects of the job script.  One session serves both graphs:

        sess.run(predict_logits, feed_dict={predict_inp: x[1,L+1,W,W,17], predict_length: [L]})
        sess.run([superresolve_logits], feed_dict={superresolve_inp: a, superresolve_inp_bilinear: b})

    Feeds are matched by tensor NAME (strings or objects with a `.name`), as Session.run does.
    """

    def __init__(self, weights=None, win_in=SIZE + 14, length=LEN, max_windows=36, device=0, zoneout=0.75,
                 dsen2_weights="package", precision="fp32", win_rows=0, one_term_layers=None, fp32_conv_form=0, dsen2_precision=None,
                 two_term_layers=0, budget=5e-4, calibration_windows=None):
        """precision: "fp32" = exact fp32 MFMA chains (default); "fp16" / "bf16" = the 16-bit engine (conv inputs stored
        as hi + lo 16-bit pairs, fp32 accumulate; per layer three split products or one, `one_term_layers` bit mask as in
        ttc.h -- default 0: EVERY layer multiplies three products; any single layer on one plain fp16 product measured
        3.0e-3 .. 2.7e-2 max |dprob| on a real tile, outside the 1e-3 contract).
        fp32_conv_form (precision fp32): 0 = fastest (Winograd F(4x4,3x3) / F(2x2,3x3)), 1 = F(2x2,3x3) at most, 2 = direct only: ttc.h.
        dsen2_precision: None = the session's precision; "fp16" / "bf16" = run the DSen2 super-resolution convs on the 16-bit engine (hi + lo
        pairs, three products: <= 1e-5 on reflectance for fp16) inside an fp32 session -- 2.4 instead of 5.0 ms per tile (ttc.h).
        two_term_layers (precision fp16): bit mask of layers that multiply x_hi * (w_hi + w_lo) only (ttc.h; 3 = both ConvGRU convs).
        precision="auto" (+ budget, calibration_windows): the fp16 engine with the product count of every conv layer CALIBRATED for these
        weights -- `calibration_windows` [n, L+1, W, W, 17] (real, normalised model inputs: e.g. the model feed of a tile the job already
        processed) run through an fp32 context and through candidate maps, and the cheapest map whose probabilities stay within `budget`
        of the fp32 engine's is kept (ttc_calibrate_precision; `self.calibration` holds the report).  Without windows the session starts on
        three products everywhere and `calibrate(windows)` can be called later."""
        auto = precision == "auto"
        if auto:
            precision, one_term_layers, two_term_layers = "fp16", 0, 0
        prec = _lib.PRECISIONS.get(precision, precision)
        # win_rows: rows of a non-square window (the 220 x 684 border graph of resegment_tiles_wide.py); 0 = square
        self.ctx = _lib.Context(win_in=win_in, length=length, max_windows=max_windows, device=device, zoneout=zoneout,
                                precision=prec, win_rows=win_rows, one_term_layers=one_term_layers, fp32_conv_form=fp32_conv_form,
                                dsen2_precision=dsen2_precision, two_term_layers=two_term_layers)
        self.win_in, self.length = win_in, length
        self._geom = dict(win_in=win_in, length=length, max_windows=max_windows, device=device, zoneout=zoneout, win_rows=win_rows)
        self._weights = None
        self.calibration = None
        if weights is not None:
            self._weights = _weights.validate(dict(weights))
            self.ctx.load_weights(self._weights)
        if dsen2_weights == "package":
            dsen2_weights = _weights.load_dsen2()
        if dsen2_weights is not None:
            self.ctx.load_dsen2_weights(dict(dsen2_weights))
        if auto and calibration_windows is not None:
            self.calibrate(calibration_windows, budget)

    def calibrate(self, windows, budget=5e-4):
        """ttc_calibrate_precision for this (16-bit) session: an fp32 context with the same weights is built for the duration of the call.
        windows [n <= max_windows, L+1, W, W, 17] normalised model inputs.  -> the report (also kept as self.calibration); the map stays applied."""
        if self._weights is None:
            raise RuntimeError("TTCSession.calibrate: the session has no model weights")
        ref = _lib.Context(precision=0, **self._geom)
        try:
            ref.load_weights(self._weights)
            self.calibration = self.ctx.calibrate_precision(ref, windows, budget)
        finally:
            ref.close()
        return self.calibration

    # -- tf.Session.run look-alike ---------------------------------------------------------
    def run(self, fetches, feed_dict=None):
        feeds = {_name(k): v for k, v in (feed_dict or {}).items()}
        single = not isinstance(fetches, (list, tuple))
        outs = []
        for f in ([fetches] if single else fetches):
            n = _name(f)
            if n in PREDICT_LOGITS or n.endswith("Sigmoid:0"):
                x = np.asarray(feeds[PREDICT_INP], dtype=np.float32)
                if PREDICT_LENGTH in feeds and int(np.asarray(feeds[PREDICT_LENGTH]).ravel()[0]) != self.length:
                    raise ValueError("predict_length does not match the session's ConvGRU length")
                outs.append(self.ctx.forward_windows(x).cpu().numpy()[..., np.newaxis])
            elif n in (PREDICT_LATEFEATS, PREDICT_EARLYFEATS):
                x = np.asarray(feeds[PREDICT_INP], dtype=np.float32)
                _, early, late = self.ctx.forward_taps(x, early=n == PREDICT_EARLYFEATS, late=n == PREDICT_LATEFEATS)
                outs.append((early if n == PREDICT_EARLYFEATS else late).cpu().numpy())
            elif n == SUPERRESOLVE_LOGITS or n.endswith("Add_2:0"):
                outs.append(self.ctx.dsen2_forward(feeds[SUPERRESOLVE_INP], feeds[SUPERRESOLVE_INP_BILINEAR]).cpu().numpy())
            else:
                raise KeyError(f"TTCSession cannot fetch {n!r}")
        return outs[0] if single else outs

    def close(self):
        self.ctx.close()


# ------------------------------------------------------------------------------------------
def to_float32(arr, sess):
    """src/tof/tof_downloading.py:64-72 on the device: uint16 -> float32 / 65535 (floats pass through)."""
    if isinstance(arr, np.ndarray) and np.issubdtype(arr.dtype, np.floating):
        return sess.ctx._dev(arr.astype(np.float32), sess.ctx.torch.float32)
    return sess.ctx.to_float32(arr)


def to_int16(arr, sess):
    """src/tof/tof_downloading.py:51-61: trunc(clip(x, 0, 1) * 65535) as uint16 (numpy, host copy)."""
    return sess.ctx.to_int16(arr).cpu().numpy().view(np.uint16)


def float_to_int16(arr, sess, precision=1000):
    """job.py:174-180 on the device; returns a numpy int16 array."""
    return sess.ctx.float_to_int16(arr, precision).cpu().numpy()


def predict_features(subtile, sess, size=SIZE):
    """The --gen_feats branch of process_subtiles (job.py:1429-1445) for one window: the early (bi-ConvGRU) and late
    (last U-Net block) feature maps, first 32 channels each, centre-cropped to `size` like predict_subtile (:360-362),
    quantised with float_to_int16 and concatenated -> int16 [size, size, 64].  Also returns the probabilities."""
    x = np.asarray(subtile, dtype=np.float32)[np.newaxis]
    probs, early, late = sess.ctx.forward_taps(x)
    clip = (early.shape[1] - size) // 2
    early = early[0, clip:early.shape[1] - clip, clip:early.shape[2] - clip, :32] if clip > 0 else early[0, ..., :32]
    late = late[0, ..., :32]
    feats = sess.ctx.torch.cat([sess.ctx.float_to_int16(early.contiguous()), sess.ctx.float_to_int16(late.contiguous())], -1)
    return probs[0].cpu().numpy(), feats.cpu().numpy()


def superresolve_large_tile(arr, sess):
    """src/download_and_predict_job.py:95-147, the call of :2001: `s2[..., :10] = superresolve_large_tile(s2[..., :10], sess)`.
    arr: (T, X, Y, 10) float32 whose bands 4.. were bilinearly upsampled -- a numpy array (mutated in place AND returned, like
    the reference's) or a cuda tensor (refined in place on the device, returned).  ONE ttc_superresolve_tile call: the 110-px
    window tiling with its 4-px reflect pad, the x_end / y_end copies, the never-refined strip (:133-143) and the double pass
    on the 110 x 42 patch all happen on the device; `sess` is the TTCSession that holds the DSen2 weights (TTCSession.run on
    the superresolve tensor names stays available for callers that keep the reference's own Python loop)."""
    ctx, t = sess.ctx, sess.ctx.torch
    if isinstance(arr, t.Tensor):
        if not arr.is_cuda:                                  # a host tensor takes the numpy route (its data_ptr() is not a device address)
            a = arr.detach().numpy()
            if a.ndim != 4 or a.shape[-1] != 10 or a.dtype != np.float32:
                raise ValueError(f"superresolve_large_tile: expected a float32 (T, X, Y, 10) tensor, got {a.dtype} {tuple(a.shape)}")
            superresolve_large_tile(a, sess)                 # shares the tensor's memory: refined in place
            return arr
        if arr.device.index != ctx.device:
            raise ValueError(f"superresolve_large_tile: the tensor lives on {arr.device}, the session on cuda:{ctx.device}")
        if arr.dtype != t.float32 or not arr.is_contiguous() or arr.dim() != 4 or int(arr.shape[-1]) != 10:
            raise ValueError(f"superresolve_large_tile: a cuda tensor must be contiguous float32 [T, X, Y, 10], got {arr.dtype} {tuple(arr.shape)}"
                             " (slice the band axis and call .contiguous() for the job's 17-channel stack)")
        ctx.superresolve_tile(arr, quirks=True)
        return arr
    a = np.asarray(arr)
    if a.ndim != 4 or a.shape[-1] != 10:
        raise ValueError(f"superresolve_large_tile: expected (T, X, Y, 10), got {a.shape}")
    d = ctx._dev(np.ascontiguousarray(a, dtype=np.float32), t.float32)
    ctx.superresolve_tile(d, quirks=True)
    out = d.cpu().numpy()
    if isinstance(arr, np.ndarray) and arr.dtype == np.float32:
        arr[...] = out                                   # the reference writes into its argument (views included)
        return arr
    return out


def sentinel1_to_db(s1_u16, sess):
    """The Sentinel-1 preparation of process_tile (src/download_and_predict_job.py:699-708) + convert_to_db
    (:74-89): uint16 [T, X, Y, 2] -> cuda float32."""
    return sess.ctx.s1_to_db(s1_u16)


def normalize_subtile(subtile, ctx=None):
    """job.py:316-325, in place on [..., 17].  (Pure float32 numpy: used only on the
    predict_subtile drop-in path where the caller normalises itself; the per-tile path
    normalises inside k_assemble.)"""
    for band in range(subtile.shape[-1]):
        mins, maxs = min_all[band], max_all[band]
        subtile[..., band] = np.clip(subtile[..., band], mins, maxs)
        subtile[..., band] = (subtile[..., band] - (maxs + mins) / 2) / ((maxs - mins) / 2)
    return subtile


def predict_subtile(subtile, sess, op=PREDICT_LOGITS[0], size=SIZE):
    """job.py:328-369: [L+1, W, W, 17] normalised window -> [size, size] float32 (255 = all-zero input)."""
    subtile = np.asarray(subtile)
    if np.sum(subtile) != 0:
        if not isinstance(subtile.flat[0], np.floating):
            assert np.max(subtile) > 1
            subtile = subtile / 65535.
        batch_x = subtile[np.newaxis].astype(np.float32)
        lengths = np.full((batch_x.shape[0]), sess.length)
        preds = sess.run(op, feed_dict={PREDICT_INP: batch_x, PREDICT_LENGTH: lengths})
        preds = preds.squeeze()
        clip = (preds.shape[0] - size) // 2
        if clip > 0:
            preds = preds[clip:-clip, clip:-clip]
        return np.float32(preds)
    return np.full((size, size), 255)


def window_grid(X, Y, size=SIZE, n_rows=6):
    """Output-window origins (folder_x, folder_y) in iteration order (job.py:1295-1316)."""
    gx = int(np.ceil((X - size) / (n_rows - 1)))
    gy = int(np.ceil((Y - size) / (n_rows - 1)))
    xs = list(range(0, X - size, gx)) + [X - size]
    ys = list(range(0, Y - size, gy)) + [Y - size]
    return [(x, y) for x in xs for y in ys]


def _process_subtiles_device(s2, dates, interp, s1, dem, sess, size, want_raw=False):
    """Device-resident core of process_subtiles: -> (windows cuda [n,size,size], raw | None, [(fx, fy)])."""
    ctx, t = sess.ctx, sess.ctx.torch
    s2d = ctx._dev(s2, t.float32)
    if isinstance(s2, t.Tensor) and s2d.data_ptr() == s2.data_ptr():
        s2d = s2d.clone()                                                  # the NaN repair below is in place
    T, X, Y = int(s2d.shape[0]), int(s2d.shape[1]), int(s2d.shape[2])
    dates = np.asarray(dates).copy()
    ctx.tile_fix_missing(s2d, do_nan=True, do_zero_one=False)              # interpolate_na_vals, job.py:1149
    counts = ctx.tile_missing_counts(s2d)                                   # id_missing_px(arr, 10), job.py:1032
    keep = counts < (X ** 2) / 10
    wmat = np.zeros((12, T), dtype=np.float32)
    if keep.sum() > 0:
        wmat[:, keep] = temporal.temporal_operator(dates[keep])
    windows, raw = ctx.process_subtiles(s2d, wmat, keep.astype(np.int32), interp, s1, dem, min_all, max_all, size,
                                        n_dates_ok=int(keep.sum()), want_raw=want_raw)
    return windows, raw, window_grid(X, Y, size)


def process_subtiles(x, y, s2, dates, interp, s1, dem, sess, bbx=None, size=SIZE, train_bbx=None,
                     return_raw=False):
    """job.py:1125-1483 numeric core on the GPU.

    s2 [T,X,Y,10] f32, dates [T], interp [T,X,Y] f32, s1 [12,X,Y,2] f32, dem [X,Y] f32 (numpy or
    cuda tensors), `sess` a TTCSession.  Returns {(folder_y, folder_x): preds[size,size] float32}
    -- exactly the arrays the reference writes to processed/{folder_y}/{folder_x}.npy.
    x, y, bbx, train_bbx are accepted for signature compatibility (file naming / GeoTIFF only).
    """
    windows, raw, grid = _process_subtiles_device(s2, dates, interp, s1, dem, sess, size, return_raw)
    wins = windows.cpu().numpy()
    out = {(fy, fx): wins[i] for i, (fx, fy) in enumerate(grid)}
    if return_raw:
        rw = raw.cpu().numpy()
        return out, {(fy, fx): rw[i] for i, (fx, fy) in enumerate(grid)}
    return out


def load_mosaic_predictions(windows, depth=1, sess=None, size=SIZE, return_float=False):
    """job.py:1515-1641, from the dict returned by process_subtiles (depth == 1) or a dict of int16 feature windows
    [size, size, depth] (depth > 1, the --gen_feats branch, :1552-1592).
    Returns uint8 [max_y + size, max_x + size] -- transposed like the reference (job.py:1578) -- or int16
    [depth, max_y + size, max_x + size]."""
    if sess is None:
        raise ValueError("load_mosaic_predictions needs the TTCSession whose GPU holds the windows")
    keys = sorted(windows.keys())
    if depth != 1:
        stack = np.stack([np.asarray(windows[k], dtype=np.int16)[..., :depth] for k in keys])
        xy = np.array([[fx, fy] for (fy, fx) in keys], dtype=np.int32)
        rows = int(max(k[0] for k in keys) + size)
        cols = int(max(k[1] for k in keys) + size)
        return sess.ctx.mosaic_features(stack, xy, size, depth, rows, cols).cpu().numpy()
    stack = np.stack([np.asarray(windows[k], dtype=np.float32) for k in keys])
    xy = np.array([[fx, fy] for (fy, fx) in keys], dtype=np.int32)
    rows = int(max(k[0] for k in keys) + size)
    cols = int(max(k[1] for k in keys) + size)
    u8, f32 = sess.ctx.mosaic(stack, xy, size, rows, cols, want_float=return_float)
    if return_float:
        return u8.cpu().numpy(), f32.cpu().numpy()
    return u8.cpu().numpy()


# ------------------------------------------------------------------------------------------
# cloud / shadow gap-fill (src/preprocessing/cloud_removal.py)
def identify_clouds_shadows(img, dem, bbx=None, sess=None, forest_mask=None, urban_masks=None):
    """cloud_removal.py:1215-1677 (+ detect_pfcp, :1109-1212) on the device.  Same positional arguments as the reference;
    `bbx` is only used there to window the two ESA-WorldCover rasters (forestmask.tif / urbanmask.tif), which the caller
    passes here already cut and resized: forest_mask [X, Y] == adjust_cloudmask_in_forests(...), urban_masks = (core, near)
    == the two masks mask_nonurban_areas builds.  None reproduces the reference's behaviour without the rasters.
    -> (clouds float32 [T, X, Y], fcps bool [T, X, Y])"""
    if sess is None:
        raise ValueError("identify_clouds_shadows needs a TTCSession")
    clouds, fcps = sess.ctx.identify_clouds_shadows(np.asarray(img, dtype=np.float32), np.asarray(dem, dtype=np.float32),
                                                    forest_mask, urban_masks)
    return clouds.cpu().numpy(), fcps.cpu().numpy().astype(bool)


def reference_sampler(evi, rng=None):
    """The reference's EVI-stratified row sample (cloud_removal.py:453-500), replayed with the stdlib global RNG:
    2 % tails repeated x10, five quintile strata truncated to n//5 after random.shuffle, shuffled again.
    Pin random.seed() before remove_cloud_and_shadows to reproduce the reference bit for bit."""
    import random
    rng = rng or random
    n_rows = len(evi)
    n_i = min(90000, n_rows) // 5
    b2, b20, b40, b60, b80, b98 = (np.percentile(evi, q) for q in (2, 20, 40, 60, 80, 98))
    strata = [np.flatnonzero(evi < b20), np.flatnonzero((evi >= b20) & (evi < b40)), np.flatnonzero((evi >= b40) & (evi < b60)),
              np.flatnonzero((evi >= b60) & (evi < b80)), np.flatnonzero(evi >= b80)]
    p2, p98 = np.repeat(np.flatnonzero(evi < b2), 10), np.repeat(np.flatnonzero(evi >= b98), 10)
    for p in [p2, p98] + strata:              # the reference's shuffle order: p2, p98, p20, p40, p60, p80, p100
        rng.shuffle(p)
    sample = np.concatenate([p2] + [q[:n_i] for q in strata] + [p98])
    rng.shuffle(sample)
    return sample[:n_rows]


def id_areas_to_interp(tiles, probs, shadows, image_dates, pfcps, sess=None):
    """cloud_removal.py:774-798 (tiles / shadows / dates / pfcps are unused by the reference too)."""
    return sess.ctx.feather(probs, closing=15, clip=True).cpu().numpy()


def remove_cloud_and_shadows(tiles, probs, shadows, image_dates, pfcps, sentinel1=None, mosaic=None, sess=None,
                             sampler="reference"):
    """cloud_removal.py:888-973 on the GPU.  tiles [T,X,Y,10] float32 numpy (modified in place, like the
    reference) or cuda tensor; returns (tiles, areas_interpolated [T,X,Y], to_remove).
    sampler = "reference": replay random.shuffle on the host (exact, but Python-speed: ~0.5 s per date);
              "expected":  deterministic expected-multiplicity weighting, entirely on the device."""
    if mosaic is not None:
        raise NotImplementedError("a caller-supplied mosaic is not used anywhere in the reference job")
    ctx, t = sess.ctx, sess.ctx.torch
    td = ctx._dev(tiles, t.float32)
    fn = reference_sampler if sampler == "reference" else None
    interp, to_remove, _ = ctx.remove_cloud_and_shadows(td, probs, pfcps, fn)
    if isinstance(tiles, np.ndarray):
        tiles[...] = td.cpu().numpy()
        return tiles, interp.cpu().numpy(), to_remove
    return td, interp, to_remove


def adjust_shape(arr, width, height):
    """job.py:260-318: crop / edge-pad the two spatial axes of [T, X, Y(, C)] (or [X, Y]) to width x height (host-side
    index plumbing on the arrays as stored, before they are uploaded)."""
    arr = np.asarray(arr)
    arr = arr[:, :, :, np.newaxis] if arr.ndim == 3 else arr
    arr = arr[np.newaxis, :, :, np.newaxis] if arr.ndim == 2 else arr
    for ax, want in ((1, width), (2, height)):
        n = arr.shape[ax]
        if n < want:
            amt = (want - n) // 2
            pad = [(0, 0)] * 4
            pad[ax] = ((1, amt) if ax == 1 else (1, 0)) if amt == 0 else (amt, amt)
            arr = np.pad(arr, pad, "edge")
    for ax, want in ((1, width), (2, height)):
        n = arr.shape[ax]
        if n > want:
            amt, even = (n - want) // 2, (n - want) % 2 == 0
            if amt == 0:
                sl = slice(1, None)
            elif even:
                sl = slice(amt, -amt)
            else:
                sl = slice(int(np.floor(amt / 2)), -int(np.ceil(amt / 2)))
            arr = arr[:, sl] if ax == 1 else arr[:, :, sl]
    return arr.squeeze()


def load_raw_tile(x, y, local_path, alloc=None, want_clouds=True):
    """The file loads at the top of process_tile (job.py:669-714): the arrays of `{local_path}{x}/{y}/raw/` as the `raw` dict
    that process_tile (below) takes.  hkl.load is replaced by the library's own HDF5 reader (ttc_read_hkl, host code);
    the Sen2Cor mask is returned at its stored 20 m resolution, the clean-up of :687-694 happens on the device.
    alloc(name) -> allocator for _lib.read_hkl (PinnedArena.allocator): the arrays the tile call uploads are then read straight
    into page-locked memory.  want_clouds=False skips the s2cloudless probabilities (18 MB inflated per T = 12 tile): the reference loads
    them (:684) but only carries them along the date steps -- the mask the gap-fill uses comes from identify_clouds_shadows (:839), so a
    tile loop that detects (predict_tiles with mask None) never reads them and the host saves a sixth of its inflate work."""
    x, y = str(int(x)), str(int(y))
    folder = f"{local_path}{x}/{y}/"
    idx = f"{x}X{y}Y"

    def rd(path, name=None):
        return _lib.read_hkl(path, alloc=alloc(name) if (alloc is not None and name is not None) else None)
    clm_file = f"{folder}raw/clouds/cloudmask_{idx}.hkl"
    return {"clouds": rd(f"{folder}raw/clouds/clouds_{idx}.hkl") if want_clouds else None,
            "clm": rd(clm_file) if os.path.exists(clm_file) else None,
            "s1": rd(f"{folder}raw/s1/{idx}.hkl", "s1"), "s2_10": rd(f"{folder}raw/s2_10/{idx}.hkl", "s2_10"),
            "s2_20": rd(f"{folder}raw/s2_20/{idx}.hkl", "s2_20"), "dem": rd(f"{folder}raw/misc/dem_{idx}.hkl", "dem"),
            "dates": rd(f"{folder}raw/misc/s2_dates_{idx}.hkl")}


class PinnedArena:
    """Page-locked buffer sets for the raw arrays of tiles that are read ahead (iter_raw_tiles) and uploaded by predict_tiles: the
    HDF5 reader inflates straight into them, so the tile loop's main thread does not copy 90 MB per tile into a staging buffer
    (measured: 10 ms of a 34 ms tile period).  Sets are handed out round-robin in tile order; a set is reused only after the tile
    that used it was FINISHED (predict_tiles keeps the raw arrays until then: a flagged tile re-runs from them)."""

    def __init__(self, torch, nsets):
        import threading
        self.t = torch
        self.sets = [dict() for _ in range(int(nsets))]
        self.free = [True] * int(nsets)
        self.cv = threading.Condition()
        self.nxt = 0

    def acquire(self):
        """-> index of the next set, waiting until the tile that last used it has been released"""
        with self.cv:
            i = self.nxt
            self.nxt = (i + 1) % len(self.sets)
            while not self.free[i]:
                if not self.cv.wait(timeout=120.0):
                    raise RuntimeError("PinnedArena: no free set after 120 s -- it needs more sets than iter_raw_tiles reads ahead "
                                       "plus the tiles predict_tiles keeps in flight (ahead + depth + 2)")
            self.free[i] = False
            return i

    def release(self, i):
        with self.cv:
            self.free[i] = True
            self.cv.notify_all()

    def allocator(self, i):
        """-> alloc(name) -> alloc(shape, dtype) for _lib.read_hkl: a numpy view of set i's pinned buffer `name`"""
        t, bufs = self.t, self.sets[i]

        def for_name(name):
            def alloc(shape, dtype):
                dtype = np.dtype(dtype)
                nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
                b = bufs.get(name)
                if b is None or b.numel() < nbytes:
                    b = bufs[name] = t.empty(max(nbytes, 1), dtype=t.uint8, pin_memory=True)
                return b.numpy()[:nbytes].view(dtype).reshape(shape)
            return alloc
        return for_name


def iter_raw_tiles(coords, local_path, workers=4, ahead=None, arena=None, want_clouds=True):
    """Iterator over load_raw_tile(x, y, local_path) for (x, y) in coords, read AHEAD by a small thread pool: the HDF5 reader is
    host C++ behind ctypes (the GIL is released for the duration of the call), so `workers` tiles are parsed / inflated in
    parallel while the GPU works on earlier ones -- feed it to predict_tiles.  At most `ahead` (default 2 x workers) tiles are
    resident.  Order is preserved; a failed read raises when its tile is reached.
    arena (PinnedArena with MORE sets than `ahead` + the tiles predict_tiles keeps in flight): the uploaded arrays are read into
    its page-locked sets (raw["_arena_set"] names the set; predict_tiles(arena=...) releases it when the tile is finished).
    A plain function that returns a generator: `arena.ahead` is set HERE, before the first next(), so that predict_tiles' up-front
    size check sees the real read-ahead; closing the generator (or abandoning the loop: predict_tiles closes it on any
    exception) waits for the reads in flight and hands their sets back to the arena."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    coords = list(coords)
    ahead = int(ahead) if ahead else 2 * int(workers)
    if ahead < 1:
        raise ValueError("iter_raw_tiles: ahead must be >= 1")
    if arena is not None:
        arena.ahead = ahead                      # predict_tiles checks the arena's size against ahead + its own depth
        if len(arena.sets) < ahead + 1:
            raise ValueError(f"iter_raw_tiles: the PinnedArena has {len(arena.sets)} sets for {ahead} tiles read ahead (+ the consumer's own)")

    def load(xy, aset):
        raw = load_raw_tile(xy[0], xy[1], local_path, alloc=arena.allocator(aset) if arena is not None else None, want_clouds=want_clouds)
        if arena is not None:
            raw["_arena_set"] = aset
        return raw

    def gen():
        q = deque()                             # (future, arena set | None) of the tiles read ahead and not yet handed out

        def submit(pool, xy):                   # sets are taken in tile order, on the consumer's thread
            aset = arena.acquire() if arena is not None else None
            try:
                q.append((pool.submit(load, xy, aset), aset))
            except BaseException:
                if aset is not None:
                    arena.release(aset)
                raise
        with ThreadPoolExecutor(max_workers=int(workers)) as pool:
            try:
                it = iter(coords)
                for xy in it:
                    submit(pool, xy)
                    if len(q) >= ahead:
                        break
                while q:
                    fut, aset = q[0]
                    raw = fut.result()           # a failed read raises here with its set still in q: released below
                    q.popleft()
                    nxt = next(it, None)
                    if nxt is not None:
                        submit(pool, nxt)
                    yield raw                    # from here on the set belongs to the consumer (predict_tiles releases it)
            finally:
                for fut, aset in q:              # abandoned (GeneratorExit) or failed: un-yielded reads give their sets back
                    fut.cancel()
                    try:
                        fut.result()             # a read that already started still writes into the set: wait for it
                    except BaseException:
                        pass
                    if aset is not None:
                        arena.release(aset)
                q.clear()
    return gen()


def process_tile(raw, sess, forest_mask=None, urban_masks=None, make_shadow=True, sampler="reference", cloudshad=None):
    """The numeric flow of process_tile (job.py:641-995) on the device, from the arrays it loads from temp/raw/* :
    raw = {"s2_10": uint16 [T, X, Y, 4], "s2_20": uint16 [T, X/2, Y/2, 6], "s1": uint16 [12, X, Y, 2], "dem": float [X, Y] (m),
           "dates": int [T], "clouds": float [T, X, Y] (s2cloudless, only kept in step), "clm": 20 m Sen2Cor mask or None}.
    Host code takes the DECISIONS the reference takes (which dates to drop, when to re-run the detection); every raster
    operation runs through the C ABI.  -> (sentinel2 cuda [T', X, Y, 10], dates, interp cuda, s1 cuda, dem cuda (/90),
    cloudshad cuda, snow cuda uint8), T' <= T.
    cloudshad [T, X, Y] (optional): a GIVEN cloud + shadow mask replaces identify_clouds_shadows (and its re-runs after dates
    are dropped: the surviving dates of the given mask) -- the staged twin of ttc_predict_tile without TTC_TILE_DETECT."""
    ctx, t = sess.ctx, sess.ctx.torch
    s2_20 = np.asarray(raw["s2_20"])
    if s2_20.ndim == 3:
        s2_20 = s2_20[np.newaxis]
    width, height = s2_20.shape[1] * 2, s2_20.shape[2] * 2
    s2_10 = adjust_shape(raw["s2_10"], width, height)
    if s2_10.ndim == 3:
        s2_10 = s2_10[np.newaxis]
    # the reference scales Sentinel-1 (the per-image median over the image AS STORED) and median-filters the DEM BEFORE adjust_shape
    # re-grids them (:699-721): same order here, the re-gridding on the device (ttc_adjust_shape)
    s1 = ctx.adjust_shape(ctx.s1_to_db(np.ascontiguousarray(raw["s1"])), width, height)            # :699-708, :718
    dem = ctx.adjust_shape(ctx.median5(np.ascontiguousarray(raw["dem"], dtype=np.float32)), width, height).contiguous()   # :713, :721
    clm = ctx.sen2cor_clean(np.asarray(raw["clm"], dtype=np.float32)) if raw.get("clm") is not None else None      # :688-697
    dates = np.array(raw["dates"], copy=True)
    if s2_10.shape[0] != s2_20.shape[0] or len(dates) != s2_20.shape[0]:
        raise ValueError(f"process_tile: s2_10 holds {s2_10.shape[0]} dates, s2_20 {s2_20.shape[0]}, the date list {len(dates)}")
    if tuple(s2_10.shape[1:3]) != (width, height):
        raise ValueError(f"process_tile: adjust_shape (job.py:260-310) cannot bring the 10 m bands {tuple(np.asarray(raw['s2_10']).shape)} onto the "
                         f"{width} x {height} grid of the 20 m stack (an odd difference of 3 or more: the reference raises there too)")
    if cloudshad is not None and tuple(np.shape(cloudshad)) != (s2_20.shape[0], width, height):
        raise ValueError(f"process_tile: the given cloud / shadow mask is {tuple(np.shape(cloudshad))}, the tile is {(s2_20.shape[0], width, height)}")
    clouds = np.array(raw["clouds"], copy=True) if raw.get("clouds") is not None else np.zeros((len(dates), 1, 1), np.float32)
    s2 = ctx.upsample_20m(ctx.to_float32(np.ascontiguousarray(s2_10)), ctx.to_float32(np.ascontiguousarray(s2_20)))  # :727-782
    interp = None
    given = ctx._dev(cloudshad, t.float32).clone() if cloudshad is not None else None

    def drop(idx):
        nonlocal clouds, dates, s2, clm, interp, given
        keep = np.setdiff1d(np.arange(len(dates)), idx)
        sel = t.as_tensor(keep, device=s2.device)
        if given is not None:
            given = given.index_select(0, sel).contiguous()
        if clouds.shape[0] == len(dates):
            clouds = np.delete(clouds, idx, axis=0)
        dates = np.delete(dates, idx)
        s2 = s2.index_select(0, sel).contiguous()
        if clm is not None:
            clm = clm.index_select(0, sel).contiguous()
        if interp is not None:
            interp = interp.index_select(0, sel).contiguous()

    X = int(s2.shape[1])
    counts = ctx.tile_missing_counts(s2)                                                       # id_missing_px(., 2), :786
    missing = np.argwhere(counts >= (X ** 2) / 2).flatten()
    if len(missing) > 0:
        drop(missing)
    snow, snow_frac = ctx.snow_map(s2)                                                         # :799-821
    snowy = np.argwhere(snow_frac > 0.25).flatten()
    if len(snowy) > 10:
        drop(snowy)
    # interpolate_missing_vals (:833) is the identity in the reference (its guard can never be true)
    if not make_shadow:
        z = t.zeros(s2.shape[:3], dtype=t.float32, device=s2.device)
        return ctx.clip01(s2), dates, z, s1, ctx.divide(dem, 90.0), z.clone(), snow

    def detect(first):
        if given is not None:
            cs, fc = given.clone(), None
        else:
            cs, fc = ctx.identify_clouds_shadows(s2, dem, forest_mask, urban_masks)           # :839
        if clm is not None:
            ctx.merge_cloud_masks(cs, clm, fc if first else None)                             # :841-846 / :880-884
        return cs, fc

    cloudshad, fcps = detect(True)
    interp = ctx.feather(cloudshad, closing=15, clip=True)                                     # id_areas_to_interp, :848
    for rnd in range(3):                                                                       # :866-921
        heavy = np.argwhere(ctx.fraction_positive(interp) > 0.9).flatten()
        if len(heavy) > 0:
            drop(heavy)
            cloudshad, fcps = detect(False)
            if rnd < 2:
                interp = ctx.feather(cloudshad, closing=15, clip=True)
    interp = ctx.feather(cloudshad, closing=15, clip=True)
    fn = reference_sampler if sampler == "reference" else None
    interp, to_remove, _ = ctx.remove_cloud_and_shadows(s2, cloudshad, fcps, fn)               # :935-945
    if len(to_remove) > 0:                                                                     # :965-983
        drop(np.asarray(to_remove))
        cloudshad, fcps = detect(False)
        interp = ctx.feather(cloudshad, closing=15, clip=True)
    dem_m = dem.clone()                      # the detector above used metres; the model wants dem / 90 (:993)
    return ctx.clip01(s2), dates, interp, s1, ctx.divide(dem_m, 90.0), cloudshad, snow


def predict_tile(s2, dates, interp, s1, dem, sess, size=SIZE, to_host=True):
    """One call, device-resident between the stages: cloud-free tile stack ->
    (float32 percent raster with NaN no-data, uint8 product), both [Y, X] like load_mosaic_predictions."""
    windows, _, grid = _process_subtiles_device(s2, dates, interp, s1, dem, sess, size)
    xy = np.array(grid, dtype=np.int32)
    rows, cols = int(xy[:, 1].max() + size), int(xy[:, 0].max() + size)
    u8, f32 = sess.ctx.mosaic(windows, xy, size, rows, cols, want_float=True)
    if to_host:
        return f32.cpu().numpy(), u8.cpu().numpy()
    return f32, u8


def predict_tile_raw_checked(raw, mask, sess, size=SIZE, to_host=True, sampler="expected", want_status=False):
    """The job's per-tile chain (job.py:1995-2020) from the arrays of temp/raw, fast path first:

      1. ttc_predict_tile -- ONE enqueue, no host round trip -- takes process_tile's rare data-dependent decisions
         speculatively and reports them in its status words (include/ttc.h);
      2. if any is set (a date that cannot be aligned, dates the gap-fill marks fully interpolated, a date with half its
         pixels missing, > 10 snowy dates, a > 90 % clouded date), the raster of step 1 is NOT the reference's: the tile is
         re-run through the staged mirror -- process_tile (given mask) -> superresolve_large_tile -> predict_tile -- which
         takes those decisions on the host exactly like the reference.

    raw: {"s2_10" uint16 [T, X, Y, 4], "s2_20" uint16 [T, X/2, Y/2, 6], "s1" uint16 [12, X, Y, 2], "dem" float [X, Y] in
    METRES, "dates" int [T]} (numpy); mask [T, X, Y] the cloud + shadow mask.  sampler: "expected" (deterministic, what
    the single call uses) or "reference" for the staged re-run only.
    -> (float32 percent raster with NaN no-data, uint8 product) [Y, X] like load_mosaic_predictions (+ the int32[4] status
    words and whether the staged path ran, with want_status)."""
    ctx, t = sess.ctx, sess.ctx.torch
    dem90 = ctx.divide(ctx.median5(np.ascontiguousarray(raw["dem"], dtype=np.float32)), 90.0)          # job.py:713, :993
    u8, f32, _, status = ctx.predict_tile_raw(raw["s2_10"], raw["s2_20"], raw["s1"], dem90, mask, raw["dates"], min_all, max_all,
                                              size, want_float=True)
    st = status.cpu().numpy()                                                  # waits for the stream
    staged = ctx.tile_needs_staged(st)
    if staged:
        s2, dates, interp, s1, dem, _, _ = process_tile(dict(raw, clouds=None, clm=None), sess, sampler=sampler, cloudshad=mask)
        ctx.superresolve_tile(s2, quirks=True)                                                         # job.py:2001
        f32, u8 = predict_tile(s2, dates, interp, s1, dem, sess, size=size, to_host=False)
    if to_host:
        f32, u8 = f32.cpu().numpy(), u8.cpu().numpy()
    return (f32, u8, st, staged) if want_status else (f32, u8)


class _PinnedStager:
    """Page-locked staging of a tile's raw arrays: numpy -> pinned buffer (host memcpy) -> non_blocking H2D on the session's
    stream, so enqueueing tile k + K never waits for tile k's kernels (a pageable .to(device) blocks the host behind everything
    queued on that stream).  `slots` buffer sets are reused round-robin; a set is rewritten only after the H2D copies that
    last read it have finished (event)."""

    def __init__(self, torch, device, slots):
        self.t, self.dev = torch, f"cuda:{device}"
        self.sets = [dict() for _ in range(slots)]
        self.done = [None] * slots

    def upload(self, slot, arrays, stream):
        """arrays: {name: numpy array}; uint16 travels as its int16 view.  -> {name: cuda tensor} (valid on `stream`)"""
        t = self.t
        if self.done[slot] is not None:
            self.done[slot].synchronize()
        out, bufs = {}, self.sets[slot]
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype == np.uint16:
                a = a.view(np.int16)
            ta = t.from_numpy(a)
            if ta.is_pinned():                                   # read straight into page-locked memory (PinnedArena): no staging copy
                with t.cuda.stream(stream):
                    out[name] = ta.to(self.dev, non_blocking=True)
                continue
            b = bufs.get(name)
            if b is None or tuple(b.shape) != a.shape or b.numpy().dtype != a.dtype:
                b = bufs[name] = t.empty(a.shape, dtype=ta.dtype, pin_memory=True)
            b.numpy()[...] = a
            with t.cuda.stream(stream):
                out[name] = b.to(self.dev, non_blocking=True)
        ev = t.cuda.Event()
        ev.record(stream)
        self.done[slot] = ev
        return out


def predict_tiles(tiles, sessions, size=SIZE, sampler="expected", to_host=True, want_status=False, depth=None, timings=None,
                  on_result=None, arena=None):
    """The job's TILE LOOP (job.py:1869-2091 processes one tile after the other) over a sequence of tiles, pipelined on one GPU:
    `sessions` = 1 .. K TTCSession of the same device; tile k is enqueued on session k % K, each session on its own HIP stream,
    with ONE ttc_predict_tile call and no host wait, so K tiles are in flight (bench.py: two saturate an MI355X -- one tile's
    latency-bound preprocessing runs under the other's convolutions).  A tile's status words are read only when the pipeline
    is `depth` (default 2 K) tiles ahead; a tile they flag is re-run through the staged mirror exactly like
    predict_tile_raw_checked does, on its own session, before its result is returned.

    tiles: iterable of (raw, mask) as predict_tile_raw_checked takes them (may be a generator that loads files lazily:
    at most `depth` tiles are resident).  mask None = run the multi-temporal cloud / shadow DETECTION inside the call
    (TTC_TILE_DETECT; what the job does: job.py:839), a flagged tile then re-runs process_tile with its own detection.
    A tile that comes with a Sen2Cor mask file (raw["clm"], job.py:685-697) and no given mask does NOT take the single call --
    ttc_predict_tile has no input for it -- but the staged chain process_tile (which merges the cleaned mask into the detected
    one, :841-846) -> superresolve -> predict_tile, and is reported as staged.
    on_result(k, result): called in input order as soon as tile k is finished (e.g. write_tif), while later tiles are in flight.
    timings: dict that receives host seconds spent staging uploads / waiting for results.  Raw arrays are staged through page-locked buffers and uploaded with non-blocking
    copies on the tile's stream (_PinnedStager), so the host never waits behind a stream's queued kernels while enqueueing; arrays that
    already live in page-locked memory (iter_raw_tiles(arena=PinnedArena)) are uploaded from where they are, and `arena` gets each
    tile's set back when the tile is finished.  -> list of (float32 percent raster, uint8 product[, status int32[4], staged]) in
    input order, numpy with to_host else cuda tensors."""
    from collections import deque
    sessions = list(sessions)
    if not sessions:
        raise ValueError("predict_tiles needs at least one session")
    t = sessions[0].ctx.torch
    dev = sessions[0].ctx.device
    if any(sx.ctx.device != dev for sx in sessions):
        raise ValueError("predict_tiles: every session must live on the same device (shard tiles across ranks with shard.py)")
    streams = [t.cuda.Stream(device=dev) for _ in sessions]
    depth = int(depth) if depth else 2 * len(sessions)
    # iter_raw_tiles(arena=...) has set arena.ahead by the time it RETURNS (it is not a generator function), so this check sees it
    if arena is not None and len(arena.sets) < getattr(arena, "ahead", 0) + depth + 2:
        raise ValueError(f"predict_tiles: the PinnedArena has {len(arena.sets)} sets but the loop keeps up to "
                         f"{getattr(arena, 'ahead', 0)} tiles read ahead + {depth} in flight: it needs >= {getattr(arena, 'ahead', 0) + depth + 2}")
    pending, results = deque(), []
    stager = _PinnedStager(t, dev, 2 * len(sessions))
    import time as _time

    host_sets = {}                                # slot -> (status, u8, f32) page-locked result buffers, reused round-robin

    def post_results(k, st, u8, f32, status):
        """queue the tile's results into page-locked host buffers behind its kernels and record an EVENT: the loop later waits for that event,
        not for the stream -- hipStreamSynchronize would also wait for the NEXT tile already queued on the same stream (measured: 19.4 vs 15.2 ms
        per tile with three streams, tools/probes/job_overlap_probe.py modes H / I)"""
        slot = k % (depth + 2)
        hb = host_sets.get(slot)
        if hb is None or hb[1].shape != u8.shape or hb[2].shape != f32.shape:
            hb = host_sets[slot] = (t.empty(4, dtype=t.int32, pin_memory=True), t.empty(u8.shape, dtype=u8.dtype, pin_memory=True),
                                    t.empty(f32.shape, dtype=f32.dtype, pin_memory=True))
        with t.cuda.stream(st):
            hb[0].copy_(status, non_blocking=True)
            if to_host:
                hb[1].copy_(u8, non_blocking=True)
                hb[2].copy_(f32, non_blocking=True)
            return hb, st.record_event()

    def finish():
        k, raw, mask, u8, f32, status = pending.popleft()
        sess, st = sessions[k % len(sessions)], streams[k % len(sessions)]
        t0 = _time.perf_counter()
        with t.cuda.stream(st):
            if status is None:                                                # went through the staged chain at enqueue time (Sen2Cor mask)
                words, staged = np.zeros(4, dtype=np.int32), True
                if to_host:
                    f32, u8 = f32.cpu().numpy(), u8.cpu().numpy()
                else:
                    st.synchronize()
            else:
                hb, ev = status
                ev.synchronize()                                              # waits for THIS tile only (not for what is queued behind it)
                words = hb[0].numpy().copy()
                staged = sess.ctx.tile_needs_staged(words)
                if staged:
                    # a GIVEN mask replaces detection and Sen2Cor mask alike; without one the re-run detects and merges raw["clm"]
                    s2, dates, interp, s1, dem, _, _ = process_tile(dict(raw, clouds=None, clm=None if mask is not None else raw.get("clm")),
                                                                    sess, sampler=sampler, cloudshad=mask)
                    sess.ctx.superresolve_tile(s2, quirks=True)
                    f32, u8 = predict_tile(s2, dates, interp, s1, dem, sess, size=size, to_host=False)
                    if to_host:
                        f32, u8 = f32.cpu().numpy(), u8.cpu().numpy()
                    else:
                        st.synchronize()
                elif to_host:
                    f32, u8 = hb[2].numpy().copy(), hb[1].numpy().copy()      # the page-locked set is reused depth + 2 tiles later
        if timings is not None:
            timings["wait_d2h_host_s"] = timings.get("wait_d2h_host_s", 0.0) + _time.perf_counter() - t0
            timings["staged"] = timings.get("staged", 0) + int(staged)
        if arena is not None and isinstance(raw, dict) and raw.get("_arena_set") is not None:
            arena.release(raw["_arena_set"])                     # its H2D copies finished long ago (the tile's results are back)
        results.append((f32, u8, words, staged) if want_status else (f32, u8))
        if on_result is not None:
            on_result(k, results[-1])

    it = iter(tiles)

    def loop():
        k = -1
        while True:
            t0 = _time.perf_counter()
            try:
                raw, mask = next(it)                                     # a lazy generator reads / waits for the tile's files here
            except StopIteration:
                break
            k += 1
            if timings is not None:
                timings["next_tile_host_s"] = timings.get("next_tile_host_s", 0.0) + _time.perf_counter() - t0
            sess, st = sessions[k % len(sessions)], streams[k % len(sessions)]
            ctx = sess.ctx
            if mask is None and isinstance(raw, dict) and raw.get("clm") is not None:
                t0 = _time.perf_counter()
                with t.cuda.stream(st):
                    s2, dates_k, interp, s1, dem, _, _ = process_tile(dict(raw, clouds=None), sess, sampler=sampler)
                    ctx.superresolve_tile(s2, quirks=True)
                    f32, u8 = predict_tile(s2, dates_k, interp, s1, dem, sess, size=size, to_host=False)
                if timings is not None:
                    timings["enqueue_host_s"] = timings.get("enqueue_host_s", 0.0) + _time.perf_counter() - t0
                pending.append((k, raw, mask, u8, f32, None))
                if len(pending) > depth:
                    finish()
                continue
            t0 = _time.perf_counter()
            arrays = {"s2_10": raw["s2_10"], "s2_20": raw["s2_20"], "s1": raw["s1"], "dem": np.asarray(raw["dem"], dtype=np.float32),
                      "dates": np.asarray(raw["dates"], dtype=np.int32)}
            if mask is not None:
                arrays["mask"] = np.asarray(mask, dtype=np.float32)
            d = stager.upload(k % (2 * len(sessions)), arrays, st)
            if timings is not None:
                timings["stage_h2d_host_s"] = timings.get("stage_h2d_host_s", 0.0) + _time.perf_counter() - t0
            t0 = _time.perf_counter()
            with t.cuda.stream(st):
                dem_m = ctx.median5(d["dem"])                                                                # job.py:713 (metres: the detector's unit)
                dem90 = ctx.divide(dem_m.clone(), 90.0)                                                      # :993
                u8, f32, _, status = ctx.predict_tile_raw(d["s2_10"], d["s2_20"], d["s1"], dem90, d.get("mask"), d["dates"], min_all, max_all,
                                                          size, dem_m=dem_m, flags=0 if mask is not None else ctx.TILE_DETECT, want_float=True)
            status = post_results(k, st, u8, f32, status)
            if timings is not None:
                timings["enqueue_host_s"] = timings.get("enqueue_host_s", 0.0) + _time.perf_counter() - t0
            pending.append((k, raw, mask, u8, f32, status))
            if len(pending) > depth:
                finish()
        while pending:
            finish()
        return results

    try:
        return loop()
    except BaseException:
        # an abandoned loop must not starve the next one on the same arena: (1) close the tile iterator -- iter_raw_tiles then waits for
        # its reads in flight and releases the sets of the tiles it read ahead; (2) wait for the streams, because the sets of the pending
        # tiles may still be the SOURCE of queued H2D copies; (3) hand those sets back
        close = getattr(it, "close", None)
        if close is not None:
            try:
                close()
            except BaseException:
                pass
        if arena is not None:
            for st in streams:
                try:
                    st.synchronize()
                except BaseException:
                    pass
            for rec in pending:
                if isinstance(rec[1], dict) and rec[1].get("_arena_set") is not None:
                    arena.release(rec[1]["_arena_set"])
        raise


def write_tif(arr, point, x, y, out_folder, suffix="_FINAL"):
    """src/downloading/io.py:229-263 without rasterio: arr [X, Y] (any dtype castable to uint8) is transposed like there and
    written as `{out_folder}{x}X{y}Y{suffix}.tif` -- LZW GeoTIFF, EPSG:4326, bounds point = [west, south, east, north]."""
    file = f"{out_folder}{str(x)}X{str(y)}Y{suffix}.tif"
    a = np.ascontiguousarray(np.asarray(arr).T.astype(np.uint8))
    return _lib.write_geotiff_u8(file, a, west=point[0], south=point[1], east=point[2], north=point[3])
