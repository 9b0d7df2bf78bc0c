"""CPU oracle for the whole per-tile chain the job runs (src/download_and_predict_job.py:1995-2020):

    process_tile (:641-995, with a given cloud / shadow mask)  ->  superresolve_large_tile (:95-147)
    ->  process_subtiles (:1125-1483)  ->  load_mosaic_predictions (:1515-1641)

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): chains the pinned restatements of the stages (restate_numpy /
restate_gapfill / restate_tile / restate_model) exactly in the reference's order, so that the product's single-call entry
(ttc_predict_tile) is checked against a SPECIFICATION of what it computes rather than against its own staged twin.
bench.py's cpu_baseline leg times these same functions (`timings`).
"""
from __future__ import annotations

import time

import numpy as np

from oracle import restate_gapfill as G
from oracle import restate_numpy as R
from oracle import restate_tile as P


class _Clock:
    def __init__(self, timings):
        self.t, self.last = timings, time.time()

    def lap(self, name):
        now = time.time()
        if self.t is not None:
            self.t[name] = self.t.get(name, 0.0) + now - self.last
        self.last = now


def _finish(s2, dates, interp, s1, dem90, net, dsen2, size, length, only_windows, clk, cache=None):
    """superresolve -> process_subtiles -> mosaic.  Returns a dict with the rounded windows, the model's raw
    (pre-mask, pre-rounding) probabilities of the windows it was run on, the model feeds, and the two rasters.
    only_windows: window indices (job.py:1295-1316 iteration order) to run the model on; the others get a constant 0.5
    (bench.py's bounded CPU sample) -- None = all."""
    if cache is not None and "s2_sr" in cache:                       # the stages before the windows do not depend on size / length
        s2 = cache["s2_sr"].copy()
    else:
        if dsen2 is not None:
            s2[..., :10] = R.superresolve_large_tile(s2[..., :10], dsen2)
        if cache is not None:
            cache["s2_sr"] = s2.copy()
    clk.lap("dsen2")
    raws, spent = [], [0.0]

    def predict(win):
        k = len(raws)
        if only_windows is not None and k not in only_windows:
            raws.append(None)
            return np.full((size, size), 0.5, np.float32)
        t0 = time.time()
        p = R.predict_subtile(win, net, size)
        spent[0] += time.time() - t0
        raws.append(None if np.ndim(p) == 0 else np.array(p, copy=True))
        return p
    wins, feeds = R.process_subtiles(s2, dates.copy(), interp, s1, dem90, predict, size=size, length=length, return_inputs=True)
    clk.lap("preprocess+post+model")
    if clk.t is not None:
        clk.t["model"] = spent[0]
        clk.t["preprocess+post"] = clk.t.pop("preprocess+post+model") - spent[0]
    u8, f32 = R.mosaic_predictions(wins, size=size, return_float=True)
    clk.lap("mosaic")
    # process_subtiles calls predict in window-grid order, for the windows it predicts (the others are 255 fills)
    order = list(wins.keys())                                        # insertion order = iteration order, keys (folder_y, folder_x)
    predicted = [k for k in order if k in feeds]
    raw = {k: raws[i] for i, k in enumerate(predicted) if raws[i] is not None}
    return {"windows": wins, "order": order, "raw": raw, "feeds": feeds, "u8": u8, "f32": f32, "dates": dates, "interp": interp, "s2": s2}


def single_call_chain(s2_10, s2_20, s1, dem90, mask, dates, net, dsen2, size=158, length=4, sampler="expected",
                      only_windows=None, timings=None, cache=None):
    """What ttc_predict_tile computes for a tile on which none of process_tile's date-dropping rules fire (its status words
    stay zero): to_float32 + convert_to_db, bilinear 20 m -> 10 m, remove_cloud_and_shadows with the GIVEN mask and the
    deterministic expected-multiplicity sampler, process_tile's final clip, then the rest of the chain.
    dem90: the elevation as process_tile returns it (median-filtered, / 90), on the DEM file's own grid.  The 10 m bands, Sentinel-1 and
    the DEM may be a pixel (or an even number of pixels) off the grid of the 20 m stack: adjust_shape reconciles them (job.py:716-721).
    cache: a dict shared by passes over the SAME tile and sampler at different window geometries (size / length): the stages up
    to and including DSen2 do not depend on the geometry and are computed by the first pass only."""
    clk = _Clock(timings)
    if cache is not None and "gapfilled" in cache:
        s2, interp, to_remove, s1db = (np.array(v, copy=True) if isinstance(v, np.ndarray) else list(v) for v in cache["gapfilled"])
    else:
        s2_10f, s2_20f, s1db = R.to_float32(s2_10), R.to_float32(s2_20), R.s1_to_db(s1)
        width, height = s2_20f.shape[1] * 2, s2_20f.shape[2] * 2                   # job.py:716-721 (ttc_predict_tile_shaped)
        s2_10f, s1db = R.adjust_shape(s2_10f, width, height), R.adjust_shape(s1db, width, height)
        dem90 = R.adjust_shape(np.asarray(dem90, dtype=np.float32), width, height)
        clk.lap("codecs")
        s2 = R.upsample_20m(s2_10f, s2_20f)
        clk.lap("bilinear")
        s2, interp, to_remove = G.remove_cloud_and_shadows(s2, np.array(mask, dtype=np.float32, copy=True),
                                                           np.zeros(s2.shape[1:3], bool), sampler)
        s2 = np.clip(s2, 0, 1)
        if cache is not None:
            cache["gapfilled"] = (s2.copy(), np.array(interp, copy=True), list(to_remove), s1db.copy())
    clk.lap("gapfill")
    out = _finish(s2, np.asarray(dates).copy(), interp, s1db, np.asarray(dem90, dtype=np.float32), net, dsen2, size, length,
                  only_windows, clk, cache)
    out["to_remove"] = list(to_remove)
    return out


def checked_chain(raw, mask, net, dsen2, size=158, length=4, sampler="expected"):
    """The job's chain with process_tile's own decisions (dates dropped for missing data / snow / heavy cloud, dates the
    gap-fill flags as fully interpolated), the cloud + shadow mask given: what job.predict_tile_raw_checked must return
    whether or not the single call's speculation held.  raw: dict s2_10 / s2_20 / s1 uint16, dem in METRES, dates."""
    s2, dates, interp, s1db, dem90, cloudshad, snow = P.process_tile_arrays(raw, sampler=sampler, cloudshad=mask)
    out = _finish(s2, dates, interp, s1db, dem90.astype(np.float32), net, dsen2, size, length, None, _Clock(None))
    out["cloudshad"] = cloudshad
    return out
