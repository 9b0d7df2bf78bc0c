"""GPU parity of the EXACT path bench.py times -- raw uint16 arrays -> ONE ttc_predict_tile call -> rasters -- against the chained
CPU oracle (oracle/restate_e2e.py: the pinned stage restatements in the reference's order, job.py:1995-2020), at the bench's
size (618^2, T = 12, bench seeds 1234 / 1235) in all three precision modes, plus the status words that tell the caller when
the single call's speculation does not hold and the checked wrapper that then re-runs the tile through the staged mirror.

The oracle's gap-fill uses sampler = "expected": a restatement of the deterministic expected-multiplicity weighting
(restate_gapfill.expected_weights), so the HIP path is checked against a specification, not against its staged twin.
Tolerances on PRE-rounding window probabilities (BASELINE.json's contract is 1e-3): fp32 / fp16 2e-4, bf16 1e-3.
"""
import json
import os
import random

import numpy as np
import pytest

from tests.helpers import ROOT, synth

pytestmark = pytest.mark.gpu

TILE, T, SIZE = 618, 12, 158
TOL = {"fp32": 2e-4, "fp16": 2e-4, "bf16": 1e-3, "fp32+ds16": 2e-4}          # measured max|dprob|: fp32 2.6e-5 / 4.8e-5, fp16 5.2e-5, bf16 3.3e-4
FEED_TOL = {"fp32": 1e-4, "fp16": 1e-4, "bf16": 5e-4, "fp32+ds16": 1e-4}
# "fp32+ds16" = an fp32 session whose DSen2 convs run on the 16-bit engine (fp16 hi + lo pairs, three products): ttc_config.dsen2_precision


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


def bench_tile(seed, X=TILE, dates_T=T):
    """the raw arrays bench.py builds for tile id seed - 1234 (bench.py: make_tile)"""
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=seed, T=dates_T, H=X, W=X)
    _, _, _, s1, dem = synth.synth_tile(seed=seed, T=2, H=X, W=X)
    return u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), probs, np.asarray(dates), u16(s1), dem


_ORACLE = {}
_STAGES = {}         # (seed, sampler) -> the oracle's geometry-independent stages (gap-fill, DSen2), shared between window geometries


def oracle_for(seed, sampler="expected", size=SIZE, length=4):
    """one whole-tile oracle pass per (seed, sampler, geometry), shared by the precision cases (~1 min of host time each)"""
    key = (seed, sampler if isinstance(sampler, str) else "reference", size, length)
    if key not in _ORACLE:
        import torch
        from oracle import restate_e2e as E, restate_gapfill as G, restate_model as M
        from ttc import weights as Wt
        s2_10, s2_20, mask, dates, s1, dem = bench_tile(seed)
        net = M.TreeCoverNet(Wt.synth_weights(0), dtype=torch.float32)
        ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
        if not isinstance(sampler, str):
            random.seed(11)
        _ORACLE[key] = E.single_call_chain(s2_10, s2_20, s1, dem, mask, dates, net, ds, size=size, length=length,
                                           sampler=sampler if isinstance(sampler, str) else G.reference_sampler,
                                           cache=_STAGES.setdefault(key[:2], {}))
    return _ORACLE[key]


def hip_tile(seed, precision, size=SIZE, length=4):
    import torch
    from ttc import job, weights as Wt
    sess = job.TTCSession(Wt.synth_weights(0), win_in=size + 14, length=length, max_windows=36, precision=precision.split("+")[0],
                          dsen2_precision="fp16" if precision.endswith("+ds16") else None)
    s2_10, s2_20, mask, dates, s1, dem = bench_tile(seed)
    u8, f32, frames, status = sess.ctx.predict_tile_raw(s2_10, s2_20, s1, dem, mask, dates, job.min_all, job.max_all, size,
                                                        want_float=True, want_inputs=True)
    torch.cuda.synchronize()
    out = {"u8": u8.cpu().numpy(), "f32": f32.cpu().numpy(), "frames": frames.cpu().numpy(), "status": status.cpu().numpy(),
           "raw": sess.ctx.debug_fetch("pt_windows_raw", (36, size, size)), "win": sess.ctx.debug_fetch("pt_windows", (36, size, size))}
    sess.close()
    return out


def window_stats(hip_raw, ref):
    """|dprob| over every window pixel both sides predicted (<= 1: not a 255 fill)"""
    d = []
    for i, k in enumerate(ref["order"]):
        if k not in ref["raw"]:
            continue
        a, b = hip_raw[i], ref["raw"][k]
        ok = (a <= 1.0) & (b <= 1.0)
        d.append(np.abs(a.astype(np.float64) - b)[ok])
    d = np.concatenate(d)
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "frac_gt_1e-3": float((d > 1e-3).mean()),
            "rms": float(np.sqrt((d ** 2).mean())), "n": int(d.size)}


# (seed, precision, window output size, steps): 158 / 4 = the geometry the reference's code runs (172-px inputs); 154 / 12 = the
# geometry BASELINE.json's wording names (168-px inputs, 12 steps; job.py:1457-1472 applies no no-image mask at 154, :1274-1283)
@pytest.mark.parametrize("seed,precision,size,length", [(1234, "fp32", SIZE, 4), (1234, "fp16", SIZE, 4), (1234, "bf16", SIZE, 4),
                                                        (1235, "fp32", SIZE, 4), (1234, "fp32", 154, 12), (1234, "fp16", 154, 12),
                                                        (1234, "fp32+ds16", SIZE, 4)])
def test_single_call_tile_vs_chained_oracle(seed, precision, size, length):
    ref = oracle_for(seed, size=size, length=length)
    got = hip_tile(seed, precision, size, length)
    st = got["status"]
    print(f"[parity] e2e seed {seed} {precision} size {size} L {length}: status {st.tolist()}")
    assert st[0] == 0 and st[2] == 0 and st[3] == 0 and st[1] == len(ref["dates"]) == T
    # model inputs: frames [36, L+1, 17, W+2, W+2] planar padded vs the oracle's feeds [L+1, W, W, 17]
    fd = 0.0
    for i, k in enumerate(ref["order"]):
        if k in ref["feeds"]:
            fd = max(fd, float(np.abs(got["frames"][i][:, :, 1:-1, 1:-1].transpose(0, 2, 3, 1) - ref["feeds"][k]).max()))
    ws = window_stats(got["raw"], ref)
    print(f"[parity] e2e seed {seed} {precision}: model inputs max|d| = {fd:.2e}; pre-rounding windows max|dprob| = {ws['max']:.2e}, "
          f"p99.9 = {ws['p999']:.2e}, rms = {ws['rms']:.2e} over {ws['n']} px")
    # normalised units (reflectance / half-range).  Measured (round 3, 158 / L = 4): fp32 3.8e-6 / 7.8e-6, fp16 3.9e-6, bf16 5.4e-5
    # (the bf16 DSen2 pass); bounds = about 10 x that
    assert fd < FEED_TOL[precision], fd
    assert ws["max"] <= TOL[precision], ws
    # what the reference saves per window (3-decimal rounding, 255 fills) and the two rasters
    same_fill = True
    for i, k in enumerate(ref["order"]):
        same_fill &= np.array_equal(got["win"][i] > 1.0, ref["windows"][k] > 1.0)
    assert same_fill
    assert np.array_equal(np.isnan(got["f32"]), np.isnan(ref["f32"]))
    df = np.abs(np.nan_to_num(got["f32"]) - np.nan_to_num(ref["f32"]))
    d8 = np.abs(got["u8"].astype(int) - ref["u8"].astype(int))
    print(f"[parity] e2e seed {seed} {precision}: percent raster max|d| = {df.max():.3f}, uint8 differing {(d8 > 0).mean():.2e}, > 1 count {(d8 > 1).mean():.2e}")
    assert df.max() <= 0.11 + 100 * TOL[precision] and (d8 > 1).mean() < 1e-5 and (d8 > 0).mean() < 3e-2


def test_expected_sampler_vs_seeded_reference_sampler_end_to_end():
    """The number under the 1e-3 contract that is NOT a kernel property: the bench's deterministic expected-multiplicity
    sampler against ONE seeded draw of the reference's stdlib-random sample (the reference itself draws a different one on
    every run), propagated through the whole chain to pre-rounding probabilities.  Measured, printed, written to
    gpurun_out/e2e_dprob.json (committed under profiles/ and quoted by bench.py as `max_dprob_e2e.sampler_effect`)."""
    got = hip_tile(1234, "fp32")
    exp = window_stats(got["raw"], oracle_for(1234))
    ref = window_stats(got["raw"], oracle_for(1234, sampler=None))
    print(f"[parity] HIP fp32 single call vs oracle(expected sampler): {exp}")
    print(f"[parity] HIP fp32 single call vs oracle(reference sampler, random.seed(11)): {ref}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_dprob.json"), "w") as f:
        json.dump({"tile": "bench seed 1234, 618x618, T=12, W=172, L=4, fp32", "vs_oracle_expected_sampler": exp,
                   "vs_oracle_reference_sampler_seed11": ref}, f, indent=1)
    assert exp["max"] <= 2e-4
    assert ref["p999"] < 5e-4               # measured 1.6e-4; one draw of the reference's own run-to-run spread; NOT bounded by 1e-3 at the maximum
    # The yardstick for that row: the reference's chain against ITSELF under three seeds of stdlib random (tools/reference_run_to_run.py,
    # CPU oracle with the replayed sampler; committed as profiles/r05_reference_run_to_run.json).  The HIP tile's distance to one
    # reference draw must not exceed the distance between two reference draws (+ the kernel's own 2e-4).
    with open(os.path.join(ROOT, "profiles", "r05_reference_run_to_run.json")) as f:
        rr = json.load(f)["reference_run_to_run"]
    print(f"[parity] reference vs itself across seeds (committed): max {rr['max']:.2e}, p999 {rr['p999']:.2e}, > 1e-3: {rr['frac_gt_1e-3']:.2e}")
    assert ref["max"] <= rr["max"] + 2e-4 and ref["p999"] <= rr["p999"] + 2e-5 and ref["frac_gt_1e-3"] <= rr["frac_gt_1e-3"] + 1e-6


# ---- status words and the checked wrapper, at a size the oracle finishes in seconds ---------------------------------------
def small_raw(seed, T=6, X=120, Y=112):
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=seed, T=T, H=X, W=Y)
    _, _, _, s1, dem = synth.synth_tile(seed=seed, T=2, H=X, W=Y)
    raw = {"s2_10": u16(s2[..., :4]), "s2_20": u16(s2[:, ::2, ::2, 4:]), "s1": u16(s1), "dem": (dem * 90.0).astype(np.float32),
           "dates": np.asarray(dates)}
    return raw, probs.astype(np.float32)


def force(case, raw, mask):
    T = mask.shape[0]
    if case == "fully_interpolated_date":            # cloud_removal.py:958-959 -> job.py:964-981
        mask[2] = 1.0
    elif case == "unalignable_date":                 # cloud_removal.py:679-680: <= 1000 usable rows on a tile with land
        mask[3] = 1.0
        mask[3, :20, :90] = 0.0                      # ~890 pixels stay below the 0.25 weight after feathering
    elif case == "half_missing_date":                # id_missing_px(., 2), job.py:786
        raw["s2_10"][4, :70] = 0
    elif case == "snowy_dates":                      # > 10 dates with > 25 % snow pixels, job.py:822
        for t in range(11):
            raw["s2_10"][t, :50] = u16(np.float32(0.5))                 # blue / green / red / nir bright
            raw["s2_20"][t, :25, :, 4] = u16(np.float32(0.05))         # band 8 (SWIR) dark -> NDSI high
    return raw, mask


@pytest.mark.parametrize("case,word,bit", [("clean", None, 0), ("fully_interpolated_date", 2, 0), ("unalignable_date", 0, 0),
                                            ("half_missing_date", 3, 1), ("snowy_dates", 3, 2), ("heavy_cloud_date", 3, 4)])
def test_status_words_and_checked_wrapper(case, word, bit):
    """ttc_predict_tile flags the tiles on which process_tile would have dropped dates (status[0] / [2] / [3]);
    job.predict_tile_raw_checked re-runs exactly those through the staged mirror, and its result equals the oracle chain that
    takes the reference's decisions (oracle/restate_e2e.checked_chain) -- for flagged AND clean tiles."""
    import torch
    from oracle import restate_e2e as E, restate_model as M
    from ttc import job, weights as Wt
    Tn = 14 if case == "snowy_dates" else 6
    raw, mask = small_raw(40 + Tn, T=Tn)
    if case == "heavy_cloud_date":                   # feathered mask > 90 % of the tile but clear rows left: job.py:866
        mask[1] = 1.0
        mask[1, :20, :] = 0.0                        # feathered: 90.8 % covered, 1232 rows stay usable
    raw, mask = force(case, raw, mask)
    w = Wt.synth_weights(0)
    sess = job.TTCSession(w, win_in=44, length=4, max_windows=36)
    f32, u8, st, staged = job.predict_tile_raw_checked(raw, mask, sess, size=30, want_status=True)
    print(f"[parity] status words, {case}: {st.tolist()} staged = {staged}")
    if word is None:
        assert not staged and st[0] == 0 and st[2] == 0 and st[3] == 0
    else:
        assert staged and st[word] != 0 and (bit == 0 or (st[3] & bit))
    net = M.TreeCoverNet(w, dtype=torch.float32)
    ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    ref = E.checked_chain({k: (v.copy() if hasattr(v, "copy") else v) for k, v in raw.items()}, mask.copy(), net, ds, size=30, length=4)
    print(f"[parity] status words, {case}: dates kept by the oracle {len(ref['dates'])} of {Tn}")
    assert np.array_equal(np.isnan(f32), np.isnan(ref["f32"]))
    d = np.abs(np.nan_to_num(f32) - np.nan_to_num(ref["f32"]))
    d8 = np.abs(u8.astype(int) - ref["u8"].astype(int))
    print(f"[parity] status words, {case}: percent raster max|d| = {d.max():.3f}, uint8 > 1 count: {(d8 > 1).mean():.2e}")
    assert d.max() <= 0.15 and (d8 > 1).mean() < 1e-4


def test_predict_tiles_pipeline_matches_per_tile_calls():
    """job.predict_tiles (the tile loop of job.py:1869-2091, K tiles in flight on K sessions / HIP streams, status words read
    late, flagged tiles re-run through the staged mirror) returns, in input order, what predict_tile_raw_checked returns for
    each tile on its own -- for clean AND flagged tiles, with the pipeline deeper than the number of sessions."""
    from ttc import job, weights as Wt
    w = Wt.synth_weights(0)
    cases = ["clean", "unalignable_date", "clean", "fully_interpolated_date", "half_missing_date", "clean", "clean"]
    tiles = []
    for k, case in enumerate(cases):
        raw, mask = small_raw(60 + k)
        tiles.append(force(case, raw, mask))
    sessions = [job.TTCSession(w, win_in=44, length=4, max_windows=36) for _ in range(2)]
    got = job.predict_tiles(((dict(r), m.copy()) for r, m in tiles), sessions, size=30, want_status=True)
    assert len(got) == len(tiles)
    solo = job.TTCSession(w, win_in=44, length=4, max_windows=36)
    n_staged = 0
    for k, ((raw, mask), (f32, u8, st, staged)) in enumerate(zip(tiles, got)):
        rf, ru, rst, rstaged = job.predict_tile_raw_checked(dict(raw), mask.copy(), solo, size=30, want_status=True)
        assert staged == rstaged == (cases[k] != "clean") and np.array_equal(st, rst), (k, cases[k], st, rst)
        assert np.array_equal(np.isnan(f32), np.isnan(rf))
        d8 = np.abs(u8.astype(int) - ru.astype(int))
        assert d8.max() <= 1 and (d8 > 0).mean() < 1e-3, (k, cases[k], d8.max(), (d8 > 0).mean())
        n_staged += staged
    print(f"[parity] predict_tiles: {len(tiles)} tiles on 2 sessions, {n_staged} re-run through the staged path, order and rasters match")
    for sx in sessions + [solo]:
        sx.close()


@pytest.mark.parametrize("n10", [617, 619, 620])
def test_tile_loop_reconciles_raw_shapes(n10):
    """a raw folder whose arrays are not all on one grid -- the case adjust_shape (job.py:260-310) exists for -- through the
    tile loop's FAST path.  The 20 m stack is 309 px (the tile is 618), the 10 m bands are 617 / 619 / 620 px, Sentinel-1 is two rows long and
    one column short, the DEM one row long and one column short: ttc_predict_tile_shaped keys the grid on the 20 m stack (:716-717) and
    re-indexes the others inside its decode passes.  Against the chained oracle (restate_e2e.single_call_chain, whose adjust_shape and the
    scale-then-adjust order are pinned to the reference by tests/test_oracle_tile.py): model inputs of all 36 windows, probabilities of
    the four corner windows + one interior window, status words clean."""
    import torch
    from scipy import ndimage as ndi
    from oracle import restate_e2e as E, restate_model as M
    from ttc import job, weights as Wt
    Tn, size, L = 4, SIZE, 4
    s2_10, s2_20, mask, dates, s1, dem90 = bench_tile(1236, dates_T=Tn)
    assert s2_20.shape[1:3] == (309, 309)
    d = n10 - TILE
    raw = synth.misshape_raw({"s2_10": s2_10, "s2_20": s2_20, "s1": s1, "dem": (dem90 * 90.0).astype(np.float32), "dates": dates},
                             d10=(d, 1 if d == 2 else -d), ds1=(2, -1), ddem=(1, -1))
    assert raw["s2_10"].shape[1] == n10 and raw["s1"].shape[1:3] == (620, 617) and raw["dem"].shape == (619, 617)
    sessions = [job.TTCSession(Wt.synth_weights(0), win_in=size + 14, length=L, max_windows=36) for _ in range(2)]
    res = job.predict_tiles([(raw, mask), (raw, mask)], sessions, size=size, want_status=True)
    assert [r[3] for r in res] == [False, False] and all(int(r[2][k]) == 0 for r in res for k in (0, 2, 3))
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][1].shape == (TILE, TILE)
    # the same tile once more through the context call, with the model feed and the pre-rounding windows
    ctx = sessions[0].ctx
    dem_f = ctx.median5(raw["dem"])
    u8, f32, frames, status = ctx.predict_tile_raw(raw["s2_10"], raw["s2_20"], raw["s1"], ctx.divide(dem_f.clone(), 90.0), mask, dates,
                                                   job.min_all, job.max_all, size, want_float=True, want_inputs=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(u8.cpu().numpy(), res[0][1])
    got_raw = ctx.debug_fetch("pt_windows_raw", (36, size, size))
    frames = frames.cpu().numpy()
    net = M.TreeCoverNet(Wt.synth_weights(0), dtype=torch.float32)
    ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    dem90_file = (ndi.median_filter(raw["dem"], size=5) / np.float32(90.0)).astype(np.float32)          # job.py:713, :993 on the file's own grid
    ref = E.single_call_chain(raw["s2_10"], raw["s2_20"], raw["s1"], dem90_file, mask, dates, net, ds, size=size, length=L,
                              only_windows={0, 5, 14, 30, 35})
    fd = 0.0
    for i, k in enumerate(ref["order"]):
        if k in ref["feeds"]:
            fd = max(fd, float(np.abs(frames[i][:, :, 1:-1, 1:-1].transpose(0, 2, 3, 1) - ref["feeds"][k]).max()))
    ws = window_stats(got_raw, ref)
    print(f"[parity] shapes 10 m {raw['s2_10'].shape[1:3]} / S1 {raw['s1'].shape[1:3]} / DEM {raw['dem'].shape} on a 618 tile: "
          f"model inputs max|d| = {fd:.2e}, windows max|dprob| = {ws['max']:.2e} over {ws['n']} px")
    assert fd < FEED_TOL["fp32"], fd
    assert ws["max"] <= TOL["fp32"] and ws["n"] >= 5 * size * size - 10, ws
    for sx in sessions:
        sx.close()


def test_fast_path_refuses_what_adjust_shape_cannot_reconcile():
    """a mask that is not on the tile's grid, date counts that disagree, a 10 m array 3 px off (the reference's adjust_shape leaves the wrong
    length there and process_tile raises): each is an error naming the array, never a silent read with the wrong strides"""
    from ttc import job, weights as Wt
    size = 30
    sess = job.TTCSession(Wt.synth_weights(0), win_in=size + 14, length=4)
    ctx = sess.ctx
    raw = synth.synth_raw_files(91, 4, 60, 64, False)          # 120 x 128: at least one 110-px DSen2 window
    T, X, Y = 4, 120, 128
    mask = np.zeros((T, X, Y), np.float32)
    dem = np.zeros((X, Y), np.float32)
    args = lambda **kw: dict(dict(s2_10=raw["s2_10"], s2_20=raw["s2_20"], s1=raw["s1"], dem=dem, mask=mask, dates=raw["dates"]), **kw)  # noqa: E731

    def call(**kw):
        a = args(**kw)
        return ctx.predict_tile_raw(a["s2_10"], a["s2_20"], a["s1"], a["dem"], a["mask"], a["dates"], job.min_all, job.max_all, size)
    call()                                                                   # the well-formed tile goes through
    with pytest.raises(ValueError, match="mask"):
        call(mask=mask[:, :-1])
    with pytest.raises(ValueError, match="mask"):
        call(mask=mask[:-1])
    with pytest.raises(ValueError, match="dates"):
        call(dates=raw["dates"][:-1])
    with pytest.raises(ValueError, match="s2_10"):
        call(s2_10=raw["s2_10"][:-1])
    with pytest.raises(ValueError, match="s1"):
        call(s1=raw["s1"][:6])
    with pytest.raises(ValueError, match="uint16"):
        call(s2_10=raw["s2_10"].astype(np.float32))
    with pytest.raises(RuntimeError, match="s2_10 is 123 x 128"):
        call(s2_10=np.pad(raw["s2_10"], ((0, 0), (3, 0), (0, 0), (0, 0))))
    with pytest.raises(RuntimeError, match="dem is 120 x 125"):
        call(dem=dem[:, :-3])
    # the tile loop: same errors, and the staged mirror refuses the same inputs
    with pytest.raises(ValueError, match="mask"):
        job.predict_tiles([(raw, mask[:, 1:])], [sess], size=size)
    with pytest.raises(ValueError, match="mask"):
        job.process_tile(dict(raw, clouds=None), sess, cloudshad=mask[:, 1:], sampler="expected")
    with pytest.raises(ValueError, match="10 m bands"):
        job.process_tile(dict(raw, clouds=None, s2_10=np.pad(raw["s2_10"], ((0, 0), (3, 0), (0, 0), (0, 0)))), sess, sampler="expected")
    sess.close()
