"""GPU parity of the 16-bit conv engine (ttc_config.precision = 2 fp16 / 3 bf16, conv3x3_h16.hip) against the fp32 / fp64
torch oracle, through the C ABI.  Weights at AS-STORED scale (weight-standardised kernels with std 1 per output channel,
SURVEY A.1): raw conv outputs are ~sqrt(9 Cin) times larger than with the round-1 stand-ins and swish saturates, the regime a
real checkpoint runs in.

Tolerances on probabilities (BASELINE.json's contract is 1e-3):
  fp16, three split products in every layer ("fp16x3", one_term_layers = 0): operands carry 22 mantissa bits -> fp32-class, 1e-4
  fp16 with the ConvGRU gates conv on plain fp16 operands (one_term_layers = 1, NOT the default: 3e-3 on a real tile):
                                                                             white-noise windows 2e-4 .. 6.5e-4 -> 1e-3
  bf16, three split products (16 mantissa bits):                              2.5e-4
"""
import numpy as np
import pytest

from tests.helpers import synth

pytestmark = pytest.mark.gpu

# max|dprob| bounds; measured (round 4): fp16 3-product <= 4.9e-6, fp16 1-product gates <= 6.4e-4, bf16 <= 1.0e-4
MODES = [("fp16", None, 4e-5), ("fp16", 1, 1e-3), ("bf16", None, 2.5e-4)]


def _setup(W, L, N, seed, precision, one_term, stored=True, dtype64=True):
    import torch
    from oracle import restate_model as M
    from ttc import _lib, weights as Wt
    w = Wt.synth_weights(seed, stored_scale=stored)
    x = synth.synth_windows(seed=seed + 1, N=N, L=L, W=W)
    trace = {}
    w64 = {k: v.astype(np.float64) for k, v in w.items()} if dtype64 else w
    ref = M.TreeCoverNet(w64, dtype=torch.float64 if dtype64 else torch.float32, trace=trace)(x.astype(np.float64) if dtype64 else x)
    ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision=precision, one_term_layers=one_term)
    ctx.load_weights(w)
    return ctx, w, x, ref, trace


def _cmp(name, got, ref, atol):
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    i = np.unravel_index(np.argmax(err), err.shape)
    rms = np.sqrt((ref.astype(np.float64) ** 2).mean())
    print(f"[parity] {name:34s} max|d|={err.max():.3e} rms|d|={np.sqrt((err ** 2).mean()):.2e} at {i}  ref_rms={rms:.3e}")
    return err.max() <= atol, f"{name}: max|d| {err.max():.3e} > {atol} at {i} (got {got[i]}, ref {ref[i]})"


@pytest.mark.parametrize("precision,one_term,tol", MODES)
@pytest.mark.parametrize("W,L,N", [(44, 4, 2), (60, 12, 1), (172, 4, 2), (168, 12, 1)])
def test_forward_16bit_matches_oracle(W, L, N, precision, one_term, tol):
    ctx, w, x, ref, tr = _setup(W, L, N, W + L, precision, one_term)
    out = ctx.forward_windows(x).cpu().numpy()
    assert out.shape == (N, W - 14, W - 14) and np.isfinite(out).all()
    ok, m = _cmp(f"prob W{W} L{L} {precision}/{one_term}", out, ref[..., 0], tol)
    assert ok, m
    np.testing.assert_array_equal(out, ctx.forward_windows(x).cpu().numpy())        # deterministic


@pytest.mark.parametrize("W,L,N", [(44, 4, 2), (172, 4, 2), (168, 12, 1)])
def test_two_term_convgru_layers(W, L, N):
    """ttc_config.two_term_layers = 3: the ConvGRU gates and candidate convs multiply x_hi * (w_hi + w_lo) -- 16-bit activations, exact
    weights (conv3x3_h16<TERMS = 2>).  An accuracy OPTION inside the 1e-3 contract (CPU study: 1.4e-4 / 5.0e-4 per layer); it must sit
    between the three-product default and the one-product form, and the other layers must still run three products."""
    import torch
    from oracle import restate_model as M
    from ttc import _lib, weights as Wt
    w = Wt.synth_weights(W + L, stored_scale=True)
    x = synth.synth_windows(seed=W + L + 1, N=N, L=L, W=W)
    ref = M.TreeCoverNet({k: v.astype(np.float64) for k, v in w.items()}, dtype=torch.float64)(x.astype(np.float64))[..., 0]
    errs = {}
    for name, kw in (("three", {}), ("two", {"two_term_layers": 3}), ("one", {"one_term_layers": 3})):
        ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision="fp16", **kw)
        ctx.load_weights(w)
        out = ctx.forward_windows(x).cpu().numpy()
        if name == "two":
            np.testing.assert_array_equal(out, ctx.forward_windows(x).cpu().numpy())    # deterministic
        errs[name] = float(np.abs(out.astype(np.float64) - ref).max())
        ctx.close()
    print(f"[parity] W{W} L{L} fp16 ConvGRU convs: three products {errs['three']:.2e}, two {errs['two']:.2e}, one {errs['one']:.2e}")
    assert errs["three"] <= 4e-5 and errs["two"] <= 1e-3
    assert errs["three"] < errs["two"] <= errs["one"] * 1.5 + 1e-5


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_single_step_intermediates_16bit(precision):
    """L = 1, three products everywhere: every raw conv output of the first ConvGRU step and of the U-Net against the
    float64 oracle, relative to the tensor's own scale (as-stored kernels make raw outputs O(10..100))."""
    W, L, N = 44, 1, 3
    ctx, w, x, ref, tr = _setup(W, L, N, 0, precision, None)
    ctx.keep_intermediates(True)
    out = ctx.forward_windows(x).cpu().numpy()
    rel = 5e-5 if precision == "fp16" else 8e-4      # of the tensor rms; bf16 pairs carry 16 bits: measured 5.5e-4 on the deepest
                                                     # raw outputs (max over elements up to 5x the rms), probabilities 5e-5
    fails = []

    def chk(name, got, want):
        scale = max(1.0, float(np.sqrt((want.astype(np.float64) ** 2).mean())))
        ok, m = _cmp(name, got, want, rel * scale)
        ok or fails.append(m)

    def raw(name, n, C, H, Wd):
        """raw conv outputs of the 16-bit engine: exact fp32, channel-blocked as a plane of top and a plane of bottom 16-bit
        halves ([n][C/8][H * (Wd + 2)][8] each, input pitch Wd + 2; csrc/h16_common.h Raw16) -> planar [n, C, H, Wd]"""
        P = H * (Wd + 2)
        halves = ctx.debug_fetch(name)[:n * C * P].view(np.uint16).reshape(2, n, C // 8, P, 8).astype(np.uint32)
        val = ((halves[0] << 16) | halves[1]).view(np.float32)                     # [n, C/8, P, 8]
        return val.transpose(0, 1, 3, 2).reshape(n, C, H, Wd + 2)[..., :Wd]

    yg = raw("yg", 2 * N, 64, W, W)
    yc = raw("yc", 2 * N, 32, W, W)
    u = ctx.debug_fetch("u", (2 * N, 32, W, W))
    for d, name in enumerate(("fw", "bw")):
        chk("yg_" + name, yg[d * N:(d + 1) * N], tr["yg_" + name])
        chk("u_" + name, u[d * N:(d + 1) * N], tr["u_" + name])
        chk("yc_" + name, yc[d * N:(d + 1) * N], tr["yc_" + name])
    c1 = W // 2 - 2; c2 = c1 // 2 - 2; u2 = 2 * c2; u3 = 2 * u2; o = u3 - 2
    for buf, name, C, H in [("y_med", "conv_median", 64, W), ("y_cat", "conv_concat", 64, W), ("y_c1", "conv1", 128, c1),
                            ("y_c2", "conv2", 256, c2), ("y_u2", "up2", 128, u2), ("y_u2o", "up2_out", 128, u2),
                            ("y_u3", "up3", 64, u3), ("y_out", "out", 64, o)]:
        chk(buf, raw(buf, N, C, H, H), tr["raw_" + name])
    ok, m = _cmp("prob", out, ref[..., 0], 2.5e-4 if precision == "bf16" else 4e-5); ok or fails.append(m)      # measured 4.1e-5 / 3.8e-6
    assert not fails, "\n".join(fails)


@pytest.mark.parametrize("precision,one_term,tol", MODES)
def test_feature_taps_16bit(precision, one_term, tol):
    import torch
    from oracle import restate_model as M
    from ttc import job, weights as Wt
    W, L, N = 44, 2, 2
    w = Wt.synth_weights(3, stored_scale=True)
    x = synth.synth_windows(seed=4, N=N, L=L, W=W)
    probs, early, late = M.TreeCoverNet(w, dtype=torch.float32).features(x)
    sess = job.TTCSession(w, win_in=W, length=L, max_windows=N, dsen2_weights=None, precision=precision, one_term_layers=one_term)
    gp, ge, gl = sess.ctx.forward_taps(x)
    fails = []
    for name, got, ref, t in [("probs", gp.cpu().numpy(), probs[..., 0], tol), ("early", ge.cpu().numpy(), early, 10 * tol),
                              ("late", gl.cpu().numpy(), late, 30 * tol)]:            # late: values up to ~10
        ok, m = _cmp(f"{name} {precision}/{one_term}", got, ref, t); ok or fails.append(m)
    assert not fails, "\n".join(fails)


@pytest.mark.parametrize("precision,tol", [("fp16", 2e-5), ("bf16", 1e-4)])
def test_dsen2_16bit(precision, tol):
    """DSen2-lite (real weights) on the 16-bit engine, three products per layer: window forward, ragged / tiny windows
    (clamped DMA tails, rim kernel for planes too small for the fused reflect rim) and the whole-tile driver."""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job, weights as Wt
    sess = job.TTCSession(None, win_in=44, length=2, max_windows=1, precision=precision)
    net = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    rng = np.random.default_rng(11)
    fails = []
    for n, H, W in [(3, 118, 118), (2, 11, 13), (1, 3, 3), (3, 17, 40), (1, 120, 7)]:
        x = rng.random((n, H, W, 10)).astype(np.float32)
        got = sess.ctx.dsen2_forward(x, x[..., 4:]).cpu().numpy()
        ok, m = _cmp(f"DSen2 {n}x{H}x{W} ({precision})", got, net(x, x[..., 4:]), tol); ok or fails.append(m)
    arr = (rng.random((2, 618, 618, 10)) * 0.6).astype(np.float32)
    ref = O.superresolve_large_tile(arr.copy(), net)
    d = torch.from_numpy(arr.copy()).cuda()
    sess.ctx.superresolve_tile(d, quirks=True)
    out = d.cpu().numpy()
    ok, m = _cmp(f"superresolve tile ({precision})", out, ref, 2.5 * tol); ok or fails.append(m)
    np.testing.assert_array_equal(out[..., :4], arr[..., :4])
    assert not fails, "\n".join(fails)


# 158 / L = 4 in all three precisions is covered from the raw uint16 tile by tests/test_gpu_e2e.py
@pytest.mark.parametrize("precision,size,length", [("fp16", 158, 4), ("fp16", 154, 12)])
def test_tile_16bit_vs_oracle(precision, size, length):
    """Whole 618^2 tile (36 windows of size + 14, L steps) on the 16-bit engine against the fp32 oracle: window probabilities
    BEFORE the reference's 3-decimal rounding within the 1e-3 contract, identical no-data, uint8 raster within one count.
    size = 154 is BASELINE.json's 168-pixel window, where the reference applies no no-image mask (job.py:1457-1472)."""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job, weights as Wt
    from tests.helpers import golden, e2e_inputs
    w = Wt.synth_weights(0, stored_scale=True)
    sess = job.TTCSession(w, win_in=size + 14, length=length, max_windows=36, precision=precision)
    g = golden("e2e_cloudy.npz")
    s2, dates, interp, s1, dem = e2e_inputs(g)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    raw_ref = []

    def model(win):
        p = O.predict_subtile(win, net, size)
        raw_ref.append(np.array(p, copy=True))       # process_subtiles writes the no-image mask into its copy
        return p
    ref_w, feeds = O.process_subtiles(s2.copy(), dates.copy(), interp.copy(), s1.copy(), dem.copy(), model, size=size, length=length,
                                      return_inputs=True)
    ref_u8, ref_f = O.mosaic_predictions(ref_w, size=size, return_float=True)
    wins, raw = job.process_subtiles(0, 0, s2, dates, interp, s1, dem, sess, size=size, return_raw=True)
    assert len(raw_ref) == len(feeds) and len(ref_w) == 36
    worst = 0.0
    for k, r in zip(feeds.keys(), raw_ref):          # windows the oracle fed to the model, in call order
        worst = max(worst, float(np.abs(raw[k].astype(np.float64) - r).max()))
    print(f"[parity] tile {precision} size {size} L {length}: max |dprob| before rounding = {worst:.3e}")
    # measured on MI355X: fp32 5.0e-5, fp16 4.9e-5 (L = 4) / 3.6e-5 (168-pixel windows, L = 12), bf16 3.7e-4
    assert worst <= (1e-3 if precision == "bf16" else 2e-4)
    u8, f32 = job.load_mosaic_predictions(wins, sess=sess, size=size, return_float=True)
    assert np.array_equal(np.isnan(f32), np.isnan(ref_f))
    d = np.abs(u8.astype(int) - ref_u8.astype(int))
    assert (d > 1).mean() < 1e-4 and (d > 0).mean() < 3e-2


def test_calibrate_precision_keeps_the_budget_and_is_reproducible():
    """ttc_calibrate_precision: on smooth windows (a real tile's texture, not white noise) the chosen map's probabilities stay within
    the budget of the in-library fp32 engine's, the map is what a context CREATED with those masks computes (bit-identical), a zero budget
    returns three products everywhere with within_budget = False, and a generous budget buys at least one cheaper layer."""
    import torch
    from ttc import _lib, job, weights as Wt
    W, L, N = 60, 4, 4
    w = Wt.synth_weights(5)
    x = synth.synth_windows(seed=9, N=N, L=L, W=W)
    ref = _lib.Context(win_in=W, length=L, max_windows=N, precision="fp32")
    ref.load_weights(w)
    p32 = ref.forward_windows(x).cpu().numpy()
    c16 = _lib.Context(win_in=W, length=L, max_windows=N, precision="fp16")
    c16.load_weights(w)
    tight = c16.calibrate_precision(ref, x, budget=0.0)
    assert tight["one_term_layers"] == 0 and tight["two_term_layers"] == 0 and not tight["within_budget"] and tight["matrix_work_ratio"] == 1.0
    assert 0 < tight["dprob_all_three"] < 2e-4
    rep = c16.calibrate_precision(ref, x, budget=2e-2)
    print("[calibrate] budget 2e-2:", {k: rep[k] for k in ("products_per_layer", "max_dprob", "dprob_all_three", "matrix_work_ratio", "trials")})
    assert rep["within_budget"] and rep["max_dprob"] <= 2e-2 and rep["matrix_work_ratio"] < 1.0
    assert (rep["one_term_layers"] | rep["two_term_layers"]) != 0 and (rep["one_term_layers"] & rep["two_term_layers"]) == 0
    got = c16.forward_windows(x).cpu().numpy()                                   # the map is left applied
    assert abs(float(np.abs(got - p32).max()) - rep["max_dprob"]) < 1e-7
    fresh = _lib.Context(win_in=W, length=L, max_windows=N, precision="fp16", one_term_layers=rep["one_term_layers"],
                         two_term_layers=rep["two_term_layers"])
    fresh.load_weights(w)
    np.testing.assert_array_equal(fresh.forward_windows(x).cpu().numpy(), got)
    # a budget between the all-three floor and the cheapest single-layer error changes nothing
    floor, cheapest = rep["dprob_all_three"], min(min(rep["layer_alone_one_product"].values()), min(rep["layer_alone_two_products"].values()))
    if cheapest > 1.5 * floor:
        mid = c16.calibrate_precision(ref, x, budget=(floor + cheapest) / 2)
        assert mid["one_term_layers"] == 0 and mid["two_term_layers"] == 0 and mid["within_budget"]
    # the session-level spelling
    sa = job.TTCSession(w, win_in=W, length=L, max_windows=N, precision="auto", budget=2e-2, calibration_windows=x, dsen2_weights=None)
    assert sa.calibration["one_term_layers"] == rep["one_term_layers"] and sa.calibration["two_term_layers"] == rep["two_term_layers"]
    np.testing.assert_array_equal(sa.ctx.forward_windows(x).cpu().numpy(), got)
    # refusals: an fp32 context cannot be calibrated, the reference must be fp32, geometry must match
    with pytest.raises(RuntimeError, match="16-bit"):
        ref.calibrate_precision(ref, x)
    with pytest.raises(RuntimeError, match="fp32"):
        c16.calibrate_precision(fresh, x)
    other = _lib.Context(win_in=44, length=L, max_windows=N, precision="fp32")
    other.load_weights(w)
    with pytest.raises(RuntimeError, match="geometry"):
        c16.calibrate_precision(other, x)
    for cx in (ref, c16, fresh, other):
        cx.close()
    sa.close()
