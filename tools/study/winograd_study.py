"""CPU study (round 4, late): what would Winograd F(4x4, 3x3) cost in ACCURACY on the fp32 engine?

The fp32 conv kernels are power-bound (the launches run at 1.8-2.3 GHz), so fewer matrix instructions is what
moves them: F(2x2, 3x3) (adopted) issues 16 multiply-accumulates per 4 outputs = 4.0 per output, F(4x4, 3x3) 36 per 16 = 2.25.
Its transforms are no longer 0 / +-1 / +-1/2 (B^T holds 4, 5, 2; G 1/4 .. 1/24; A^T up to 8), so the fp32 error grows.  This
runs the torch oracle (oracle/restate_model.py) with the convs of chosen layers replaced by an fp32 Winograd evaluation in the
order a kernel would use (U = G g G^T in double then rounded to fp32 -- conv_pack_wino; V = B^T d B, the cin sum and A^T M A in
fp32) and reports max / rms / p99.9 |dprob| against the fp64 forward, beside the direct fp32 conv.

    python tools/study/winograd_study.py [--win 172] [--length 4] [--n 2] [--scaled] [--smooth]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import restate_model as M  # noqa: E402
from tools.study.precision_study import QNet, LAYERS, stored_scale  # noqa: E402

# F(m x m, 3 x 3) transform matrices (Lavin & Gray 2015): Y = A^T [ (G g G^T) . (B^T d B) ] A
MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                  [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_conv(x, k, m):
    """VALID 3x3 conv of x [N, C, H, W] with k [Cout, C, 3, 3] by F(m x m, 3 x 3), every step in x's dtype except U = G g G^T"""
    Bt, G, At = MATS[m]
    dt = x.dtype
    N, C, H, W = x.shape
    Ho, Wo = H - 2, W - 2
    ty, tx = -(-Ho // m), -(-Wo // m)
    xp = F.pad(x, (0, tx * m + 2 - W, 0, ty * m + 2 - H))                              # zero columns / rows past the image: cropped below
    d = xp.unfold(2, m + 2, m).unfold(3, m + 2, m)                                        # [N, C, ty, tx, a, b]
    Btt, Att = torch.as_tensor(Bt, dtype=dt), torch.as_tensor(At, dtype=dt)
    U = torch.as_tensor(np.einsum("xu,ocuv,yv->xyoc", G, k.double().numpy(), G)).to(dt)  # [a, b, Cout, C], double then rounded
    V = torch.einsum("xa,ncijab,yb->xynijc", Btt, d, Btt)                                 # [a, b, N, ty, tx, C]
    Mx = torch.einsum("xynijc,xyoc->xynijo", V, U)                                        # the 16 / 36 GEMMs over cin
    Y = torch.einsum("pa,abnijo,qb->noipjq", Att, Mx, Att)                                # [N, Cout, ty, m, tx, m]
    return Y.reshape(N, k.shape[0], ty * m, tx * m)[:, :, :Ho, :Wo].contiguous()


class WNet(QNet):
    """TreeCoverNet with the 3x3 convs of chosen layers evaluated by Winograd F(m x m).  wino: {layer: m}"""

    def __init__(self, weights, wino, **kw):
        super().__init__(weights, {}, **kw)
        self.wino = wino

    def _conv(self, layer, a, k, padding=0):
        m = self.wino.get(layer)
        if padding:
            a = F.pad(a, (1, 1, 1, 1))
        return wino_conv(a, k, m) if m else F.conv2d(a, k)

    def _cell(self, d, x, h):
        w, dt = self.w, self.dt
        p = f"gru/{d}/"
        inp = F.pad(torch.cat([x, h], 1), (1, 1, 1, 1), mode="reflect")
        y = self._conv("gates", inp, M._k(w, p + "gates/kernel", dt))
        r, u = torch.chunk(y, 2, dim=1)
        r = torch.sigmoid(M.group_norm(r, M._v(w, p + "gates_r/gamma", dt), M._v(w, p + "gates_r/beta", dt)))
        u = torch.sigmoid(M.group_norm(u, M._v(w, p + "gates_u/gamma", dt), M._v(w, p + "gates_u/beta", dt)))
        inp = F.pad(torch.cat([x, r * h], 1), (1, 1, 1, 1), mode="reflect")
        y = self._conv("cand", inp, M._k(w, p + "candidate/kernel", dt))
        y = y * torch.sigmoid(F.conv2d(y, M._k(w, p + "candidate/kernel_1", dt)))
        y = M.group_norm(y, M._v(w, p + "candidate_y/gamma", dt), M._v(w, p + "candidate_y/beta", dt))
        return u * h + (1 - u) * torch.tanh(y)

    def block(self, name, x, padding):
        w, dt = self.w, self.dt
        k = M._k(w, name + "/kernel", dt)
        if padding == "SAME":
            y = self._conv(name, x, k, padding=1)
            ones = torch.ones(1, 1, x.shape[2], x.shape[3], dtype=dt)
            cnt = F.conv2d(ones, torch.ones(1, 1, 3, 3, dtype=dt), padding=1)
            y = y * (9.0 / cnt)
        else:
            y = self._conv(name, x, k)
        y = y * torch.sigmoid(y)
        y = M.group_norm(y, M._v(w, name + "/gamma", dt), M._v(w, name + "/beta", dt))
        gate = torch.sigmoid(F.conv2d(y, M._k(w, name + "/sse_kernel", dt), M._v(w, name + "/sse_bias", dt)))
        return y * gate


class WDSen2(M.DSen2Lite):
    """DSen2Lite with its 32 -> 32 (and 10 -> 32) convs by Winograd F(m x m); the 6-channel head stays direct"""

    def __init__(self, weights, m, **kw):
        super().__init__(weights, **kw)
        self.m = m

    def _conv(self, x, name):
        if name == "out_conv" or not self.m:
            return super()._conv(x, name)
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        return wino_conv(x, M._k(self.w, name + "/kernel", self.dt), self.m) + M._v(self.w, name + "/bias", self.dt).view(1, -1, 1, 1)


def dsen2_part(seed):
    """DSen2 with the package's real weights on the golden 118-px input and a seeded reflectance-like one: |d reflectance| vs fp64"""
    from ttc import weights as W
    wd = W.load_dsen2()
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "dsen2_graph.npz"))
    rng = np.random.default_rng(seed)
    cases = [("golden b (118 px)", g["b_x"], g["b_bil"]),
             ("seeded 118 px x 4", (rng.random((4, 118, 118, 10)) * 0.6).astype(np.float32), (rng.random((4, 118, 118, 6)) * 0.6).astype(np.float32))]
    for name, x, bil in cases:
        y64 = M.DSen2Lite(wd, dtype=torch.float64)(x.astype(np.float64), bil.astype(np.float64))
        for tag, net in (("direct fp32", M.DSen2Lite(wd)), ("F(2x2) fp32", WDSen2(wd, 2)), ("F(4x4) fp32", WDSen2(wd, 4))):
            d = np.abs(net(x, bil).astype(np.float64) - y64)
            print(f"  DSen2, {name:18s} {tag:12s}: max |d reflectance| {d.max():.2e}  rms {np.sqrt((d ** 2).mean()):.2e}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--win", type=int, default=172)
    ap.add_argument("--length", type=int, default=4)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--scaled", action="store_true", help="round-1 synthetic scale (kernels / sqrt(9 cin))")
    ap.add_argument("--smooth", action="store_true", help="spatially smooth inputs instead of white noise")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--per-layer", action="store_true", help="also: F(4x4) in ONE layer at a time")
    ap.add_argument("--dsen2-only", action="store_true", help="only the DSen2 part (real weights)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    dsen2_part(a.seed)
    if a.dsen2_only:
        return
    w = M.synth_weights(a.seed)
    if not a.scaled:
        w = stored_scale(w)
    rng = np.random.default_rng(a.seed + 1)
    x = rng.uniform(-1, 1, (a.n, a.length + 1, a.win, a.win, 17)).astype(np.float32)
    if a.smooth:
        t = torch.as_tensor(x).permute(0, 1, 4, 2, 3).reshape(-1, 1, a.win, a.win)
        t = F.conv2d(F.pad(t, (4, 4, 4, 4), mode="reflect"), torch.ones(1, 1, 9, 9) / 81) * 4
        x = t.reshape(a.n, a.length + 1, 17, a.win, a.win).permute(0, 1, 3, 4, 2).clamp(-1, 1).numpy().copy()
    # the conv itself first: one gates-shaped layer, error of the raw outputs relative to their rms
    xi = torch.as_tensor(rng.uniform(-1, 1, (1, 49, 64, 64)).astype(np.float32))
    kk = torch.as_tensor(M._k(w, "gru/fw/gates/kernel", torch.float32))
    y64 = F.conv2d(xi.double(), kk.double())
    for name, y in (("direct fp32", F.conv2d(xi, kk)), ("F(2x2) fp32", wino_conv(xi, kk, 2)), ("F(4x4) fp32", wino_conv(xi, kk, 4)),
                    ("F(4x4) fp64", wino_conv(xi.double(), kk.double(), 4))):
        e = (y.double() - y64).abs()
        print(f"  gates-shaped conv, 49 -> 64, {name:12s}: max |err| / rms(y) {e.max() / y64.pow(2).mean().sqrt():.2e}   rms {e.pow(2).mean().sqrt() / y64.pow(2).mean().sqrt():.2e}")
    ref64 = M.TreeCoverNet(w, dtype=torch.float64)(x.astype(np.float64))
    ref = M.TreeCoverNet(w)(x)
    print(f"W={a.win} L={a.length} n={a.n} stored_scale={not a.scaled} smooth={a.smooth}: direct fp32 vs fp64 max |dprob| {np.abs(ref - ref64).max():.2e}")

    def run(tag, wino):
        got = WNet(w, wino)(x)
        d = np.abs(got.astype(np.float64) - ref64)
        print(f"  {tag:52s} max {d.max():.2e}  rms {np.sqrt((d ** 2).mean()):.2e}  p99.9 {np.quantile(d, 0.999):.2e}", flush=True)

    run("all GroupNorm layers F(2x2)   [the adopted kernel]", {l: 2 for l in LAYERS})
    run("all GroupNorm layers F(4x4)", {l: 4 for l in LAYERS})
    run("ConvGRU F(2x2), U-Net blocks F(4x4)", {**{l: 2 for l in LAYERS[:2]}, **{l: 4 for l in LAYERS[2:]}})
    run("ConvGRU F(4x4), U-Net blocks F(2x2)", {**{l: 4 for l in LAYERS[:2]}, **{l: 2 for l in LAYERS[2:]}})
    if a.per_layer:
        for l in LAYERS:
            run(f"only {l} F(4x4), the others F(2x2)", {**{k: 2 for k in LAYERS}, l: 4})


if __name__ == "__main__":
    main()
