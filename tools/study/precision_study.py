"""CPU study: which conv layers tolerate 16-bit operands?  (round 2, informs the per-layer precision map)

Runs the torch oracle (oracle/restate_model.py) with the OPERANDS of chosen conv layers rounded to fp16 / bf16 before
an fp32-accumulated convolution -- the arithmetic of a v_mfma_f32_32x32x16_{f16,bf16} engine -- and reports
max / rms |dprob| against the exact fp32 forward.  Weights: as-stored scale (WS kernels with std 1 per output
channel) unless --scaled.

    python tools/study/precision_study.py [--win 172] [--length 4] [--n 2] [--scaled]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import restate_model as M  # noqa: E402

LAYERS = ["gates", "cand"] + [b[0] for b in M.BLOCKS]


def rnd(t, mode):
    if mode == "f16":
        return t.to(torch.float16).to(t.dtype)
    if mode == "bf16":
        return t.to(torch.bfloat16).to(t.dtype)
    return t


class QNet(M.TreeCoverNet):
    """TreeCoverNet whose conv operands are rounded per layer.  modes: {layer: (x_mode, w_mode)}"""

    def __init__(self, weights, modes, **kw):
        super().__init__(weights, **kw)
        self.modes = modes

    def _q(self, layer, x, k):
        xm, wm = self.modes.get(layer, ("f32", "f32"))
        return rnd(x, xm), rnd(k, wm)

    def _cell(self, d, x, h):
        w, dt = self.w, self.dt
        p = f"gru/{d}/"
        inp = F.pad(torch.cat([x, h], 1), (1, 1, 1, 1), mode="reflect")
        a, k = self._q("gates", inp, M._k(w, p + "gates/kernel", dt))
        y = F.conv2d(a, k)
        r, u = torch.chunk(y, 2, dim=1)
        r = torch.sigmoid(M.group_norm(r, M._v(w, p + "gates_r/gamma", dt), M._v(w, p + "gates_r/beta", dt)))
        u = torch.sigmoid(M.group_norm(u, M._v(w, p + "gates_u/gamma", dt), M._v(w, p + "gates_u/beta", dt)))
        inp = F.pad(torch.cat([x, r * h], 1), (1, 1, 1, 1), mode="reflect")
        a, k = self._q("cand", inp, M._k(w, p + "candidate/kernel", dt))
        y = F.conv2d(a, k)
        y = y * torch.sigmoid(F.conv2d(y, M._k(w, p + "candidate/kernel_1", dt)))
        y = M.group_norm(y, M._v(w, p + "candidate_y/gamma", dt), M._v(w, p + "candidate_y/beta", dt))
        return u * h + (1 - u) * torch.tanh(y)

    def block(self, name, x, padding):
        w, dt = self.w, self.dt
        a, k = self._q(name, x, M._k(w, name + "/kernel", dt))
        if padding == "SAME":
            y = F.conv2d(a, k, padding=1)
            ones = torch.ones(1, 1, x.shape[2], x.shape[3], dtype=dt)
            cnt = F.conv2d(ones, torch.ones(1, 1, 3, 3, dtype=dt), padding=1)
            y = y * (9.0 / cnt)
        else:
            y = F.conv2d(a, k)
        y = y * torch.sigmoid(y)
        y = M.group_norm(y, M._v(w, name + "/gamma", dt), M._v(w, name + "/beta", dt))
        gate = torch.sigmoid(F.conv2d(y, M._k(w, name + "/sse_kernel", dt), M._v(w, name + "/sse_bias", dt)))
        return y * gate


def stored_scale(w):
    """undo synth_weights' 1/sqrt(9 cin) so that the WS kernels have std 1 per output channel, as stored"""
    w = dict(w)
    for name, cin, _ in M.BLOCKS:
        w[name + "/kernel"] = (w[name + "/kernel"] * np.sqrt(9.0 * cin)).astype(np.float32)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--win", type=int, default=172)
    ap.add_argument("--length", type=int, default=4)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--scaled", action="store_true", help="round-1 synthetic scale (kernels / sqrt(9 cin))")
    ap.add_argument("--smooth", action="store_true", help="spatially smooth inputs instead of white noise")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--two-product", action="store_true",
                    help="round 4: the two 2-product forms of the hi/lo engine per layer -- x_hi*(w_hi+w_lo) = 16-bit activations, exact "
                         "weights; (x_hi+x_lo)*w_hi = exact activations, 16-bit weights -- instead of the 1-product table")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    w = M.synth_weights(a.seed)
    if not a.scaled:
        w = stored_scale(w)
    rng = np.random.default_rng(a.seed + 1)
    x = rng.uniform(-1, 1, (a.n, a.length + 1, a.win, a.win, 17)).astype(np.float32)
    if a.smooth:
        t = torch.as_tensor(x).permute(0, 1, 4, 2, 3).reshape(-1, 1, a.win, a.win)
        k = torch.ones(1, 1, 9, 9) / 81
        t = F.conv2d(F.pad(t, (4, 4, 4, 4), mode="reflect"), k) * 4
        x = t.reshape(a.n, a.length + 1, 17, a.win, a.win).permute(0, 1, 3, 4, 2).clamp(-1, 1).numpy().copy()
    ref64 = M.TreeCoverNet(w, dtype=torch.float64)(x.astype(np.float64))
    ref = M.TreeCoverNet(w)(x)
    print(f"W={a.win} L={a.length} n={a.n} stored_scale={not a.scaled} smooth={a.smooth}: fp32 vs fp64 max {np.abs(ref - ref64).max():.2e};"
          f" p range [{ref.min():.3f}, {ref.max():.3f}] mean {ref.mean():.3f}")

    def run(tag, modes):
        got = QNet(w, modes)(x)
        d = np.abs(got.astype(np.float64) - ref64)
        print(f"  {tag:46s} max {d.max():.2e}  rms {np.sqrt((d ** 2).mean()):.2e}  p99.9 {np.quantile(d, 0.999):.2e}", flush=True)
        return d.max()

    if a.two_product:
        for m in ("f16", "bf16"):
            print(f"  -- {m}: 2-product forms (the dropped cross term is the one with the rounded operand's lo part)")
            run(f"all layers {m} x, exact w   [x_hi*w_hi + x_hi*w_lo]", {l: (m, "f32") for l in LAYERS})
            run(f"all layers exact x, {m} w   [x_hi*w_hi + x_lo*w_hi]", {l: ("f32", m) for l in LAYERS})
            for l in LAYERS:
                run(f"only {l}: {m} x, exact w", {l: (m, "f32")})
                run(f"only {l}: exact x, {m} w", {l: ("f32", m)})
        return
    for m in ("f16", "bf16"):
        run(f"all layers {m} x {m}", {l: (m, m) for l in LAYERS})
    run("all layers f16 x exact-w", {l: ("f16", "f32") for l in LAYERS})
    run("all layers exact-x x f16 w", {l: ("f32", "f16") for l in LAYERS})
    print("  -- one layer f16 x f16, others exact")
    for l in LAYERS:
        run(f"only {l}", {l: ("f16", "f16")})
    print("  -- one layer exact, others f16 x f16")
    for l in LAYERS:
        run(f"all but {l}", {k: ("f16", "f16") for k in LAYERS if k != l})
    run("GRU exact, U-Net f16", {k: ("f16", "f16") for k in LAYERS[2:]})
    run("GRU f16, U-Net exact", {k: ("f16", "f16") for k in LAYERS[:2]})


if __name__ == "__main__":
    main()
