"""CPU oracle (torch) for the two neural graphs on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* TreeCoverNet -- bi-directional ConvGRU encoder + median-frame conv + 2-level
  U-Net + 1x1 sigmoid head, restated from
    src/train/src/model.py:100-121 (group_norm), :152-205 (gru_block / convGRU),
    :208-290 (ConvGRUCell), :396-444 (partial_conv), :448-538 (conv_swish_gn),
    :45-61 (sse_block), :540-579 (ZoneoutWrapper, inference branch),
    src/train/train-model.py:140-231 (graph assembly).
  PARITY UNPINNED vs the frozen TF graph (TensorFlow and the weights are absent).
* DSen2Lite -- models-release/supres-40k-swir/superresolve_graph.pb as decoded in
  SURVEY.md Appendix A.5 (weights extracted by tools/extract_dsen2.py).

Weights are a flat {name: ndarray} dict in TF layout (HWIO kernels) -- the same
dict the product's weight packer consumes.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

GRU_DIRS = ("fw", "bw")
BLOCKS = [  # name, cin, cout  (F = base filters = 64)
    ("conv_median", 17, 64), ("conv_concat", 128, 64), ("conv1", 64, 128),
    ("conv2", 128, 256), ("up2", 256, 128), ("up2_out", 256, 128),
    ("up3", 128, 64), ("out", 128, 64),
]


def synth_weights(seed=0, n_in=17, hidden=32, dtype=np.float32, stored_scale=False):
    """Seeded synthetic weights with the shapes of models-release/master-ckpt-nonfrozen/-0.meta
    (SURVEY.md A.1).  Conv kernels ~ He-normal then weight-standardised like WSConv2D
    (model.py:384-390); gamma/beta perturbed around 1/0 so GN affine is exercised."""
    rng = np.random.default_rng(seed)
    w = {}

    def he(shape):
        fan_in = shape[0] * shape[1] * shape[2]
        return rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)

    def ws(k):
        k = k - k.mean(axis=(0, 1, 2), keepdims=True)
        return k / (k.std(axis=(0, 1, 2), keepdims=True) + 1e-5)

    for d in GRU_DIRS:
        p = f"gru/{d}/"
        w[p + "gates/kernel"] = he((3, 3, n_in + hidden, 2 * hidden)) * 0.7
        w[p + "candidate/kernel"] = he((3, 3, n_in + hidden, hidden)) * 0.7
        w[p + "candidate/kernel_1"] = rng.standard_normal((1, 1, hidden, 1)) * 0.3
        for g in ("gates_r", "gates_u", "candidate_y"):
            w[p + g + "/gamma"] = 1.0 + 0.1 * rng.standard_normal(hidden)
            w[p + g + "/beta"] = 0.1 * rng.standard_normal(hidden)
    for name, cin, cout in BLOCKS:
        # stored WS kernels are already standardised (std 1 per output channel); scale them
        # down so activations stay O(1) through the stack like a trained net's do after GN
        w[name + "/kernel"] = ws(he((3, 3, cin, cout))) / (1.0 if stored_scale else np.sqrt(9.0 * cin))
        w[name + "/gamma"] = 1.0 + 0.1 * rng.standard_normal(cout)
        w[name + "/beta"] = 0.1 * rng.standard_normal(cout)
        w[name + "/sse_kernel"] = rng.standard_normal((1, 1, cout, 1)) * (1.0 / np.sqrt(cout))
        w[name + "/sse_bias"] = 0.1 * rng.standard_normal(1)
    w["head/kernel"] = rng.standard_normal((1, 1, 64, 1)) * (1.0 / 8.0)
    w["head/bias"] = np.array([-np.log(0.68 / 0.32)])
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in w.items()}


def _k(w, name, dtype):
    """TF HWIO -> torch OIHW."""
    return torch.as_tensor(np.asarray(w[name])).permute(3, 2, 0, 1).to(dtype).contiguous()


def _v(w, name, dtype):
    return torch.as_tensor(np.asarray(w[name])).to(dtype)


def group_norm(x, gamma, beta, G=8, eps=1e-5):
    """model.py:100-121: stats over (C/G, H, W), biased variance."""
    N, C, H, W = x.shape
    g = x.reshape(N, G, C // G, H, W)
    mean = g.mean(dim=(2, 3, 4), keepdim=True)
    var = ((g - mean) ** 2).mean(dim=(2, 3, 4), keepdim=True)
    g = (g - mean) / torch.sqrt(var + eps)
    return g.reshape(N, C, H, W) * gamma.view(1, C, 1, 1) + beta.view(1, C, 1, 1)


class TreeCoverNet:
    def __init__(self, weights, zoneout=0.75, dtype=torch.float32, trace=None, ws_restandardize=False):
        """ws_restandardize: model.py's WSConv2D.call (:392-394) standardises the kernel on EVERY call, also at inference,
        `(k - mean) / (std + 1e-5)` per output channel; checkpoints store the standardised kernel (training assigns it
        back) and the frozen inference graphs use it as stored (export notebook: no-op assign), so the default
        is False.  The two differ by a ~1e-5 per-channel rescale; True reproduces model.py's graph code exactly
        (tests/test_oracle_model.py pins it to 2e-9 against that code run through tools/tf_shim)."""
        self.w, self.z, self.dt = weights, float(zoneout), dtype
        self.ws_re = bool(ws_restandardize)
        self.trace = trace          # optional dict: name -> ndarray of intermediates (NCHW)

    def _t(self, name, x):
        if self.trace is not None:
            self.trace[name] = x.detach().numpy().copy()

    # -- ConvGRU cell, model.py:240-290 -----------------------------------------------
    def _cell(self, d, x, h):
        w, dt = self.w, self.dt
        p = f"gru/{d}/"
        inp = F.pad(torch.cat([x, h], 1), (1, 1, 1, 1), mode="reflect")
        y = F.conv2d(inp, _k(w, p + "gates/kernel", dt))
        self._t(f"yg_{d}", y)
        r, u = torch.chunk(y, 2, dim=1)
        r = torch.sigmoid(group_norm(r, _v(w, p + "gates_r/gamma", dt), _v(w, p + "gates_r/beta", dt)))
        u = torch.sigmoid(group_norm(u, _v(w, p + "gates_u/gamma", dt), _v(w, p + "gates_u/beta", dt)))
        inp = F.pad(torch.cat([x, r * h], 1), (1, 1, 1, 1), mode="reflect")
        y = F.conv2d(inp, _k(w, p + "candidate/kernel", dt))
        y = y * torch.sigmoid(F.conv2d(y, _k(w, p + "candidate/kernel_1", dt)))
        self._t(f"yc_{d}", y)
        self._t(f"u_{d}", u)
        y = group_norm(y, _v(w, p + "candidate_y/gamma", dt), _v(w, p + "candidate_y/beta", dt))
        return u * h + (1 - u) * torch.tanh(y)

    # -- bi-directional driver + zoneout (model.py:556-579 inference branch, :192-194) --
    def gru(self, x):                       # x [B, L, C, H, W]
        B, L, C, H, W = x.shape
        outs = []
        for d in GRU_DIRS:
            h = torch.zeros(B, self.w[f"gru/{d}/candidate/kernel"].shape[-1], H, W, dtype=self.dt)
            order = range(L) if d == "fw" else range(L - 1, -1, -1)
            for t in order:
                hn = self._cell(d, x[:, t], h)
                h = h * self.z + hn * (1 - self.z)      # state carried on = zoneout mix
            outs.append(h)                              # final STATE, not last output
        self._t("gru", torch.cat(outs, 1))
        return torch.cat(outs, 1)

    # -- conv_swish_gn, model.py:448-538 ---------------------------------------------
    def block(self, name, x, padding):
        w, dt = self.w, self.dt
        k = _k(w, name + "/kernel", dt)
        if self.ws_re:                                   # WSConv2D.standardize_weight, model.py:384-390 (OIHW here)
            k = k - k.mean(dim=(1, 2, 3), keepdim=True)
            k = k / (k.std(dim=(1, 2, 3), keepdim=True, unbiased=False) + 1e-5)
        if padding == "SAME":
            y = F.conv2d(x, k, padding=1)
            ones = torch.ones(1, 1, x.shape[2], x.shape[3], dtype=dt)
            cnt = F.conv2d(ones, torch.ones(1, 1, 3, 3, dtype=dt), padding=1)
            y = y * (9.0 / cnt)                          # partial conv ratio, model.py:403-424
        else:
            y = F.conv2d(x, k)
        y = y * torch.sigmoid(y)                         # swish
        self._t("raw_" + name, y)
        y = group_norm(y, _v(w, name + "/gamma", dt), _v(w, name + "/beta", dt))
        gate = torch.sigmoid(F.conv2d(y, _k(w, name + "/sse_kernel", dt), _v(w, name + "/sse_bias", dt)))
        return y * gate                                  # sSE, model.py:45-61

    def forward(self, inp):
        """inp [B, L+1, W, W, 17] (NHWC per frame) -> [B, W-14, W-14, 1]."""
        x = torch.as_tensor(np.asarray(inp)).to(self.dt).permute(0, 1, 4, 2, 3)
        gru = self.gru(x[:, :-1])
        med = self.block("conv_median", x[:, -1], "SAME")
        concat = self.block("conv_concat", torch.cat([gru, med], 1), "SAME")
        conv1 = self.block("conv1", F.max_pool2d(concat, 2), "VALID")
        conv2 = self.block("conv2", F.max_pool2d(conv1, 2), "VALID")
        up2 = self.block("up2", F.interpolate(conv2, scale_factor=2, mode="nearest"), "SAME")
        up2 = self.block("up2_out", torch.cat([up2, conv1[:, :, 2:-2, 2:-2]], 1), "SAME")
        up3 = self.block("up3", F.interpolate(up2, scale_factor=2, mode="nearest"), "SAME")
        up3 = self.block("out", torch.cat([up3, concat[:, :, 6:-6, 6:-6]], 1), "VALID")
        self._t("late", up3)                             # == predict/csse_out_mul/mul:0 (job.py:1808)
        fm = torch.sigmoid(F.conv2d(up3, _k(self.w, "head/kernel", self.dt), _v(self.w, "head/bias", self.dt)))
        return fm.permute(0, 2, 3, 1).contiguous().numpy()

    __call__ = forward

    def features(self, inp):
        """job.py:1808-1809 taps: (probs [B,o,o,1], early = bi-GRU output [B,W,W,64], late = last block output [B,o,o,64]),
        NHWC like the tensors Session.run returns."""
        keep, self.trace = self.trace, {}
        try:
            probs = self.forward(inp)
            early = np.moveaxis(self.trace["gru"], 1, -1).copy()
            late = np.moveaxis(self.trace["late"], 1, -1).copy()
        finally:
            self.trace = keep
        return probs, early, late


def model_flops(W, L):
    """SURVEY.md 8(d) formula."""
    c1 = W // 2 - 2
    c2 = c1 // 2 - 2
    u2, u3 = 2 * c2, 4 * c2
    o = u3 - 2
    return (2 * 9 * 49 * 96 * W * W * 2 * L
            + 2 * 9 * (17 * 64 * W * W + 128 * 64 * W * W + 64 * 128 * c1 * c1 + 128 * 256 * c2 * c2
                       + 2 * 256 * 128 * u2 * u2 + 128 * 64 * u3 * u3 + 128 * 64 * o * o)
            + 2 * 64 * o * o)


class DSen2Lite:
    """SURVEY.md A.5.  weights: in_conv, 01_conv, 02_conv, 11_conv, 12_conv, out_conv
    each {name}/kernel (HWIO) and {name}/bias."""

    def __init__(self, weights, dtype=torch.float32):
        self.w, self.dt = weights, dtype

    def _conv(self, x, name):
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        return F.conv2d(x, _k(self.w, name + "/kernel", self.dt), _v(self.w, name + "/bias", self.dt))

    def forward(self, inp, bilinear):
        """inp [T,H,W,10], bilinear [T,H,W,6] -> [T,H,W,6]."""
        x = torch.as_tensor(np.asarray(inp)).to(self.dt).permute(0, 3, 1, 2)
        b = torch.as_tensor(np.asarray(bilinear)).to(self.dt).permute(0, 3, 1, 2)
        c = float(np.float32(0.1))          # the graph's Const / Const_1 are float32 0x3DCCCCCD (pinned by tests/golden/dsen2_graph.npz)
        x0 = F.relu(self._conv(x, "in_conv"))
        x1 = x0 + c * self._conv(F.relu(self._conv(x0, "01_conv")), "02_conv")
        x2 = x1 + c * self._conv(F.relu(self._conv(x1, "11_conv")), "12_conv")
        out = b + torch.tanh(self._conv(x2, "out_conv"))
        return out.permute(0, 2, 3, 1).contiguous().numpy()

    __call__ = forward
