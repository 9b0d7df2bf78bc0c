"""Weight handling for the two graphs of the hot path.

Canonical names (what libttc's ttc_load_weights expects; kernels in TF HWIO layout):

  gru/{fw,bw}/gates/kernel            [3,3,49,64]     gru/{fw,bw}/candidate/kernel   [3,3,49,32]
  gru/{fw,bw}/candidate/kernel_1      [1,1,32,1]      (in-cell sSE, no bias)
  gru/{fw,bw}/{gates_r,gates_u,candidate_y}/{gamma,beta}                              [32]
  {conv_median,conv_concat,conv1,conv2,up2,up2_out,up3,out}/kernel                    [3,3,Cin,Cout]
  .../{gamma,beta} [Cout]   .../sse_kernel [1,1,Cout,1]   .../sse_bias [1]
  head/kernel [1,1,64,1]    head/bias [1]

`TF_NAME_MAP` maps these to the variable names of the reference checkpoint
(models-release/master-ckpt-nonfrozen/-0.meta) so a real checkpoint
exported to .npz (name -> array) loads by name through `from_tf_checkpoint_npz`.
The ConvGRU/U-Net weights themselves are NOT in the reference checkout
(.MISSING_LARGE_BLOBS); `synth_weights` provides seeded stand-ins of the right shapes.
The DSen2-lite weights are the public ones, extracted by tools/extract_dsen2.py.
"""
from __future__ import annotations

import os

import numpy as np

GRU_DIRS = ("fw", "bw")
BLOCKS = [("conv_median", 17, 64), ("conv_concat", 128, 64), ("conv1", 64, 128), ("conv2", 128, 256),
          ("up2", 256, 128), ("up2_out", 256, 128), ("up3", 128, 64), ("out", 128, 64)]

_WS = {"conv_median": "conv_median_conv/conv_median/x/ws_conv2d/kernel",
       "conv_concat": "conv_concat_conv/conv_concat/x/ws_conv2d_1/kernel",
       "conv1": "conv1_conv/conv1/ws_conv2d_2/kernel", "conv2": "conv2_conv/conv2/ws_conv2d_3/kernel",
       "up2": "up2_conv/up2/x/ws_conv2d_4/kernel", "up2_out": "up2_out_conv/up2_out/x/ws_conv2d_5/kernel",
       "up3": "up3_conv/up3/x/ws_conv2d_6/kernel", "out": "out_conv/out/ws_conv2d_7/kernel"}


def _tf_name_map():
    m = {}
    for d in GRU_DIRS:
        cell = f"down_16/bidirectional_rnn/{d}/conv_gru_cell/"
        loop = cell          # exports differ in the while-loop scope; see from_tf_checkpoint_npz(fuzzy)
        m[f"gru/{d}/gates/kernel"] = cell + "gates/kernel"
        m[f"gru/{d}/candidate/kernel"] = cell + "candidate/kernel"
        m[f"gru/{d}/candidate/kernel_1"] = cell + "candidate/kernel_1"
        for g, scope in (("gates_r", "gates/gates_r_norm"), ("gates_u", "gates/gates_u_norm"),
                         ("candidate_y", "candidate/candidate_y_norm")):
            suffix = g
            m[f"gru/{d}/{g}/gamma"] = f"{loop}{scope}/gamma_{suffix}"
            m[f"gru/{d}/{g}/beta"] = f"{loop}{scope}/beta_{suffix}"
    for name, _, _ in BLOCKS:
        m[f"{name}/kernel"] = _WS[name]
        m[f"{name}/gamma"] = f"{name}_norm/gamma_{name}"
        m[f"{name}/beta"] = f"{name}_norm/beta_{name}"
        m[f"{name}/sse_kernel"] = f"csse_{name}_conv/kernel"
        m[f"{name}/sse_bias"] = f"csse_{name}_conv/bias"
    m["head/kernel"] = "conv2d_5/kernel"
    m["head/bias"] = "conv2d_5/bias"
    return m


TF_NAME_MAP = _tf_name_map()


def expected_shapes(n_in=17, hidden=32):
    s = {}
    for d in GRU_DIRS:
        p = f"gru/{d}/"
        s[p + "gates/kernel"] = (3, 3, n_in + hidden, 2 * hidden)
        s[p + "candidate/kernel"] = (3, 3, n_in + hidden, hidden)
        s[p + "candidate/kernel_1"] = (1, 1, hidden, 1)
        for g in ("gates_r", "gates_u", "candidate_y"):
            s[p + g + "/gamma"] = (hidden,)
            s[p + g + "/beta"] = (hidden,)
    for name, cin, cout in BLOCKS:
        s[name + "/kernel"] = (3, 3, cin, cout)
        s[name + "/gamma"] = (cout,)
        s[name + "/beta"] = (cout,)
        s[name + "/sse_kernel"] = (1, 1, cout, 1)
        s[name + "/sse_bias"] = (1,)
    s["head/kernel"] = (1, 1, 64, 1)
    s["head/bias"] = (1,)
    return s


def validate(weights: dict):
    exp = expected_shapes()
    missing = [k for k in exp if k not in weights]
    if missing:
        raise ValueError(f"missing weight tensors: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    for k, shp in exp.items():
        if tuple(np.shape(weights[k])) != shp:
            raise ValueError(f"{k}: expected shape {shp}, got {np.shape(weights[k])}")
    return weights


def from_tf_checkpoint_npz(path_or_dict, fuzzy=True):
    """Load a reference checkpoint exported as {tf variable name: array}.  Names are
    matched exactly first, then (fuzzy) by suffix, because the while-loop scopes of
    tf.nn.bidirectional_dynamic_rnn differ between exports."""
    src = dict(np.load(path_or_dict)) if isinstance(path_or_dict, (str, os.PathLike)) else dict(path_or_dict)
    out = {}
    for ours, theirs in TF_NAME_MAP.items():
        if theirs in src:
            out[ours] = src[theirs]
            continue
        if fuzzy:
            tail = "/".join(theirs.split("/")[-2:])
            d = ours.split("/")[1] if ours.startswith("gru/") else None
            cands = [k for k in src if k.endswith(tail) and (d is None or f"/{d}/" in k)]
            if len(cands) == 1:
                out[ours] = src[cands[0]]
                continue
        raise KeyError(f"checkpoint has no variable for {ours} (expected {theirs})")
    return validate({k: np.asarray(v, dtype=np.float32) for k, v in out.items()})


def synth_weights(seed=0, n_in=17, hidden=32, dtype=np.float32, stored_scale=False):
    """Seeded stand-in weights: He-normal conv kernels, block kernels
    weight-standardised like WSConv2D (model.py:384-390) and rescaled to keep activations
    O(1); gamma/beta perturbed around 1/0."""
    rng = np.random.default_rng(seed)
    w = {}

    def he(shape):
        return rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))

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
        # stored_scale: the kernels as a checkpoint stores them (weight-standardised, std 1 per output channel, SURVEY A.1:
        # inference uses them as stored); default: divided by sqrt(fan-in) so that raw conv outputs stay O(1)
        w[name + "/kernel"] = ws(he((3, 3, cin, cout))) / (1.0 if stored_scale else np.sqrt(9.0 * cin))
        w[name + "/gamma"] = 1.0 + 0.1 * rng.standard_normal(cout)
        w[name + "/beta"] = 0.1 * rng.standard_normal(cout)
        w[name + "/sse_kernel"] = rng.standard_normal((1, 1, cout, 1)) * (1.0 / np.sqrt(cout))
        w[name + "/sse_bias"] = 0.1 * rng.standard_normal(1)
    w["head/kernel"] = rng.standard_normal((1, 1, 64, 1)) * (1.0 / 8.0)
    w["head/bias"] = np.array([-np.log(0.68 / 0.32)])
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in w.items()}


def load_dsen2():
    """The public DSen2-lite weights shipped with the package (extracted from the reference's
    models-release/supres-40k-swir/superresolve_graph.pb by tools/extract_dsen2.py)."""
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights", "dsen2.npz")))
