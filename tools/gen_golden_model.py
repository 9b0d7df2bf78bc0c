"""Pins oracle/restate_model.py against the REFERENCE'S OWN model code.

TensorFlow cannot be installed in the build container, but the reference's model definition is plain Python that only
CONSTRUCTS a graph: src/train/src/model.py (group_norm / weighted_group_norm :100-150, gru_block / convGRU :152-205,
ConvGRUCell :208-290, WSConv2D / partial_conv :380-444, conv_swish_gn :448-538, ZoneoutWrapper :540-579, DropBlock layers)
and the assembly in src/train/train-model.py:117-231.  tools/tf_shim/ provides a torch-backed eager `tensorflow` / `keras`
stand-in for the symbols that code touches, so this script IMPORTS model.py from /root/reference and EXECUTES lines 117-231
of train-model.py (read at run time, nothing copied) with seeded variables injected by TF variable name, at the production
window sizes, and writes tests/golden/model_tfshim.npz: {seed, config, fm, gru samples, csse_out samples, variable names}.

Build container only (needs /root/reference):   python tools/gen_golden_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "tools", "tf_shim"), os.path.join(REF, "src", "train"), ROOT]

import tensorflow as tf  # noqa: E402  (the shim)
from tests.helpers import synth  # noqa: E402
import importlib  # noqa: E402

W_ = importlib.import_module("sentinel-tree-cover_amd.weights")

ASSEMBLY = (117, 231)          # train-model.py lines: model definition up to and including the sigmoid head


def tf_to_ours(name):
    """TF variable name (as the reference's code creates it) -> canonical name of sentinel-tree-cover_amd/weights.py"""
    for ours, theirs in W_.TF_NAME_MAP.items():
        if name == theirs:
            return ours
    # names that differ between exports (weights.from_tf_checkpoint_npz matches them by suffix as well):
    # the while-loop scope of bidirectional_dynamic_rnn does not exist in eager execution, the head is the first unnamed Conv2D
    for d in ("fw", "bw"):
        pre = f"down_16/bidirectional_rnn/{d}/conv_gru_cell/"
        if name.startswith(pre):
            tail = name[len(pre):]
            m = {"gates/gates_r_norm/gamma_gates_r": "gates_r/gamma", "gates/gates_r_norm/beta_gates_r": "gates_r/beta",
                 "gates/gates_u_norm/gamma_gates_u": "gates_u/gamma", "gates/gates_u_norm/beta_gates_u": "gates_u/beta",
                 "candidate/candidate_y_norm/gamma_candidate_y": "candidate_y/gamma",
                 "candidate/candidate_y_norm/beta_candidate_y": "candidate_y/beta"}
            if tail in m:
                return f"gru/{d}/{m[tail]}"
    if name == "conv2d/kernel":
        return "head/kernel"
    if name == "conv2d/bias":
        return "head/bias"
    return None


def run_reference(win, length, weights, x):
    """execute the reference's graph code on x [B, L+1, W, W, 17]; returns (fm, gru, csse_out, used variable names)"""
    tf.reset()
    import src.model as ref_model            # /root/reference/src/train/src/model.py
    used = {}

    def provider(full, shape):
        ours = tf_to_ours(full)
        if ours is None:
            raise KeyError(f"reference graph asks for variable {full} {shape}: no canonical weight maps to it")
        v = np.asarray(weights[ours], dtype=np.float64).reshape(shape)
        used[full] = ours
        return v
    tf.VARIABLE_PROVIDER = provider
    # placeholders in creation order (train-model.py:125-138): inp, length, labels, keep_rate, is_training, alpha, ...
    tf.FEEDS[:] = [x.astype(np.float64), np.full((x.shape[0],), length), None, 1.0, False]
    src_lines = open(os.path.join(REF, "src", "train", "train-model.py")).read().split("\n")
    body = "\n".join(l[4:] if l.startswith("    ") else l for l in src_lines[ASSEMBLY[0] - 1:ASSEMBLY[1]])
    import types
    srcns = types.SimpleNamespace(model=ref_model)
    ns = {"tf": tf, "np": np, "src": srcns, "MaxPool2D": tf.MaxPool2D, "Cropping2D": tf.Cropping2D, "Conv2D": tf.Conv2D,
          "args": {"zoneout": 0.75, "base_filters": 64, "in_size": win, "out_size": win - 14, "length": length, "n_bands": 17}}
    exec(compile(body, "train-model.py[117:231]", "exec"), ns)
    return ns["fm"].numpy(), ns["gru"].numpy(), ns["up3"].numpy(), used


def main():
    import io
    import contextlib
    out = {}
    seed = 0
    w = W_.synth_weights(seed, stored_scale=True)
    cfgs = [(44, 4, 2), (172, 4, 1), (168, 12, 1)]
    names = None
    for win, L, N in cfgs:
        x = synth.synth_windows(seed=100 + win + L, N=N, L=L, W=win)
        with contextlib.redirect_stdout(io.StringIO()):            # the reference's code prints its layer table
            fm, gru, late, used = run_reference(win, L, w, x)
        tag = f"w{win}_l{L}"
        out[tag + "_cfg"] = np.array([win, L, N, 100 + win + L])
        out[tag + "_fm"] = fm[..., 0].astype(np.float64)
        st = 1 if win < 100 else 4                                 # strided samples of the two 64-channel feature maps
        out[tag + "_gru"] = gru[:, ::st, ::st, :].astype(np.float32)
        out[tag + "_late"] = late[:, ::st, ::st, :].astype(np.float32)
        out[tag + "_stride"] = np.array(st)
        names = used
        print(tag, "fm", fm.shape, float(fm.min()), float(fm.max()), "gru", gru.shape, "late", late.shape, "variables", len(used))
    missing = sorted(set(W_.expected_shapes()) - set(names.values()))
    assert not missing, f"canonical weights the reference graph never asked for: {missing}"
    out["weights_seed"] = np.array(seed)
    out["tf_variable_names"] = np.array(sorted(names.keys()))
    out["canonical_names"] = np.array([names[k] for k in sorted(names.keys())])
    path = os.path.join(ROOT, "tests", "golden", "model_tfshim.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
