"""CPU: the C-ABI library loads and exports every symbol include/ttc.h declares; host-side
weight handling.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.helpers import ROOT
import ttc
from ttc import _lib, weights as W


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_exported():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "ttc.h")).read()
    declared = set(re.findall(r"\b(ttc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ttc_status"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # probe entry points live in csrc/ttc_internal.h, not in the public header
    internal = open(os.path.join(ROOT, "sentinel-tree-cover_amd", "csrc", "ttc_internal.h")).read()
    for name in _lib.INTERNAL_EXPORTS:
        assert name not in hdr and re.search(r"\b" + name + r"\s*\(", internal) and hasattr(lib, name), name
    lib.ttc_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ttc_version()


def test_loader_prototypes():
    _ensure_built()
    lib = _lib.load()
    assert lib.ttc_create.argtypes is not None
    # the binding's struct and the library's agree (ttc_config grows at its end: _lib.load() refuses a stale library)
    assert lib.ttc_config_size() == ctypes.sizeof(_lib.TTCConfig) == 13 * 4
    hdr = open(os.path.join(ROOT, "include", "ttc.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("Model / workspace configuration") if "Model / workspace configuration" in hdr else 0):hdr.index("} ttc_config;")]
    fields = re.findall(r"^\s*(?:u?int32_t|float)\s+(\w+);", body, flags=re.M)
    assert fields == [f[0] for f in _lib.TTCConfig._fields_], fields


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        _lib.Context(win_in=44, length=1, max_windows=1)


def test_weights_match_oracle_generator_and_validate():
    from oracle import restate_model as M
    a, b = W.synth_weights(3), M.synth_weights(3)
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    W.validate(a)
    bad = dict(a)
    bad.pop("head/bias")
    with pytest.raises(ValueError):
        W.validate(bad)
    # round trip through the TF variable names
    tf_named = {W.TF_NAME_MAP[k]: v for k, v in a.items()}
    back = W.from_tf_checkpoint_npz(tf_named)
    for k in a:
        np.testing.assert_array_equal(a[k], back[k])
    assert sum(v.size for v in W.load_dsen2().values()) == 41638


def test_multiply_shift_division_of_the_winograd_tile_walk():
    """conv3x3_wino.hip decomposes a tile id with x / d == (x * (2^40 / d + 1)) >> 40 (launch_w refuses ids >= 2^24 and divisors >= 2^12):
    exact on that whole domain -- checked here on the boundaries of every divisor and a random sample"""
    import numpy as np
    rng = np.random.default_rng(0)
    for d in range(1, 4096):
        m = (1 << 40) // d + 1
        xs = np.concatenate([np.arange(0, min(4 * d, 1 << 24)), (1 << 24) - 1 - np.arange(0, 2 * d),
                             (rng.integers(0, 1 << 24, 64) // d) * d, (rng.integers(1, 1 << 24, 64) // d) * d - 1]).astype(np.uint64)
        xs = xs[xs < (1 << 24)]
        q = (xs * np.uint64(m)) >> np.uint64(40)
        assert np.array_equal(q, xs // np.uint64(d)), d
