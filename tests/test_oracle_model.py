"""Self-consistency of the torch-CPU model oracle (parity vs TF is unpinned: SURVEY.md F3/F4)."""
import numpy as np
import pytest
import torch

from oracle import restate_model as M
from tests.helpers import synth


@pytest.mark.parametrize("W,L", [(44, 4), (40, 2)])
def test_shapes_and_fp64_agreement(W, L):
    w = M.synth_weights(seed=0)
    x = synth.synth_windows(seed=1, N=1, L=L, W=W)
    y32 = M.TreeCoverNet(w, dtype=torch.float32)(x)
    y64 = M.TreeCoverNet(w, dtype=torch.float64)(x)
    assert y32.shape == (1, W - 14, W - 14, 1)
    assert np.all((y32 > 0) & (y32 < 1))
    assert np.abs(y32 - y64).max() < 2e-5
    assert y32.std() > 1e-3           # not a constant map


def test_flops_formula():
    assert abs(M.model_flops(172, 4) / 1e9 - 41.97) < 0.01
    assert abs(M.model_flops(172, 12) / 1e9 - 82.05) < 0.01
    assert abs(M.model_flops(168, 4) / 1e9 - 39.99) < 0.01


def test_dsen2_real_weights():
    import os
    from tests.helpers import ROOT
    w = dict(np.load(os.path.join(ROOT, "sentinel-tree-cover_amd", "weights", "dsen2.npz")))
    assert sum(v.size for v in w.values()) == 41638
    rng = np.random.default_rng(0)
    x = rng.random((2, 30, 30, 10)).astype(np.float32)
    y = M.DSen2Lite(w)(x, x[..., 4:])
    assert y.shape == (2, 30, 30, 6)
    y64 = M.DSen2Lite(w, dtype=torch.float64)(x, x[..., 4:])
    assert np.abs(y - y64).max() < 1e-5
    assert np.abs(y - x[..., 4:]).max() < 1.0 and np.abs(y - x[..., 4:]).max() > 1e-4
