import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ttc
from ttc import _lib, synth, weights
from oracle.restate_model import TreeCoverNet
for (W, L, N) in [(44, 2, 3), (172, 4, 2)]:
    x = synth.synth_windows(seed=1, N=N, L=L, W=W)
    w = weights.synth_weights(0)
    want = TreeCoverNet(w)(x)[..., 0]
    for prec in (0, 1):
        ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision=prec)
        ctx.load_weights(w)
        got = ctx.forward_windows(torch.from_numpy(x).cuda()).cpu().numpy()
        print(W, L, N, "precision", prec, "max|dp|", np.abs(got - want).max())
