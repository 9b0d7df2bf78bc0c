"""Host-side probe: torch oracle time per window vs thread count (the oracle is the test checker and bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ttc  # noqa
from oracle import restate_model as M
from ttc import weights as Wt, synth
print("cpus", os.cpu_count(), "torch default threads", torch.get_num_threads())
net = M.TreeCoverNet(Wt.synth_weights(0), dtype=torch.float32)
ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
for W, N in ((44, 6), (172, 3)):
    x = synth.synth_windows(seed=1, N=N, L=4, W=W)
    for nt in (128, 64, 32, 16, 8):
        torch.set_num_threads(nt)
        net(x[:1])
        t = time.time()
        for i in range(N):
            net(x[i:i + 1])
        print(W, "threads", nt, f"{(time.time() - t) / N * 1e3:.1f} ms per window", flush=True)
win = np.random.default_rng(0).random((12, 118, 118, 10)).astype(np.float32)
for nt in (128, 32, 16, 8):
    torch.set_num_threads(nt)
    ds(win, win[..., 4:])
    t = time.time(); ds(win, win[..., 4:]); print("dsen2 12 x 118^2 threads", nt, f"{(time.time() - t) * 1e3:.1f} ms", flush=True)
