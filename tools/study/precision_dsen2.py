"""CPU study: DSen2-lite with fp16 / bf16 conv operands (fp32 accumulate) vs the exact fp32 forward (real weights)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import restate_model as M  # noqa: E402

W = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "sentinel-tree-cover_amd", "weights", "dsen2.npz")))


def rnd(t, mode):
    return t.to({"f16": torch.float16, "bf16": torch.bfloat16}[mode]).to(t.dtype) if mode != "f32" else t


class Q(M.DSen2Lite):
    def __init__(self, w, modes):
        super().__init__(w)
        self.modes = modes

    def _conv(self, x, name):
        xm, wm = self.modes.get(name, ("f32", "f32"))
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        return F.conv2d(rnd(x, xm), rnd(M._k(self.w, name + "/kernel", self.dt), wm), M._v(self.w, name + "/bias", self.dt))


names = ["in_conv", "01_conv", "02_conv", "11_conv", "12_conv", "out_conv"]
rng = np.random.default_rng(0)
T, H = 6, 118
base = rng.uniform(0.02, 0.45, (T, 1, 10, H // 4 + 2, H // 4 + 2)).astype(np.float32)
x = F.interpolate(torch.as_tensor(base[:, 0]), size=(H, H), mode="bilinear").permute(0, 2, 3, 1).numpy()
x = (x + rng.normal(0, 0.01, x.shape)).astype(np.float32).clip(0, 1)
b = x[..., 4:].copy()
ref = M.DSen2Lite(W)(x, b)
ref64 = M.DSen2Lite(W, dtype=torch.float64)(x.astype(np.float64), b.astype(np.float64))
print("fp32 vs fp64", np.abs(ref - ref64).max(), " |out - bilinear| max", np.abs(ref - b).max())
for k in names:
    print(k, "kernel absmax", np.abs(W[k + "/kernel"]).max(), "min nonzero", np.abs(W[k + "/kernel"])[W[k + "/kernel"] != 0].min())
for m in ("f16", "bf16"):
    d = np.abs(Q(W, {k: (m, m) for k in names})(x, b) - ref64)
    print(f"all {m}: max {d.max():.2e} rms {np.sqrt((d**2).mean()):.2e}")
for k in names:
    d = np.abs(Q(W, {k: ("f16", "f16")})(x, b) - ref64)
    print(f"only {k} f16: max {d.max():.2e} rms {np.sqrt((d**2).mean()):.2e}")
