"""GPU probe: per-workgroup phase timeline of one 16-bit conv launch (s_memtime stamps of thread 0 at the tile boundaries).
python tools/probes/h16_trace.py [fp16|bf16] [epi: 0 gates | 1 candidate | 2 swish blocks] [desync]"""
import os
os.environ["TTC_ENABLE_PROBE_KNOBS"] = "1"     # ttc_debug_knob is refused otherwise (process-wide probe state)
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa: F401
from ttc import _lib, synth, weights

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16"
EPI = int(sys.argv[2]) if len(sys.argv) > 2 else 0
DESYNC = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, L, N = 172, 4, 36
ctx = _lib.Context(win_in=W, length=L, max_windows=N, precision=PREC)
ctx.load_weights(weights.synth_weights(0))
x = torch.from_numpy(synth.synth_windows(seed=1, N=N, L=L, W=W)).cuda()
lib = _lib.load()
for _ in range(2):
    ctx.forward_windows(x)
torch.cuda.synchronize()
G = 512
buf = torch.zeros((G, 64), dtype=torch.int64, device="cuda")
ptr = buf.data_ptr()
lo, hi = ptr & 0xffffffff, ptr >> 32
lib.ttc_debug_knob(1, DESYNC)
lib.ttc_debug_knob(4, EPI)
lib.ttc_debug_knob(2, lo - (1 << 32) if lo >= (1 << 31) else lo)
lib.ttc_debug_knob(3, hi)
ctx.timing(2); ctx.kernel_ms(None)
ctx.forward_windows(x)
torch.cuda.synchronize()
ms, n = ctx.kernel_ms("conv_gates")
lib.ttc_debug_knob(2, -1); lib.ttc_debug_knob(3, -1)
t = buf.cpu().numpy().astype(np.uint64)
st = t[:, :48].reshape(G, 12, 4).astype(np.float64)
ok = st[:, :, 3] > 0
ids = t[:, 63]
hw, xcc = (ids >> np.uint64(32)).astype(np.int64), (ids & np.uint64(0xffffffff)).astype(np.int64)
cu = (xcc & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
if not ok.any():
    sys.exit(f"no workgroup was traced: the traced instantiations exist for the fp16 GroupNorm layers only (precision {PREC}, epilogue {EPI})")
print(f"conv_gates event time {ms:.3f} ms x {n}; traced workgroups {int(ok[:, 0].sum())}, tiles stamped {int(ok.sum())}")
# shader clock under this launch: s_memtime ticks per 100 MHz reference tick over each workgroup's whole walk
rt = (t[:, 61].astype(np.float64) - t[:, 60].astype(np.float64))
mt = (t[:, 59].astype(np.float64) - t[:, 62].astype(np.float64))
good = (rt > 0) & (mt > 0)
if good.any():
    ghz = mt[good] / rt[good] * 0.1
    print(f"  shader clock during the walk (s_memtime / s_memrealtime): mean {ghz.mean():.3f} GHz, p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}; "
          f"walk length mean {rt[good].mean() / 100:.1f} us of the launch's {ms * 1e3:.0f} us")
nt = int(ok.sum(axis=1).min())
nt = max(2, min(8, nt))
full = ok[:, :nt].all(axis=1)
s = st[full]
tile = s[:, 1:nt, 0] - s[:, 0:nt - 1, 0]                    # start -> next start
body = s[:, :nt, 1] - s[:, :nt, 0]                      # all chunks but the last
lastc = s[:, :nt, 2] - s[:, :nt, 1]
epi = s[:, :nt, 3] - s[:, :nt, 2]
gap = s[:, 1:nt, 0] - s[:, 0:nt - 1, 3]
for name, a in (("tile period", tile), ("chunks 0..n-2", body), ("last chunk", lastc), ("epilogue", epi), ("epilogue end -> next tile start", gap)):
    print(f"  {name:34s} mean {a.mean():9.0f}  p10 {np.percentile(a, 10):9.0f}  p50 {np.percentile(a, 50):9.0f}  p90 {np.percentile(a, 90):9.0f} ticks")
# phase offset between the two workgroups of a CU, relative to the tile period
offs = []
for c in np.unique(cu[full]):
    m = np.flatnonzero((cu == c) & full)
    if len(m) == 2:
        d = abs(st[m[0], 1, 0] - st[m[1], 1, 0])
        per = np.mean(tile)
        offs.append((d % per) / per)
if offs:
    print(f"  co-resident pairs found: {len(offs)}; start offset of tile 3 as a fraction of the period: mean {np.mean(offs):.2f}, p10 {np.percentile(offs, 10):.2f}, p90 {np.percentile(offs, 90):.2f}")
