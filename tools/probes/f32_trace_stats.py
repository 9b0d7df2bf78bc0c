"""Summarise a TTC_F32_TRACE dump (per-workgroup s_memtime stamps of the planar fp32 gates kernel, wave 0 of each workgroup)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64).astype(np.int64)
a = a[a[:, 0] > 0]
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 5          # chunks of the traced layer (stamps exist for the first 8)
kp = int(sys.argv[3]) if len(sys.argv) > 3 else 5           # k-pairs per tap and chunk (CK / 2)
nst = min(nch, 8)
seg = np.zeros((len(a), nst, 4))
for c in range(nst):
    t = a[:, 8 + 4 * c:8 + 4 * c + 4]
    seg[:, c, 0] = t[:, 1] - t[:, 0]        # global loads issued -> barrier (previous chunk's MFMA reads done, loads waited)
    seg[:, c, 1] = t[:, 2] - t[:, 1]        # LDS writes + barrier
    seg[:, c, 2] = t[:, 3] - t[:, 2]        # MFMA block
    if c + 1 < nst: seg[:, c, 3] = a[:, 8 + 4 * (c + 1)] - t[:, 3]
names = ["loads->sync", "lds write+sync", "mfma", "gap"]
for c in sorted({0, 1, 2, min(nch, 8) - 1}):
    print(f"chunk {c}: ", {n: float(np.median(seg[:, c, i])) for i, n in enumerate(names)})
print("per tile: start->chunk0", np.median(a[:, 8] - a[:, 0]), "loop", np.median(a[:, 2] - a[:, 8]), "epilogue", np.median(a[:, 3] - a[:, 2]),
      "drain", np.median(a[:, 4] - a[:, 3]), "total", np.median(a[:, 4] - a[:, 0]))
e = a[:, 52:60]
print("epilogue: stats(accum+dpp, stores) ", float(np.median(e[:, 7] - e[:, 0])), float(np.median(e[:, 1] - e[:, 7])),
      " group 0: ->sync, lds write, sync, stores:", [float(np.median(e[:, i + 1] - e[:, i])) for i in range(1, 5)])
print(f"own MFMA issue per tile: {nch} chunks x 9 taps x {kp} k-pairs x 8 MFMAs x 64 cycles =", nch * 9 * kp * 8 * 64)
