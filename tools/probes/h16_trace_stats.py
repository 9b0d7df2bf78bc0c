"""Summarise a TTC_H16_TRACE dump (per-workgroup s_memtime stamps of the 16-bit gates kernel, wave 0 of each workgroup)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64).astype(np.int64)
a = a[a[:, 0] > 0]
seg = np.zeros((len(a), 7, 6))
for c in range(7):
    t = a[:, 8 + c * 6:8 + c * 6 + 6]
    for i in range(5):
        seg[:, c, i] = t[:, i + 1] - t[:, i]
names = ["wait_vm", "barrier1", "prod1", "barrier2", "prod23"]
print("median ticks per chunk segment, chunks 1..5:", {n: float(np.median(seg[:, 1:6, i])) for i, n in enumerate(names)},
      "chunk", float(np.median(seg[:, 1:6, :5].sum(-1))))
print("startup", np.median(a[:, 1] - a[:, 0]), "loop", np.median(a[:, 2] - a[:, 1]), "epilogue", np.median(a[:, 3] - a[:, 2]),
      "drain", np.median(a[:, 4] - a[:, 3]), "total", np.median(a[:, 4] - a[:, 0]))
e = a[:, 52:58]
print("epilogue stats split: accumulate", float(np.median(a[:, 58] - a[:, 52])), "dpp", float(np.median(a[:, 59] - a[:, 58])), "stores", float(np.median(a[:, 53] - a[:, 59])))
print("epilogue: stats, ->sync, lds write, sync, stores (group 0):", [float(np.median(e[:, i + 1] - e[:, i])) for i in range(5)])
