"""Summarise a TTC_WINO_TRACE dump (per-workgroup s_memtime stamps of the Winograd fp32 kernel: thread 0 of each workgroup, the
workgroup's third tile).  usage: python tools/probes/wino_trace_stats.py <file> [chunks]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64).astype(np.int64)
fine = (a[:, 56] > 0).any()
whole = a[a[:, 61] > 0]
if len(whole):
    nt = whole[:, 31].astype(float)
    names8 = ["chunk loop", "wait exchange barrier 1", "row transform + exchange writes", "wait exchange barrier 2", "exchange reads + adds",
              "epilogue op / args / GroupNorm sums", "output stores", "tile advance (tile_of)"]
    per = [np.median(whole[:, k] / nt) for k in (7, 47, 55, 32, 15, 33, 23, 39)]
    print("cycles per tile, wave 0 of each workgroup, mean over its whole walk (median over workgroups):")
    for n, v in zip(names8, per):
        print(f"  {n:38s} {v:8.0f}")
    print(f"  {'sum':38s} {sum(per):8.0f}   tiles/workgroup {np.median(nt):.0f}")
    ref = (whole[:, 61] - whole[:, 60]) / 100e6
    clk = (whole[:, 63] - whole[:, 62]) / ref
    print(f"  whole walk {np.median(ref) * 1e3:.3f} ms (median), shader clock {np.median(clk) / 1e9:.3f} GHz")
if not fine:
    a = a[:0]
a = a[a[:, 56] > 0]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 7
names = ["wait barrier A", "transform (+B loads)", "MFMA half 1", "wait barrier B", "stage store/load issue", "MFMA half 2"]
tot = np.zeros(6)
for c in range(min(T, 7) if len(a) else 0):
    t = a[:, 8 * c:8 * c + 7]
    d = np.diff(t, axis=1)
    med = np.median(d, axis=0)
    tot += med
    print(f"chunk {c}: " + "  ".join(f"{n} {m:.0f}" for n, m in zip(names, med)))
if len(a): print("sum over traced chunks:", {n: float(v) for n, v in zip(names, tot)})
if len(a): print("tile: chunk loop", np.median(a[:, 57] - a[:, 56]), " output transform + exchange", np.median(a[:, 58] - a[:, 57]), " epilogue (op, stats, stores)",
      np.median(a[:, 59] - a[:, 58]), " total", np.median(a[:, 59] - a[:, 56]), f"  ({len(a)} workgroups)")
print(f"own MFMA issue per tile: {T - 1} x 32 + last chunk, x 64 cycles = {((T - 1) * 32 + 8) * 64} (gates)")
if a.shape[1] >= 64 and (a[:, 61] > 0).any():
    b = a[a[:, 61] > 0]
    ref = (b[:, 61] - b[:, 60]) / 100e6          # s_memrealtime: 100 MHz
    clk = (b[:, 63] - b[:, 62]) / ref
    print(f"whole walk per workgroup: {np.median(ref) * 1e3:.3f} ms (median), shader clock {np.median(clk) / 1e9:.3f} GHz (p10 {np.percentile(clk, 10) / 1e9:.3f}, p90 {np.percentile(clk, 90) / 1e9:.3f})")
    print(f"shader cycles per workgroup walk: {np.median(b[:, 63] - b[:, 62]):.0f}")
