"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --job-level-only`: GPU busy fraction, copy time and how much of it
overlaps kernels, per-stream occupancy, idle-gap histogram.  usage: python tools/probes/job_loop_timeline.py <results.db> [frac_from] [frac_to]"""
import sqlite3
import sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t and "rocpd_kernel_dispatch" in t]
kt = kt[0] if kt else [t for t in tabs if "kernel_dispatch" in t][0]
kcols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
print("kernel table", kt, kcols)
qcol = "queue_id" if "queue_id" in kcols else None
scol = "stream_id" if "stream_id" in kcols else qcol
rows = c.execute(f"select start, end, {scol or 0} from {kt} order by start").fetchall()
a = np.array(rows, dtype=np.int64)
mt = [t for t in tabs if "memory_copy" in t]
m = None
if mt:
    mcols = [r[1] for r in c.execute(f"pragma table_info({mt[0]})")]
    print("copy table", mt[0], mcols)
    sz = "size" if "size" in mcols else ("bytes" if "bytes" in mcols else "0")
    m = np.array(c.execute(f"select start, end, {sz} from {mt[0]} order by start").fetchall(), dtype=np.int64)


def union_iv(b):
    out, (cs, ce) = [], b[0]
    for s, e in b[1:]:
        if s > ce:
            out.append((cs, ce)); cs, ce = s, e
        else:
            ce = max(ce, e)
    out.append((cs, ce))
    return np.array(out, dtype=np.int64)


t0, T = a[:, 0].min(), a[:, 1].max() - a[:, 0].min()
lo, hi = t0 + f0 * T, t0 + f1 * T
b = a[(a[:, 0] >= lo) & (a[:, 1] <= hi)]
span = b[:, 1].max() - b[:, 0].min()
u = union_iv(b[:, :2])
busy = (u[:, 1] - u[:, 0]).sum()
print(f"window [{f0},{f1}] of the run: span {span/1e6:.1f} ms, dispatches {len(b)}, kernel time {(b[:,1]-b[:,0]).sum()/1e6:.1f} ms, GPU busy {busy/1e6:.1f} ms = {busy/span:.3f}")
gaps = u[1:, 0] - u[:-1, 1]
for th in (5e3, 2e4, 1e5, 5e5, 2e6):
    print(f"  idle gaps > {th/1e3:.0f} us: {int((gaps > th).sum())}, total {gaps[gaps > th].sum()/1e6:.1f} ms")
# concurrency: time-weighted number of kernels in flight
ev = np.concatenate([np.stack([b[:, 0], np.ones(len(b), np.int64)], 1), np.stack([b[:, 1], -np.ones(len(b), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
lvl = np.cumsum(ev[:, 1])[:-1]
dt = np.diff(ev[:, 0])
for k in range(0, 6):
    print(f"  {k} kernels in flight: {dt[lvl == k].sum()/1e6:.1f} ms")
print("per stream/queue: dispatches, kernel ms")
for s in np.unique(b[:, 2]):
    x = b[b[:, 2] == s]
    print(f"  {s}: {len(x)}  {(x[:,1]-x[:,0]).sum()/1e6:.1f} ms")
if m is not None and len(m):
    mm = m[(m[:, 0] >= lo) & (m[:, 1] <= hi)]
    big = mm[mm[:, 2] > 1 << 20]
    print(f"copies: {len(mm)}, total {(mm[:,1]-mm[:,0]).sum()/1e6:.1f} ms, {mm[:,2].sum()/1e6:.0f} MB; > 1 MB: {len(big)}, {(big[:,1]-big[:,0]).sum()/1e6:.1f} ms, "
          f"{big[:,2].sum()/max(1,(big[:,1]-big[:,0]).sum()):.1f} GB/s while active")
    # overlap of copies with kernel-busy intervals
    ov = 0
    for s, e, _ in mm:
        i = np.searchsorted(u[:, 1], s)
        while i < len(u) and u[i, 0] < e:
            ov += min(e, u[i, 1]) - max(s, u[i, 0]); i += 1
    print(f"  copy time overlapped by kernels: {ov/1e6:.1f} ms")
