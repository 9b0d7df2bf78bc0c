"""GPU busy fraction of a rocprofv3 --kernel-trace run (e.g. of `bench.py --job-level-only`): union of the kernel intervals over the span of the
run's second half (the pipelined tile loop).  usage: python tools/probes/job_loop_busy.py <results.db>"""
import sqlite3
import sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
a = np.array(c.execute(f"select start, end from {kt} order by start").fetchall(), dtype=np.int64)


def union(b):
    busy, (cs, ce) = 0, b[0]
    for s, e in b[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


t0, T = a[:, 0].min(), a[:, 1].max() - a[:, 0].min()
b = a[a[:, 0] > t0 + 0.5 * T]
span = b[:, 1].max() - b[:, 0].min()
print(f"dispatches {len(a)}, kernel time {(a[:, 1] - a[:, 0]).sum() / 1e6:.1f} ms; second half of the run: span {span / 1e6:.1f} ms, "
      f"GPU busy {union(b) / 1e6:.1f} ms = {union(b) / span:.2f}")
