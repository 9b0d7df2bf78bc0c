"""Who holds the GPU in the headline step?  (rocprofv3 --kernel-trace of `bench.py --profile-leg live`.)
A sweep over kernel start / end stamps splits the timed span into slices with a constant set of running kernels and books every slice to
  conv only | conv + other | other only | idle
and, per kernel family, the time it ran ALONE (nothing else on the GPU), the time it shared, and its dispatch count.
usage: python tools/probes/live_timeline.py <results.db> [tiles]"""
import collections
import sqlite3
import sys

import numpy as np

c = sqlite3.connect(sys.argv[1])
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
st = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
names = {}
if st:
    cols = [r[1] for r in c.execute(f"pragma table_info({st[0]})")]
    nm = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    for i, n in c.execute(f"select id, {nm} from {st[0]}"):
        names[i] = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
k = np.array(c.execute(f"select start, end, stream_id, kernel_id from {kt} order by start").fetchall(), dtype=np.int64)
# the timed region: the densest part -- drop the first 15 % and last 5 % of dispatches (warm-up, isolated tails)
lo, hi = k[int(len(k) * 0.15), 0], k[int(len(k) * 0.95), 1]
k = k[(k[:, 0] >= lo) & (k[:, 1] <= hi)]
is_conv = np.array([names.get(int(i), "").startswith("conv3x3") for i in k[:, 3]])
ev = []
for idx, (s, e, _, _) in enumerate(k):
    ev.append((s, 1, idx)); ev.append((e, 0, idx))
ev.sort()
running = set()
book = collections.Counter()
alone = collections.Counter(); shared = collections.Counter(); cnt = collections.Counter()
for i in k[:, 3]:
    cnt[names.get(int(i), str(i))] += 1
prev = ev[0][0]
for t, kind, idx in ev:
    d = t - prev
    if d > 0:
        if not running:
            book["idle"] += d
        else:
            nc = sum(1 for r in running if is_conv[r])
            no = len(running) - nc
            book["conv only" if no == 0 else ("other only" if nc == 0 else "conv + other")] += d
            if nc >= 2:
                book["(of which >= 2 convs together)"] += d
            for r in running:
                (alone if len(running) == 1 else shared)[names.get(int(k[r, 3]), "?")] += d
    if kind:
        running.add(idx)
    else:
        running.discard(idx)
    prev = t
span = hi - lo
print(f"span {span/1e6:.1f} ms, {len(k)} dispatches" + (f", ~{span/1e6/tiles:.2f} ms per tile" if tiles else ""))
for key in ("conv only", "conv + other", "other only", "idle", "(of which >= 2 convs together)"):
    print(f"  {key:32s} {book[key]/1e6:9.2f} ms  {100*book[key]/span:5.1f} %")
print("per kernel: dispatches, total ms, ms ALONE on the GPU, ms shared  (sorted by alone)")
tot = collections.Counter()
for n in set(list(alone) + list(shared)):
    tot[n] = alone[n] + shared[n]
for n, a in sorted(alone.items(), key=lambda kv: -kv[1])[:45]:
    print(f"  {cnt[n]:6d} {tot[n]/1e6:9.2f} {a/1e6:9.2f} {shared[n]/1e6:9.2f}  {n[:70]}")
