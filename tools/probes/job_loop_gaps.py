"""Where the GPU idles in the job-level tile loop (rocprofv3 --kernel-trace --memory-copy-trace of `bench.py --job-level-only`).
Groups HIP streams into loops (the three session streams of one predict_tiles call carry the same number of dispatches), and for each loop prints
span, kernel time, busy union, the idle gaps > 0.3 ms with the kernel that ended before / started after each, and the copies inside the gap.
usage: python tools/probes/job_loop_gaps.py <results.db>"""
import sqlite3
import sys
import collections
import numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
mt = [t for t in tabs if t.startswith("rocpd_memory_copy")][0]
st = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
names = {}
if st:
    cols = [r[1] for r in c.execute(f"pragma table_info({st[0]})")]
    nm = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    for i, n in c.execute(f"select id, {nm} from {st[0]}"):
        names[i] = n
k = np.array(c.execute(f"select start, end, stream_id, kernel_id from {kt} order by start").fetchall(), dtype=np.int64)
m = np.array(c.execute(f"select start, end, size, stream_id from {mt} order by start").fetchall(), dtype=np.int64)
cnt = collections.Counter(k[:, 2].tolist())
print("dispatches per stream:", dict(cnt))
groups = collections.defaultdict(list)
for s, n in cnt.items():
    groups[round(n, -2)].append(s)          # streams of one loop have (nearly) the same count


def short(i):
    n = names.get(int(i), str(i))
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:48]


for g, ss in sorted(groups.items()):
    if len(ss) < 2 or g < 2000:
        continue
    x = k[np.isin(k[:, 2], ss)]
    lo, hi = x[:, 0].min(), x[:, 1].max()
    allk = k[(k[:, 0] >= lo) & (k[:, 1] <= hi)]
    order = np.argsort(allk[:, 0])
    allk = allk[order]
    span = hi - lo
    # union
    u, cs, ce, last = [], allk[0, 0], allk[0, 1], 0
    ends = []                                # (gap start, gap end, index of kernel that ended last, index of next kernel)
    lastk = 0
    for i in range(1, len(allk)):
        s, e = allk[i, 0], allk[i, 1]
        if s > ce:
            ends.append((ce, s, lastk, i))
            u.append((cs, ce)); cs, ce, lastk = s, e, i
        elif e > ce:
            ce, lastk = e, i
    u.append((cs, ce))
    busy = sum(b - a for a, b in u)
    print(f"\n== loop on streams {sorted(ss)}: span {span/1e6:.1f} ms, kernel time {(allk[:,1]-allk[:,0]).sum()/1e6:.1f} ms, busy {busy/1e6:.1f} ms = {busy/span:.3f}; "
          f"dispatches {len(allk)}")
    gaps = [(b - a, a, b, i, j) for a, b, i, j in ends]
    tot = sum(g_[0] for g_ in gaps)
    print(f"   idle {tot/1e6:.1f} ms in {len(gaps)} gaps; > 0.3 ms: {sum(1 for g_ in gaps if g_[0] > 3e5)} gaps, {sum(g_[0] for g_ in gaps if g_[0] > 3e5)/1e6:.1f} ms; "
          f"<= 20 us: {sum(g_[0] for g_ in gaps if g_[0] <= 2e4)/1e6:.1f} ms; 20 us - 0.3 ms: {sum(g_[0] for g_ in gaps if 2e4 < g_[0] <= 3e5)/1e6:.1f} ms")
    byname = collections.Counter()
    bynext = collections.Counter()
    for d, a, b, i, j in gaps:
        if d > 3e5:
            byname[short(allk[i, 3])] += d
            bynext[short(allk[j, 3])] += d
    print("   idle > 0.3 ms by the kernel that ended before the gap:")
    for n, d in byname.most_common(8):
        print(f"      {d/1e6:8.1f} ms  {n}")
    print("   ... by the kernel that started after the gap:")
    for n, d in bynext.most_common(8):
        print(f"      {d/1e6:8.1f} ms  {n}")
    big = sorted(gaps, reverse=True)[:12]
    print("   largest gaps (ms since loop start, length ms, before[stream] -> after[stream], copies inside: n, MB, ms):")
    for d, a, b, i, j in sorted(big, key=lambda t: t[1]):
        mm = m[(m[:, 1] > a) & (m[:, 0] < b)]
        print(f"      {(a-lo)/1e6:8.1f} {d/1e6:6.2f}  {short(allk[i,3])}[{allk[i,2]}] -> {short(allk[j,3])}[{allk[j,2]}]  copies {len(mm)}, {mm[:,2].sum()/1e6:.0f} MB, {(np.minimum(mm[:,1],b)-np.maximum(mm[:,0],a)).sum()/1e6:.2f} ms")
    # tile cadence: the first kernel of each tile call on a stream ~ k_decode / first dispatch after a long gap on that stream
    for s in sorted(ss):
        xs = allk[allk[:, 2] == s]
        g2 = xs[1:, 0] - xs[:-1, 1]
        print(f"   stream {s}: kernel time {(xs[:,1]-xs[:,0]).sum()/1e6:.1f} ms, own-stream gaps > 1 ms: {(g2 > 1e6).sum()} totalling {g2[g2 > 1e6].sum()/1e6:.1f} ms")
    mm = m[(m[:, 0] >= lo) & (m[:, 1] <= hi)]
    print(f"   copies in the loop: {len(mm)}, {mm[:,2].sum()/1e6:.0f} MB, {(mm[:,1]-mm[:,0]).sum()/1e6:.1f} ms; by stream: " +
          ", ".join(f"{s}: {len(mm[mm[:,3]==s])}/{mm[mm[:,3]==s][:,2].sum()/1e6:.0f} MB" for s in np.unique(mm[:, 3])))
