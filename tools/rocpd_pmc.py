"""Per-kernel PMC averages from a rocprofv3 rocpd database:  python tools/rocpd_pmc.py <db> [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "counter" not in c][0]
cn_c = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
val_c = "value" if "value" in ix else "counter_value"
dur = None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r[ix[name_c]]
    if sub in k:
        agg[k][r[ix[cn_c]]].append(r[ix[val_c]])
for k, d in agg.items():
    print(k[:120])
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
