"""Sum of a PMC counter over EVERY dispatch of a rocprofv3 rocpd database, and per kernel:
    python tools/rocpd_pmc_total.py <db> <counter> [tiles]   -> one JSON object on stdout
Used for the preprocessing chain's HBM traffic (VERDICT r5 #4d): FETCH_SIZE / WRITE_SIZE (KB) summed over all kernels of
`bench.py --preprocess-only --tiles N --inflight 1`, divided by the tiles the run processed (warm-up included: pass the total)."""
import collections
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
want = sys.argv[2]
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
ix = {c: i for i, c in enumerate(cols)}
name_c = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "counter" not in c][0]
cn_c = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
val_c = "value" if "value" in ix else "counter_value"
per = collections.defaultdict(lambda: [0, 0.0])
for r in cur.execute("select * from counters_collection"):
    if r[ix[cn_c]] != want:
        continue
    k = r[ix[name_c]]
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = k.split("(")[0]
    per[k][0] += 1
    per[k][1] += float(r[ix[val_c]])
total = sum(v[1] for v in per.values())
print(json.dumps({"counter": want, "tiles": tiles, "total": total, "per_tile": total / tiles, "dispatches": sum(v[0] for v in per.values()),
                  "per_kernel_per_tile": {k: round(v[1] / tiles, 1) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]}}))
