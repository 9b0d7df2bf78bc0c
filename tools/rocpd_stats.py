"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel stats table
(the `--stats` view): calls, total / average / min / max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof_X/X_results.db > profiles/r01_X_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
