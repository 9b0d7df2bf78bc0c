#!/usr/bin/env python3
"""usage: python tools/profile_ref.py <tag> [<tag2> ...]   (after tools/gpu_profiles.sh <tag>; later tags override the legs they re-profiled)

<dir>/<tag>_{isolated,live}_<leg>_kernel_stats.md  (rocprofv3 --kernel-trace --stats summaries of `bench.py --profile-leg ...` runs, written by
tools/rocpd_stats.py)  ->  profiles/<round>_bench_profile.json: per leg the command, the stats file and every kernel's calls / average.
bench.py reads the newest such file and reports, next to roofline.frac, whether its own HIP-event time of the ConvGRU gates launch agrees
with the committed row (`roofline.profile`, `roofline.live_profile`) -- the fraction is then recomputable from a file under profiles/."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = {"fp32": "--precision fp32", "fp16": "--precision fp16", "bf16": "--precision bf16"}


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            name = re.sub(r"^void ", "", m.group(1))
            name = re.sub(r"^\(anonymous namespace\)::", "", name)
            rows[name] = {"calls": int(m.group(2)), "total_ms": float(m.group(3)), "avg_us": float(m.group(4)), "min_us": float(m.group(5)),
                          "max_us": float(m.group(6))}
    return rows


def main():
    tags = sys.argv[1:]
    d = os.path.join(ROOT, "profiles")
    out = {"tags": tags, "made_by": "tools/profile_ref.py from tools/gpu_profiles.sh %s (rocprofv3 --kernel-trace --stats, one bench leg per run)" % " / ".join(tags),
           "isolated": {}, "live": {}}
    for tag in tags:
        for path in sorted(glob.glob(os.path.join(d, "%s_*_kernel_stats.md" % tag))):
            m = re.match(r"%s_(isolated|live)_(w(\d+)_l(\d+)_(\w+))_kernel_stats\.md" % re.escape(tag), os.path.basename(path))
            if not m:
                continue
            leg, key, win, length, prec = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), m.group(5)
            steps = "--steps 10" if leg == "isolated" else "--steps 20 --warmup 3"
            out[leg][key] = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --profile-leg %s %s --win %d --length %d %s" % (leg, steps, win, length, FLAGS[prec]),
                             "stats_file": "profiles/" + os.path.basename(path), "kernels": parse(path)}
    dst = os.path.join(ROOT, "profiles", "%s_bench_profile.json" % tags[-1].split("_")[0])
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(dst, {k: sorted(v) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
