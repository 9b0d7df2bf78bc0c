#!/usr/bin/env python3
"""usage: python tools/pmc_json.py <tag>  (after tools/gpu_profiles.sh <tag>): gpurun_out/<tag>_pmc_{f32,h16}_gates.txt ->
profiles/<round>_pmc_conv_{f32,h16}_gates.json, the files bench.py's roofline.traffic reads.  HBM bytes per launch =
FETCH_SIZE [KB] x 1024 x 2 (gfx950: 16 B/lane coalesced loads are counted at half size, MI355X_MICROARCH.md HBM section)
+ WRITE_SIZE [KB] x 1024 (as reported)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out, kern = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("==") or not line.strip():
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
        if m:
            out.setdefault(kern, {})[m.group(1)] = {"n": int(m.group(2)), "avg": float(m.group(3))}
        else:
            kern = line.strip()
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05_a"
    rnd = tag.split("_")[0]
    algo = {"f32": {"read": 72 * 49 * 174 ** 2 * 4, "write": 72 * 64 * 172 * 174 * 4,
                    "what": "read 72 planes-sets x 49 ch x 174^2 x 4 B, write 72 x 64 x 172 x 174 x 4 B (input pitch)"},
            "h16": {"read": 72 * 56 * 174 ** 2 * 4, "write": 72 * 64 * 172 * 174 * 4,
                    "what": "read 72 x 56 ch (49 padded) x 174^2 x (hi + lo) 4 B, write 72 x 64 x 172 x 174 x (top + bottom halves) 4 B"}}
    done = set()
    for eng, sub in (("f32", "conv3x3_wino4<0"), ("f32", "conv3x3_wino<2, 0"), ("f32", "conv3x3_f32<10, 2, 0"), ("h16", "conv3x3_h16<0, 3, 2, 0, 1")):
        if eng in done:                                                                # the first kernel of an engine that ran is the product's
            continue
        src = os.path.join(ROOT, "gpurun_out", "%s_pmc_%s_gates.txt" % (tag, eng))
        if not os.path.exists(src):
            continue
        k = [v for name, v in parse(src).items() if name and sub in name]
        if not k:
            continue
        k = k[0]
        done.add(eng)
        traffic = int(k["FETCH_SIZE"]["avg"] * 1024 * 2 + k["WRITE_SIZE"]["avg"] * 1024)
        d = {"kernel": "%s...> (ConvGRU gates conv), W=172 L=4 36 windows, tools/gpu_probe.py 172 4 36 %s" % (sub, "fp32" if eng == "f32" else "fp16"),
             "method": "rocprofv3 --kernel-trace --pmc <one counter set per pass> (tools/gpu_pmc.sh via tools/gpu_profiles.sh %s); "
                       "averages over %d launches (tools/rocpd_pmc.py)" % (tag, k["FETCH_SIZE"]["n"]),
             "FETCH_SIZE_KB_avg": k["FETCH_SIZE"]["avg"], "WRITE_SIZE_KB_avg": k["WRITE_SIZE"]["avg"],
             "SQ_VALU_MFMA_BUSY_CYCLES_avg": k.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg"),
             "SQ_BUSY_CYCLES_avg": k.get("SQ_BUSY_CYCLES", {}).get("avg"),
             "corrections": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced loads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
             "traffic_bytes_per_launch": traffic,
             "algorithmic_bytes_per_launch": algo[eng],
             "traffic_over_algorithmic": round(traffic / (algo[eng]["read"] + algo[eng]["write"]), 3),
             "raw": "profiles/%s_pmc_%s_gates.txt" % (tag, eng)}
        dst = os.path.join(ROOT, "profiles", "%s_pmc_conv_%s_gates.json" % (rnd, eng))
        with open(dst, "w") as f:
            json.dump(d, f, indent=1)
        print(dst, traffic, d["traffic_over_algorithmic"])


if __name__ == "__main__":
    main()
