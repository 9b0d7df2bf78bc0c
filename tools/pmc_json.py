#!/usr/bin/env python3
"""usage: python tools/pmc_json.py <tag>  (after tools/gpu_profiles.sh <tag>): gpurun_out/<tag>_pmc_gates_<precision>_w<win>_l<len>.txt ->
profiles/<round>_pmc_gates_<precision>_w<win>_l<len>.json, the files bench.py's roofline.traffic reads (one per engine / geometry).  HBM bytes per
launch = FETCH_SIZE [KB] x 1024 x 2 (gfx950: 16 B/lane coalesced loads are counted at half size, MI355X_MICROARCH.md HBM section)
+ WRITE_SIZE [KB] x 1024 (as reported).  The raw per-kernel text is copied next to it."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATES = {"fp32": "conv3x3_wino4<0", "fp16": "conv3x3_h16<0, 3, 2, 0, 1", "bf16": "conv3x3_h16<1, 3, 2, 0, 1"}


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
    tag = sys.argv[1]
    rnd = tag.split("_")[0]
    for src in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_pmc_gates_*.txt" % tag))):
        m = re.match(r"%s_pmc_gates_(\w+?)_w(\d+)_l(\d+)\.txt" % re.escape(tag), os.path.basename(src))
        if not m:
            continue
        prec, win, length = m.group(1), int(m.group(2)), int(m.group(3))
        k = [v for name, v in parse(src).items() if name and GATES[prec] in name and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if not k:
            print("no gates kernel row in", src)
            continue
        k = k[0]
        cin = 49 if prec == "fp32" else 56                       # the 16-bit engine pads 49 -> 56 channels, hi + lo = 4 B per element
        algo = {"read": 72 * cin * (win + 2) ** 2 * 4, "write": 72 * 64 * win * (win + 2) * 4,
                "what": "read 72 window-directions x %d ch x %d^2 x 4 B, write 72 x 64 x %d x %d x 4 B (rows keep the input pitch)" % (cin, win + 2, win, win + 2)}
        traffic = int(k["FETCH_SIZE"]["avg"] * 1024 * 2 + k["WRITE_SIZE"]["avg"] * 1024)
        raw = "profiles/" + os.path.basename(src)
        shutil.copy(src, os.path.join(ROOT, raw))
        d = {"kernel": "%s...> (ConvGRU gates conv), W=%d L=%d 36 windows, tools/gpu_probe.py %d %d 36 %s" % (GATES[prec], win, length, win, length, prec),
             "method": "rocprofv3 --kernel-trace --pmc <one counter set per pass> (tools/gpu_pmc.sh via tools/gpu_profiles.sh %s); averages over %d launches "
                       "(tools/rocpd_pmc.py); a forward's step-0 launch (17 live channels) is in the average like in bench.py's mean launch" % (tag, k["FETCH_SIZE"]["n"]),
             "FETCH_SIZE_KB_avg": k["FETCH_SIZE"]["avg"], "WRITE_SIZE_KB_avg": k["WRITE_SIZE"]["avg"],
             "SQ_VALU_MFMA_BUSY_CYCLES_avg": k.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg"), "SQ_BUSY_CYCLES_avg": k.get("SQ_BUSY_CYCLES", {}).get("avg"),
             "corrections": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced loads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
             "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": algo,
             "traffic_over_algorithmic": round(traffic / (algo["read"] + algo["write"]), 3), "raw": raw}
        dst = os.path.join(ROOT, "profiles", "%s_pmc_gates_%s_w%d_l%d.json" % (rnd, prec, win, length))
        with open(dst, "w") as f:
            json.dump(d, f, indent=1)
        print(dst, traffic, d["traffic_over_algorithmic"])


if __name__ == "__main__":
    main()
