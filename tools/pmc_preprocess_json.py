#!/usr/bin/env python3
"""usage: python tools/pmc_preprocess_json.py <tag>   (after ONLY=pmc_pre_total bash tools/gpu_profiles.sh <tag>)
gpurun_out/<tag>_pmc_preprocess_total_{FETCH_SIZE,WRITE_SIZE}.json -> profiles/<round>_pmc_preprocess_total.json: the HBM bytes the WHOLE
preprocessing chain moves per tile (every kernel of `bench.py --preprocess-only`, separate --pmc passes; FETCH_SIZE [KB] x 1024 x 2 -- gfx950
counts 16 B/lane coalesced loads at half size, MI355X_MICROARCH.md HBM section -- + WRITE_SIZE [KB] x 1024) against the 405 MB of
SURVEY 8(d); bench.py quotes it as preprocess_only.roofline.traffic."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ALG = (4.0 * 12 * 15 + 4.0 * 5 * 17) * 618 * 618
f, w = (json.load(open(os.path.join(ROOT, "gpurun_out", "%s_pmc_preprocess_total_%s.json" % (tag, c)))) for c in ("FETCH_SIZE", "WRITE_SIZE"))
fetch, write = f["per_tile"] * 1024 * 2, w["per_tile"] * 1024
out = {"what": "HBM traffic of the whole preprocessing chain per 618x618 T=12 tile: sum over every dispatch of bench.py --preprocess-only --tiles 8 "
               "--inflight 1 --warmup 1 (9 tiles), one counter per rocprofv3 --pmc pass",
       "tag": tag, "dispatches_per_tile": f["dispatches"] / f["tiles"],
       "fetch_bytes_per_tile": fetch, "write_bytes_per_tile": write, "traffic_bytes_per_tile": fetch + write,
       "corrections": "FETCH_SIZE x 2 (gfx950, 16 B/lane coalesced loads), WRITE_SIZE as reported; FETCH_SIZE uncorrected would be %.0f MB" % (fetch / 2 / 1e6),
       "algorithmic_bytes_per_tile": ALG, "traffic_over_algorithmic": (fetch + write) / ALG,
       "largest_readers_KB_per_tile": dict(list(f["per_kernel_per_tile"].items())[:12]),
       "largest_writers_KB_per_tile": dict(list(w["per_kernel_per_tile"].items())[:12])}
dst = os.path.join(ROOT, "profiles", "%s_pmc_preprocess_total.json" % tag.split("_")[0])
json.dump(out, open(dst, "w"), indent=1)
print(dst, "traffic %.0f MB = %.2f x algorithmic" % ((fetch + write) / 1e6, out["traffic_over_algorithmic"]))
