"""Is the preprocessing-only leg bound by the host's enqueue rate?  Enqueue N tiles (flags INPUTS_ONLY | NO_SUPERRES) round-robin on K sessions /
streams; host seconds spent enqueueing vs total.  usage: python tools/probes/preprocess_host_probe.py"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa
from ttc import job, synth, weights as Wt
dev = "cuda:0"
torch.cuda.set_device(0)
TILE, T = 618, 12
W = Wt.synth_weights(0)


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


def make_tile(tile_id):
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234 + tile_id, T=T, H=TILE, W=TILE)
    _, _, _, s1, dem = synth.synth_tile(seed=1234 + tile_id, T=2, H=TILE, W=TILE)
    host = {"s2_10": u16(s2[..., :4]).view(np.int16), "s2_20": u16(s2[:, ::2, ::2, 4:]).view(np.int16), "mask": probs.astype(np.float32),
            "dates": np.asarray(dates, dtype=np.int32), "s1": u16(s1).view(np.int16), "dem": dem.astype(np.float32)}
    return {k: torch.from_numpy(v).to(dev) for k, v in host.items()}


pool = [make_tile(k) for k in range(4)]
for K in (1, 3, 6):
    sessions = [job.TTCSession(W, win_in=172, length=4, max_windows=36, device=0, precision="fp32") for _ in range(K)]
    streams = [torch.cuda.Stream(device=0) for _ in range(K)]
    for N in (6, 60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(N):
            d = pool[k % 4]
            with torch.cuda.stream(streams[k % K]):
                sessions[k % K].ctx.predict_tile_raw(d["s2_10"], d["s2_20"], d["s1"], d["dem"], d["mask"], d["dates"], job.min_all, job.max_all, 158,
                                                     dem_m=d["dem"], flags=6)
        h = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        if N == 60:
            print(f"{K} streams: host enqueue {h / N * 1e3:.2f} ms per tile, total {tot / N * 1e3:.2f} ms per tile", flush=True)
    for s in sessions:
        s.close()
