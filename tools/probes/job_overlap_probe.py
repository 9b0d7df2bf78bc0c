"""Which part of the job-level tile loop serialises the three tile streams?  The headline loop (inputs resident) runs 15.4 ms per tile with three
tiles in flight, job.predict_tiles 22 ms with the mask given.  Modes, all on three sessions / three torch streams, mask given, 48 tiles each:
  A resident inputs, one ttc_predict_tile per tile (= bench.py's step)        B  A + the DEM median / divide prologue of the job loop
  C  B + per-tile H2D of the raw arrays from pinned memory in the tile's stream (fresh device tensors, as _PinnedStager does)
  D  C with the copies into per-slot device buffers allocated once              E  D with the copies on ONE upload stream + event
  F  C + reading the status words back 6 tiles late (= predict_tiles)           G  F with E's upload stream
usage: python tools/probes/job_overlap_probe.py"""
import os
import sys
import time
from collections import deque
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc  # noqa
from ttc import job, synth, weights as Wt
dev = "cuda:0"
torch.cuda.set_device(0)
TILE, T = 618, 12
W = Wt.synth_weights(0)
sessions = [job.TTCSession(W, win_in=172, length=4, max_windows=36, device=0, precision="fp32") for _ in range(3)]
streams = [torch.cuda.Stream(device=0) for _ in range(3)]
up = torch.cuda.Stream(device=0)


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


def make_tile(tile_id):
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234 + tile_id, T=T, H=TILE, W=TILE)
    _, _, _, s1, dem = synth.synth_tile(seed=1234 + tile_id, T=2, H=TILE, W=TILE)
    host = {"s2_10": u16(s2[..., :4]).view(np.int16), "s2_20": u16(s2[:, ::2, ::2, 4:]).view(np.int16), "mask": probs.astype(np.float32),
            "dates": np.asarray(dates, dtype=np.int32), "s1": u16(s1).view(np.int16), "dem": (dem * 90.0).astype(np.float32)}
    pinned = {k: torch.from_numpy(v).pin_memory() for k, v in host.items()}
    d = {k: v.to(dev) for k, v in pinned.items()}
    return pinned, d


pool = [make_tile(k) for k in range(4)]
slots = [{k: torch.empty_like(v, device=dev) for k, v in pool[0][0].items()} for _ in range(6)]
size = 158
N = 48
hbufs = [(torch.empty(4, dtype=torch.int32).pin_memory(), torch.empty((TILE, TILE), dtype=torch.uint8).pin_memory(), torch.empty((TILE, TILE), dtype=torch.float32).pin_memory()) for _ in range(8)]


def tile_call(sess, d, dem90, dem_m):
    return sess.ctx.predict_tile_raw(d["s2_10"], d["s2_20"], d["s1"], dem90, d["mask"], d["dates"], job.min_all, job.max_all, size, dem_m=dem_m, flags=0, want_float=True)


def run(mode):
    pending = deque()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        sess, st = sessions[k % 3], streams[k % 3]
        pinned, res = pool[k % 4]
        if mode in "AB":
            d = res
        elif mode in "CFHIJ":
            with torch.cuda.stream(st):
                d = {n: v.to(dev, non_blocking=True) for n, v in pinned.items()}
        elif mode == "D":
            d = slots[k % 6]
            with torch.cuda.stream(st):
                for n, v in pinned.items():
                    d[n].copy_(v, non_blocking=True)
        else:                                              # E, G: one upload stream, the tile's stream waits for its event
            d = slots[k % 6]
            with torch.cuda.stream(up):
                up.wait_stream(st)                         # the slot's previous reader (same stream, two tiles ago) must be done
                for n, v in pinned.items():
                    d[n].copy_(v, non_blocking=True)
                ev = up.record_event()
            st.wait_event(ev)
        with torch.cuda.stream(st):
            if mode == "A":
                dem90, dem_m = d["dem"] if False else res["dem"] / 90.0, res["dem"]
            else:
                dem_m = sess.ctx.median5(d["dem"])
                dem90 = sess.ctx.divide(dem_m.clone(), 90.0)
            u8, f32, _, status = tile_call(sess, d, dem90, dem_m)
        if mode in "FGHIJ":
            ev = None
            if mode == "I":                                # results into pinned host buffers by non-blocking copies in the tile's stream + an event
                with torch.cuda.stream(st):
                    hb = hbufs[k % 8]
                    hb[0].copy_(status, non_blocking=True); hb[1].copy_(u8, non_blocking=True); hb[2].copy_(f32, non_blocking=True)
                    ev = st.record_event()
            pending.append((k, u8, f32, status, ev))
            if len(pending) > 6:
                kk, a, b, s, e = pending.popleft()
                if mode == "H":                            # only wait for the tile's stream
                    streams[kk % 3].synchronize()
                elif mode == "I":
                    e.synchronize()
                elif mode == "J":                          # status words only
                    with torch.cuda.stream(streams[kk % 3]):
                        s.cpu()
                else:
                    with torch.cuda.stream(streams[kk % 3]):
                        s.cpu(); a.cpu(); b.cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


for m in "AACFHIJFIC":
    print(f"mode {m}: {run(m):.2f} ms per tile", flush=True)
