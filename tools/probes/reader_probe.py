"""Reader-side throughput of the job-level loop, no GPU work: job.iter_raw_tiles into a PinnedArena over the bench's tile folders, for several
(reader threads, inflate threads) settings.  usage: python tools/probes/reader_probe.py"""
import importlib.util
import os
import shutil
import sys
import tempfile
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ttc  # noqa
from ttc import job, synth
spec = importlib.util.spec_from_file_location("write_hdf5_fixture", os.path.join(ROOT, "tools", "write_hdf5_fixture.py"))
WF = importlib.util.module_from_spec(spec)
spec.loader.exec_module(WF)
TILE, T = 618, 12


def u16(a):
    return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)


root = tempfile.mkdtemp(prefix="ttc_rd_") + "/"


def dump(path, arr, chunks=None):
    w = WF.Writer()
    w.finish({"data": w.chunked_dataset(arr, chunks=chunks) if chunks else w.contiguous_dataset(arr)}, path)


n_tiles = 6
for k in range(n_tiles):
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234 + k % 4, T=T, H=TILE, W=TILE)
    _, _, _, s1, dem = synth.synth_tile(seed=1234 + k % 4, T=2, H=TILE, W=TILE)
    f, idx = f"{root}{k}/0/raw/", f"{k}X0Y"
    dump(f"{f}s2_10/{idx}.hkl", u16(s2[..., :4]), (1, 155, 155, 4))
    dump(f"{f}s2_20/{idx}.hkl", u16(s2[:, ::2, ::2, 4:]), (1, 155, 155, 6))
    dump(f"{f}s1/{idx}.hkl", u16(s1), (1, 155, 155, 2))
    dump(f"{f}clouds/clouds_{idx}.hkl", probs.astype(np.float32), (1, 155, 155))
    dump(f"{f}misc/dem_{idx}.hkl", (dem * 90.0).astype(np.float32), (155, 155))
    dump(f"{f}misc/s2_dates_{idx}.hkl", np.asarray(dates, dtype=np.int64))
try:
    t0 = time.perf_counter(); job.load_raw_tile(0, 0, root); print(f"one tile alone: {(time.perf_counter() - t0) * 1e3:.1f} ms (first touch)")
    t0 = time.perf_counter(); job.load_raw_tile(0, 0, root); print(f"one tile alone: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    for readers, inflate, mode, use_arena in ((4, 8, "0", True), (8, 8, "0", True), (4, 8, "1", True), (8, 8, "1", True), (16, 8, "1", True),
                                              (4, 8, "0", False), (8, 8, "0", False), (8, 8, "1", False)):
        os.environ["TTC_IO_THREADS"] = str(inflate)
        os.environ["TTC_HKL_READ"] = mode
        arena = job.PinnedArena(torch, 2 * readers + 4) if use_arena else None
        coords = [(k % n_tiles, 0) for k in range(8 + 48)]
        n = 0
        for raw in job.iter_raw_tiles(coords, root, workers=readers, arena=arena):
            n += 1
            if n == 8:
                t0 = time.perf_counter()
            if arena is not None:
                arena.release(raw["_arena_set"])
        dt = time.perf_counter() - t0
        print(f"readers {readers:2d} x inflate {inflate:2d}, TTC_HKL_READ={mode}, arena {use_arena}: {dt / 48 * 1e3:.1f} ms per tile", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
