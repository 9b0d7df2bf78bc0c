"""CPU, world_size 2, gloo: the N > 1 path of bench.py (static tile sharding + raster gather + max-over-ranks)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ttc  # noqa: F401
from ttc import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.tiles_for_rank(7, rank, world)
        last = None
        for tile_id in mine[:7 // world]:      # every rank runs the same number of collective steps (as bench.py does)
            raster = torch.full((6, 5), tile_id, dtype=torch.uint8)
            got = shard.gather_rasters(raster, rank, world)
            if rank == 0:
                last = [int(g[0, 0]) for g in got]
        t = shard.max_over_ranks(1.0 + rank, "cpu", world)
        out.put((rank, mine, last, t))
    finally:
        dist.destroy_process_group()


def test_shard_gather_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, tiles0, last0, t0), (r1, tiles1, last1, t1) = res
    assert tiles0 == [0, 2, 4, 6] and tiles1 == [1, 3, 5]
    assert sorted(tiles0 + tiles1) == list(range(7))
    assert last0 == [4, 5]                   # third joint step: rank 0 holds tile 4, rank 1 tile 5
    assert t0 == t1 == 2.0                   # max over ranks


def test_single_rank_is_passthrough():
    r = torch.zeros((3, 3), dtype=torch.uint8)
    assert shard.gather_rasters(r, 0, 1)[0] is r
    assert shard.tiles_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert shard.max_over_ranks(0.5, "cpu", 1) == 0.5


def _tile(tid, T, X=20, Y=40):
    g = torch.Generator().manual_seed(100 + tid)
    return {"s2": torch.rand((T, X, Y, 10), generator=g), "interp": torch.rand((T, X, Y), generator=g),
            "s1": torch.rand((12, X, Y, 2), generator=g), "dem": torch.rand((X, Y), generator=g), "dates": [10 * tid + 3 * i for i in range(T)]}


def _border_worker(rank, world, port, out, n_tiles=5):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        size = 30                              # strips of size // 2 + 7 = 22 columns; tiles keep 3.. dates
        mine = {t: _tile(t, 3 + t) for t in shard.tiles_for_rank(n_tiles, rank, world)}
        strips = shard.exchange_border_strips(mine, n_tiles, rank, world, size)
        ok = {}
        for t, st in strips.items():
            want = shard.neighbour_strip(_tile(t + 1, 3 + t + 1), size)
            ok[t] = all(torch.equal(st[k], want[k]) for k in ("s2", "interp", "s1", "dem")) and list(st["dates"]) == list(want["dates"])
        out.put((rank, sorted(strips), ok))
    finally:
        dist.destroy_process_group()


def test_border_strip_exchange_two_ranks():
    """resegmentation: the neighbour's border columns travel point-to-point to the rank that owns the border"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_border_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, b0, ok0), (r1, b1, ok1) = res
    assert b0 == [0, 2] and b1 == [1, 3]                # border t belongs to the rank of tile t
    assert all(ok0.values()) and all(ok1.values())
    assert shard.borders_for_rank(5, 0, 1) == [0, 1, 2, 3]
    one = shard.exchange_border_strips({t: _tile(t, 4) for t in range(3)}, 3, 0, 1, 30)
    assert sorted(one) == [0, 1] and one[0]["s2"].shape == (4, 20, 22, 10)


def test_border_strip_exchange_three_ranks():
    """world 3 (neither 1 nor 2): a rank both sends and receives in the same batch, every border has its two tiles on
    different ranks, and rank 2 owns fewer tiles than the others (7 tiles -> 3 / 2 / 2)"""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_border_worker, args=(r, world, port, q, 7)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    owned = [r[1] for r in res]
    assert owned == [[0, 3], [1, 4], [2, 5]]            # border t (tiles t, t + 1) belongs to rank t % 3
    assert all(all(r[2].values()) for r in res)


def test_comm_smoke_check_two_ranks_gloo():
    """tools/rccl_smoke.py (shard.smoke_check: verified raster gather + border-strip exchange + max-over-ranks) over gloo, the CPU
    rehearsal of what bench.py --gpus N > 1 runs first over RCCL"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_smoke.py"), "--spawn", "2", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["ranks_ok"] == [0, 1] and out["rccl_smoke"]["strips_received"] == 1
