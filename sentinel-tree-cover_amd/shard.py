"""Multi-GPU sharding of the hot path: tiles are independent end to end (the reference shards them across
EC2 instances with --start/--end, job.py:1716-1717, :1869), so ranks never exchange data while computing.
The only collective is the gather of finished uint8 rasters to rank 0 (RCCL over xGMI with backend "nccl";
"gloo" on CPU in the tests)."""
from __future__ import annotations


def tiles_for_rank(n_tiles: int, rank: int, world: int):
    """static round-robin: tile_id % world == rank (uniform cost per tile -> no work stealing)"""
    return list(range(rank, n_tiles, world))


def gather_rasters(raster, rank: int, world: int, dst: int = 0, bufs=None):
    """Gather one [H, W] uint8 raster per rank to `dst`.  Returns the list (dst) or None (others).
    `bufs` lets the caller reuse pre-allocated receive buffers on the hot path."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [raster]
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(raster) for _ in range(world)]
    if raster.is_cuda and dist.get_backend() == "gloo":          # debug runs without RCCL: stage through the host
        host = [torch.empty(raster.shape, dtype=raster.dtype) for _ in range(world)] if rank == dst else None
        dist.gather(raster.cpu(), host, dst=dst)
        if rank == dst:
            for b, h in zip(bufs, host):
                b.copy_(h)
        return bufs if rank == dst else None
    dist.gather(raster, bufs if rank == dst else None, dst=dst)
    return bufs if rank == dst else None


def max_over_ranks(seconds: float, device, world: int) -> float:
    import torch
    import torch.distributed as dist
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- tile-border resegmentation: the one real exchange of the path ---------------------------------------------------------
# A border (tile t, tile t+1 of a row, src/resegment_tiles_wide.py:847) needs the first SIZE/2+7 columns of the right-hand
# tile next to the last SIZE/2+7 of the left one.  With tile_id % world sharding the two live on different GPUs, so the
# rank of t+1 sends its border columns to the rank of t, which owns the border: a point-to-point copy over xGMI
# (ncclSend / ncclRecv through torch.distributed), ~110 MB per border for T = 12 -- no collective, nothing on the tile path.
_STRIP_KEYS = ("s2", "interp", "s1", "dem")


def borders_for_rank(n_tiles: int, rank: int, world: int):
    """borders (t, t+1) of a row of n_tiles owned by this rank = the rank of their left tile"""
    return [t for t in range(n_tiles - 1) if t % world == rank]


def neighbour_strip(tile: dict, size: int):
    """what resegment_border needs of the right-hand tile: its first size//2 + 7 columns (split_fn(..., 'neighbor'), :84-93)"""
    keep = size // 2 + 7
    out = {"s2": tile["s2"][:, :, :keep], "interp": tile["interp"][:, :, :keep], "s1": tile["s1"][:, :, :keep],
           "dem": tile["dem"][:, :keep], "dates": tile["dates"]}
    return out


def exchange_border_strips(tiles: dict, n_tiles: int, rank: int, world: int, size: int, device=None):
    """tiles: {tile_id: {s2 [T, X, Y, 10], interp [T, X, Y], s1 [12, X, Y, 2], dem [X, Y] (tensors), dates (int sequence)}} for
    the tiles of this rank.  Returns {t: neighbour strip dict of tile t+1} for the borders this rank owns; strips of tiles on
    the same rank are views, the others arrive through one batch of point-to-point operations.
    Strips travel RAW (process_tile's outputs): the per-tile branch of resegment_border (fewer than 3 shared dates, :1003-1110)
    preprocesses the neighbour on its FULL tile, so for such a pair the owner of the neighbour has to run preprocess_tile first
    and send the preprocessed strip (resegment_border(..., neighb_is_strip=True) says so too); the shared-strip branch needs
    nothing of the kind."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return {t: neighbour_strip(tiles[t + 1], size) for t in range(n_tiles - 1)}
    keep = size // 2 + 7
    backend = dist.get_backend()
    if backend != "gloo" and device is None:
        raise ValueError("exchange_border_strips: `device` (the rank's GPU) is required with the nccl backend")
    too_long = [tid for tid, tl in tiles.items() if len(tl["dates"]) > 64]
    if too_long:
        raise ValueError(f"exchange_border_strips: tiles {too_long} hold more than 64 dates (the strip header carries 64)")
    stage = (lambda x: x.cpu()) if backend == "gloo" else (lambda x: x)
    sends = [t + 1 for t in range(n_tiles - 1) if (t + 1) % world == rank and t % world != rank]
    recvs = [t for t in borders_for_rank(n_tiles, rank, world) if (t + 1) % world != rank]
    # 1) headers: T and X of every strip in flight (tiles keep different numbers of dates)
    hdr_ops, hdr_out, hdr_in = [], {}, {}
    for tid in sends:
        tl = tiles[tid]
        hdr_out[tid] = torch.tensor([tl["s2"].shape[0], tl["s2"].shape[1]] + [int(d) for d in tl["dates"]] + [0] * (64 - len(tl["dates"])),
                                    dtype=torch.int64)
        hdr_ops.append(dist.P2POp(dist.isend, hdr_out[tid] if backend == "gloo" else hdr_out[tid].to(device), (tid - 1) % world, tag=tid))
    for t in recvs:
        hdr_in[t] = torch.empty(66, dtype=torch.int64, device="cpu" if backend == "gloo" else device)
        hdr_ops.append(dist.P2POp(dist.irecv, hdr_in[t], (t + 1) % world, tag=t + 1))
    if hdr_ops:
        for w in dist.batch_isend_irecv(hdr_ops):
            w.wait()
    # 2) payloads
    ops, keepalive, got = [], [], {}
    for tid in sends:
        strip = neighbour_strip(tiles[tid], size)
        for k in _STRIP_KEYS:
            buf = stage(strip[k].contiguous())
            keepalive.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, (tid - 1) % world, tag=tid))
    for t in recvs:
        h = hdr_in[t].cpu()
        T, X = int(h[0]), int(h[1])
        dev = "cpu" if backend == "gloo" else device
        bufs = {"s2": torch.empty((T, X, keep, 10), dtype=torch.float32, device=dev),
                "interp": torch.empty((T, X, keep), dtype=torch.float32, device=dev),
                "s1": torch.empty((12, X, keep, 2), dtype=torch.float32, device=dev),
                "dem": torch.empty((X, keep), dtype=torch.float32, device=dev)}
        for k in _STRIP_KEYS:
            ops.append(dist.P2POp(dist.irecv, bufs[k], (t + 1) % world, tag=t + 1))
        bufs["dates"] = [int(v) for v in h[2:2 + T]]
        got[t] = bufs
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    out = {}
    for t in borders_for_rank(n_tiles, rank, world):
        if t in got:
            out[t] = {k: (v.to(device) if (device is not None and k != "dates") else v) for k, v in got[t].items()}
        else:
            out[t] = neighbour_strip(tiles[t + 1], size)
    return out


# ---- first-contact check of the collectives this path uses -------------------------------------------------------------------
def smoke_check(rank: int, world: int, device=None, n_rasters: int = 4, tile: int = 618, size: int = 158):
    """Runs the path's two communication patterns once on the initialised process group -- the batched raster gather to rank 0
    (`gather_rasters`, the only collective of the tile path; job.py:1716-1717 / :1869 shard tiles with --start / --end instead)
    and the point-to-point border-strip exchange (`exchange_border_strips`) -- with rank-stamped data, checks every value
    that arrives and returns a dict (seconds per pattern, bytes moved, `ok`).  Raises on a mismatch.  `bench.py --gpus N > 1`
    calls it before its warm-up and tools/rccl_smoke.py prints it as one JSON line: whoever first gets a multi-GPU node learns
    in seconds whether the `nccl` (RCCL) branch works before spending a bench run on it."""
    import time
    import torch
    import torch.distributed as dist
    backend = dist.get_backend() if world > 1 else "none"
    dev = device if (device is not None and backend != "gloo") else "cpu"
    out = {"backend": backend, "world": world, "ok": False}
    # 1) gather: n_rasters x tile x tile uint8 per rank, value = 10 * rank + index
    rasters = torch.stack([torch.full((tile, tile), (10 * rank + i) % 251, dtype=torch.uint8) for i in range(n_rasters)]).to(dev)
    if dev != "cpu":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = gather_rasters(rasters, rank, world, 0)
    if dev != "cpu":
        torch.cuda.synchronize()
    out["gather_s"] = time.perf_counter() - t0
    out["gather_bytes_per_rank"] = int(rasters.numel())
    if rank == 0:
        for r, g in enumerate(got):
            want = torch.tensor([(10 * r + i) % 251 for i in range(n_rasters)], dtype=torch.uint8)
            if not (torch.equal(g[:, 0, 0].cpu(), want) and torch.equal(g[:, -1, -1].cpu(), want) and int(g.cpu().to(torch.int64).sum()) == int(want.to(torch.int64).sum()) * tile * tile):
                raise RuntimeError(f"smoke_check: gathered rasters of rank {r} are wrong")
    # 2) border strips: one tile per rank in a row of `world` tiles, tile t keeps 3 + t % 2 dates, values stamped with the tile id
    X = 64
    T = 3 + rank % 2
    mine = {rank: {"s2": torch.full((T, X, X, 10), float(rank), dtype=torch.float32, device=dev),
                   "interp": torch.full((T, X, X), rank + 0.5, dtype=torch.float32, device=dev),
                   "s1": torch.full((12, X, X, 2), -float(rank), dtype=torch.float32, device=dev),
                   "dem": torch.full((X, X), 100.0 + rank, dtype=torch.float32, device=dev),
                   "dates": [10 * rank + k for k in range(T)]}}
    t0 = time.perf_counter()
    strips = exchange_border_strips(mine, world, rank, world, size=32, device=None if dev == "cpu" else dev) if world > 1 else {}
    if dev != "cpu":
        torch.cuda.synchronize()
    out["strips_s"] = time.perf_counter() - t0
    for t, sdict in strips.items():
        n = t + 1
        Tn = 3 + n % 2
        ok = (sdict["dates"] == [10 * n + k for k in range(Tn)] and sdict["s2"].shape[0] == Tn
              and float(sdict["s2"].min()) == float(sdict["s2"].max()) == float(n)
              and float(sdict["interp"].max()) == n + 0.5 and float(sdict["s1"].min()) == -float(n) and float(sdict["dem"].max()) == 100.0 + n)
        if not ok:
            raise RuntimeError(f"smoke_check: border strip of tile {n} arrived wrong on rank {rank}")
    out["strips_received"] = len(strips)
    # 3) the timing reduction bench.py uses
    out["max_over_ranks_ok"] = max_over_ranks(1.0 + rank, dev, world) == float(world)
    if not out["max_over_ranks_ok"]:
        raise RuntimeError("smoke_check: all_reduce(MAX) returned the wrong value")
    out["ok"] = True
    return out
