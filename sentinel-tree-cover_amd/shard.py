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
