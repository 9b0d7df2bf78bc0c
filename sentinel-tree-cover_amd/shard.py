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
