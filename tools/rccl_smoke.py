"""First-contact kit for a multi-GPU node: do the path's collectives work over RCCL?

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/rccl_smoke.py
    python tools/rccl_smoke.py --spawn 2 --backend gloo          # CPU rehearsal of the same code (no GPU needed)

Runs `shard.smoke_check` (sentinel-tree-cover_amd/shard.py): the batched uint8 raster gather to rank 0 (the tile path's only
collective) and the point-to-point border-strip exchange of the resegmentation, with rank-stamped data that is verified on
arrival, then rank 0 prints ONE JSON line.  `bench.py --gpus N > 1` runs the same check before its warm-up.
The reference shards tiles over EC2 instances with --start / --end (src/download_and_predict_job.py:1716-1717, :1869); this is
the replacement's communication surface, exercised in ~10 s.
"""
import argparse
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, backend, port=None, queue=None):
    import torch
    import torch.distributed as dist
    import ttc  # noqa: F401
    from ttc import shard
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL on this image
    if port is not None:
        os.environ["MASTER_PORT"] = str(port)
    device = None
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        device = f"cuda:{local}"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        res = shard.smoke_check(rank, world, device)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    if queue is not None:
        queue.put((rank, res))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--spawn", type=int, default=0, help="spawn this many ranks locally instead of reading RANK / WORLD_SIZE")
    args = ap.parse_args()
    if args.spawn:
        import torch.multiprocessing as mp
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=run, args=(r, args.spawn, args.backend, port, q)) for r in range(args.spawn)]
        [p.start() for p in procs]
        res = dict(q.get(timeout=300) for _ in procs)
        [p.join(60) for p in procs]
        ok = all(p.exitcode == 0 for p in procs) and all(v["ok"] for v in res.values())
        print(json.dumps({"rccl_smoke": res[0], "ranks_ok": sorted(k for k, v in res.items() if v["ok"]), "ok": ok}))
        sys.exit(0 if ok else 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    res = run(rank, world, args.backend)
    if rank == 0:
        print(json.dumps({"rccl_smoke": res, "ok": res["ok"]}))


if __name__ == "__main__":
    main()
