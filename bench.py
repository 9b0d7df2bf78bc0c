"""Benchmark of the MI355X tree-cover inference hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--win 172] [--length 4] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the built hot path over one synthetic 618x618 tile that is already
resident in HBM (config.stages lists exactly what runs inside the timed region):

    bilinear 20 m->10 m  ->  cloud / shadow gap-fill (feather, aligned mosaic, per-date NNLS fit, blend)
    ->  DSen2 super-resolution (31 windows x T dates, reference tiling)
    ->  repair / indices / 12xT temporal operator / medians  ->  36 overlapping windows
    ->  bi-ConvGRU + U-Net forward (fp32 MFMA)  ->  post-masks  ->  Gaussian overlap mosaic
    [-> RCCL gather of the uint8 raster to rank 0 when N > 1]

Tiles shard embarrassingly (one process per GPU, static assignment, weak scaling); the only
collective is the gather of finished rasters.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE = 618
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3        # v_mfma_f32_32x32x2_f32, dense


def pmc_traffic(args):
    """HBM bytes per conv_gates launch from the committed PMC passes (profiles/; rocprofv3 cannot run inside the
    timed process).  Only valid for the configuration the counters were collected on."""
    name = "r01_c_pmc_conv_gates.json" if args.precision == "fp32" else "r01_f_pmc_conv_b3_gates.json"
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    if args.win != 172 or not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def roofline(args, gates_ms, gates_n):
    """dominant kernel = the ConvGRU gates convolution (49 -> 64, both directions, 36 windows per launch)."""
    W = args.win
    flops = conv_gates_flops(W, 36)
    if args.precision == "fp32":
        ach = flops / (gates_ms * 1e-3) / 1e12 if gates_ms > 0 else 0.0
        return {"kernel": "conv3x3_f32<CK=10,NCG=2,EPI_RAW> (ConvGRU gates, 49->64, both directions)",
                "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TF,
                "traffic": pmc_traffic(args), "launch_ms": gates_ms, "launches_timed": gates_n, "flops_per_launch": flops}
    # split-bf16 engine: 3 bf16 MFMAs per term put the matrix floor (0.18 ms) below the HBM floor of the fp32 activations
    nbytes = 4.0 * 72 * (49 * (W + 2) ** 2 + 64 * W * W)          # algorithmic: padded input planes + output planes
    ach = nbytes / (gates_ms * 1e-3) / 1e9 if gates_ms > 0 else 0.0
    return {"kernel": "conv3x3_b3<NCG=2,EPI_RAW> (ConvGRU gates, 49->64, both directions, split-bf16 MFMA)",
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args), "launch_ms": gates_ms, "launches_timed": gates_n, "bytes_per_launch": nbytes,
            "mfma_bf16_frac": 3.0 * flops * (56.0 / 49) * (10.0 / 9) / (gates_ms * 1e-3) / 2.5e15 if gates_ms > 0 else 0.0}


def conv_gates_flops(W, n_windows):
    """algorithmic FLOPs of ONE conv_gates launch: 3x3, 49 -> 64, W^2 px, both directions"""
    return 2.0 * 9 * 49 * 64 * W * W * (2 * n_windows)


def cpu_baseline(args, tile):
    """The oracle (CPU restatement of the reference; kind = "port") on the host cores, on a BOUNDED sample of the
    same tile, stage by stage, extrapolated to one whole tile (factors stated in `sample`)."""
    import random
    import torch
    from oracle import restate_gapfill as G, restate_model as M, restate_numpy as O
    from ttc import weights as Wt
    s2_10, s2_20, probs, dates, s1, dem = tile
    w = Wt.synth_weights(0)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    ds = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    size, T = args.win - 14, args.dates
    tm = {}
    t0 = time.time()
    s2_10, s2_20, s1 = O.to_float32(s2_10), O.to_float32(s2_20), O.s1_to_db(s1)
    tm["codecs"] = time.time() - t0
    t0 = time.time(); s2 = O.upsample_20m(s2_10, s2_20); tm["bilinear"] = time.time() - t0
    # gap-fill on a quarter tile (x4)
    q = TILE // 2
    random.seed(0)
    t0 = time.time()
    _, qi, _ = G.remove_cloud_and_shadows(s2[:, :q, :q].copy(), probs[:, :q, :q].copy(), np.zeros((q, q), bool))
    tm["gapfill"] = 4.0 * (time.time() - t0)
    interp = np.zeros(probs.shape, np.float32); interp[:, :q, :q] = qi
    # DSen2 on 4 of the 31 windows (x 31/4), all T dates
    t0 = time.time()
    for k in range(4):
        win = np.pad(s2[:, 110 * k:110 * k + 110, :110], ((0, 0), (4, 4), (4, 4), (0, 0)), "reflect")
        ds(win, win[..., 4:])
    tm["dsen2"] = (time.time() - t0) * 31.0 / 4.0
    # process_subtiles numerics on the whole tile with a stub model, + the model on 6 of the 36 windows (x6)
    feeds = []

    def stub(x):
        feeds.append(x)
        return np.full((size, size), 0.5, np.float32)
    t0 = time.time()
    wins = O.process_subtiles(s2, dates.copy(), interp, s1.copy(), dem.copy(), stub, size=size, length=args.length)
    tm["preprocess+post"] = time.time() - t0
    t0 = time.time()
    for x in feeds[:6]:
        O.predict_subtile(x, net, size)
    tm["model"] = (time.time() - t0) * 36.0 / max(1, min(6, len(feeds)))
    t0 = time.time(); O.mosaic_predictions(wins, size=size); tm["mosaic"] = time.time() - t0
    total = sum(tm.values())
    return {"value": TILE * TILE / total, "unit": "px/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "oracle/ on one 618x618 T=%d tile, extrapolated per stage: gap-fill on a quarter tile x4, DSen2 on 4 of 31 "
                      "windows x7.75, ConvGRU/U-Net on 6 of 36 windows x6, other stages whole; seconds per tile: %s"
                      % (T, json.dumps({k: round(v, 2) for k, v in tm.items()}))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--win", type=int, default=172, help="model input window (172 = reference default; 168 also legal)")
    ap.add_argument("--length", type=int, default=4, help="ConvGRU steps (reference default 4; 12 = monthly)")
    ap.add_argument("--dates", type=int, default=12, help="raw acquisition dates T")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="fp32",
                    help="conv engine: exact fp32 MFMA chains (BASELINE configs[1], default) or split-bf16 MFMA (3 products per "
                         "term, fp32 accumulate; max |dprob| 7e-5 vs fp32)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="tiles in flight per GPU per step, each on its own HIP stream + context: one tile's latency-bound "
                         "gap-fill / tile kernels run under another tile's convolutions")
    ap.add_argument("--detect", action="store_true",
                    help="also run the multi-temporal cloud/shadow DETECTION (cloud_removal.py:1215-1677, the row after SURVEY 8's "
                         "a1-a20) inside the step and gap-fill with ITS mask instead of the given one")
    ap.add_argument("--from-host", action="store_true",
                    help="informational: every step uploads the raw tile from pinned host memory first (PCIe-inclusive rate, "
                         "reported in DESIGN.md, never the headline value)")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational second measurement with the other conv engine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import ttc  # noqa: F401
    from ttc import job, shard, synth, weights as Wt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # debugging aid for 1-GPU boxes: TTC_BENCH_BACKEND=gloo + TTC_BENCH_DEVICE=0 runs every rank on one device with CPU-staged
    # collectives, which exercises the multi-rank control flow (barriers, gathers, rank-0-only sections) without RCCL
    backend = os.environ.get("TTC_BENCH_BACKEND", "nccl")
    if "TTC_BENCH_DEVICE" in os.environ:
        local = int(os.environ["TTC_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    size = args.win - 14
    def make_sessions(precision):
        return [job.TTCSession(Wt.synth_weights(0), win_in=args.win, length=args.length, max_windows=36, device=local,
                               precision=precision) for _ in range(args.inflight)]
    sessions = make_sessions(args.precision)
    streams = [torch.cuda.Stream(device=local) for _ in range(args.inflight)] if args.inflight > 1 else [None]

    # synthetic tile, seed 1234 + tile_id (tile_id = rank): 10 m bands, 20 m bands, interp, S1, DEM -> HBM
    # (cloudy S2 stack + binary cloud/shadow mask from synth_gapfill_scene; S1 / DEM from synth_tile)
    s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234 + rank, T=args.dates, H=TILE, W=TILE)
    _, _, _, s1, dem = synth.synth_tile(seed=1234 + rank, T=2, H=TILE, W=TILE)
    # the raw tile as stored (uint16, src/tof/tof_downloading.py:51-61): decoding is part of the step
    def u16(a):
        return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)
    s2_10 = u16(s2[..., :4])
    s2_20 = u16(s2[:, ::2, ::2, 4:])
    s1 = u16(s1)
    host_tile = (s2_10, s2_20, probs, dates, s1, dem)
    dev = f"cuda:{local}"
    d10, d20 = torch.from_numpy(s2_10.view(np.int16)).to(dev), torch.from_numpy(s2_20.view(np.int16)).to(dev)
    dprobs, ds1, ddem = torch.from_numpy(probs).to(dev), torch.from_numpy(s1.view(np.int16)).to(dev), torch.from_numpy(dem).to(dev)
    ddem_m = ddem * 12.0                    # metres for the detector's elevation rules (synthetic DEM is in units of 90 m)
    gather_bufs = [[torch.empty((TILE, TILE), dtype=torch.uint8, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
                   for _ in range(args.inflight)]

    def step(sessions):
        for slot, (sess, st) in enumerate(zip(sessions, streams)):
            if st is None:
                tile_step(sess, slot)
            else:
                with torch.cuda.stream(st):
                    tile_step(sess, slot)

    pinned = None
    if args.from_host:
        pinned = [torch.from_numpy(a).pin_memory() for a in (s2_10.view(np.int16), s2_20.view(np.int16), probs, s1.view(np.int16), dem)]

    def tile_step(sess, slot, gather=True):
        ctx = sess.ctx
        if pinned is not None:                      # H2D on the tile's own stream: overlaps the other tile's kernels
            d10_, d20_, dprobs_, ds1_, ddem_ = (t.to(dev, non_blocking=True) for t in pinned)
            return tile_body(sess, slot, gather, d10_, d20_, dprobs_, ds1_, ddem_)
        return tile_body(sess, slot, gather, d10, d20, dprobs, ds1, ddem)

    def tile_body(sess, slot, gather, d10, d20, dprobs, ds1, ddem):
        ctx = sess.ctx
        f10, f20, s1db = ctx.to_float32(d10), ctx.to_float32(d20), ctx.s1_to_db(ds1)   # tof_downloading.py:64-72, job.py:699-708
        s2d = ctx.upsample_20m(f10, f20)                              # job.py:734-782
        mask, pf = dprobs, None
        if args.detect:                                               # cloud_removal.py:1215-1677 (process_tile: job.py:837)
            mask, pf = ctx.identify_clouds_shadows(s2d, ddem_m, None, None)
        dint, _, _ = ctx.remove_cloud_and_shadows(s2d, mask, pf, None)   # cloud_removal.py:888-973 (deterministic sampler)
        ctx.superresolve_tile(s2d, quirks=True)                       # job.py:95-147
        f32, u8 = job.predict_tile(s2d, dates, dint, s1db, ddem, sess, size=size, to_host=False)   # job.py:1125-1641
        if world > 1 and gather:
            shard.gather_rasters(u8, rank, world, 0, gather_bufs[slot])     # final-mosaic gather (RCCL over xGMI)
        return u8

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(sessions):
        ctx = sessions[0].ctx
        sync()
        for _ in range(args.warmup):
            step(sessions)
        ctx.timing(2)                 # HIP events around the conv-engine launches only (on the launch stream)
        ctx.kernel_ms(None)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(sessions)
        sync()
        dt = time.perf_counter() - t0
        gates_ms, gates_n = ctx.kernel_ms("conv_gates")
        ctx.timing(0)
        return shard.max_over_ranks(dt, dev, world), gates_ms, gates_n

    dt, gates_ms, gates_n = measure(sessions)
    # the same kernel without a second tile competing for the CUs (informational; the roofline line uses the live number)
    iso_ms = None
    if args.inflight > 1 and rank == 0:
        c0 = sessions[0].ctx
        c0.timing(2); c0.kernel_ms(None)
        for _ in range(3):
            tile_step(sessions[0], 0, gather=False)      # rank-local: no collective here
        torch.cuda.synchronize()
        iso_ms, _ = c0.kernel_ms("conv_gates")
        c0.timing(0)
    if world > 1:
        dist.barrier()
    alt = None
    if world == 1 and not args.no_alt:
        other = "bf16x3" if args.precision == "fp32" else "fp32"
        for sx in sessions:
            sx.close()
        dt2, g2, _ = measure(make_sessions(other))
        alt = {"precision": other, "value": args.inflight * TILE * TILE * args.steps / dt2, "unit": "px/s", "ms_per_step": dt2 / args.steps * 1e3,
               "conv_gates_launch_ms": g2,
               "note": "same step with the other conv engine; informational, not the headline value"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        ach = conv_gates_flops(args.win, 36) / (gates_ms * 1e-3) / 1e12 if gates_ms > 0 else 0.0
        out = {
            "metric": "10m pixels/s tree-cover inference", "value": world * args.inflight * TILE * TILE * args.steps / dt, "unit": "px/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 storage/accumulate, split-bf16 (bf16x3) MFMA products",
            "data": "synthetic",
            "config": {
                "workload": f"{args.inflight} x 618x618 tile per GPU per step, T={args.dates} dates, 36 overlapping {args.win}x{args.win} "
                            f"windows (out {size}), L={args.length}, {args.precision} (BASELINE.json configs[1])",
                "stages": ["u16_decode+s1_db", "bilinear_20m"] + (["cloud_shadow_detection"] if args.detect else []) + [ "cloud_gapfill(feather+aligned_mosaic+NNLS fit+blend, expected-multiplicity sampler)",
                           "dsen2_superresolve(31 windows x T)", "temporal_operator+indices+medians",
                           "window_assembly+normalise", "biConvGRU+UNet forward", "post_masks", "gaussian_mosaic"]
                          + (["rccl_gather_u8"] if world > 1 else []),
                "not_in_timed_region": [("nothing: the raw tile is uploaded from pinned host memory every step" if args.from_host
                                         else "H2D of the raw tile (inputs resident in HBM)"),
                                        ("-" if args.detect else "cloud/shadow DETECTION (SURVEY 8f-1, built: --detect): the mask is an input")],
                "weights": "synthetic seed 0 (ConvGRU/U-Net weights absent from the reference checkout); DSen2 real",
                "tiles_per_step_per_gpu": args.inflight, "streams_per_gpu": args.inflight, "win_in": args.win, "length": args.length, "dates": args.dates,
            },
            "roofline": roofline(args, gates_ms, gates_n),
        }
        if iso_ms:
            out["roofline"]["isolated_launch_ms"] = iso_ms
            out["roofline"]["isolated_frac"] = out["roofline"]["frac"] * gates_ms / iso_ms
            out["roofline"]["note"] = ("launch_ms / frac are live values with %d tiles in flight (kernels of the other tile share the CUs); "
                                       "isolated_* = the same launch with one tile in flight" % args.inflight)
        if alt:
            out["alt_precision"] = alt
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, host_tile)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
