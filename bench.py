"""Benchmark of the MI355X tree-cover inference hot path (BASELINE.json metric: 10 m pixels/s; max |dprob| vs the oracle).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|fp16|bf16] [--win 172] [--length 4]
    python bench.py --preprocess-only --tiles 256          # BASELINE configs[2]: preprocessing only, HBM roofline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = `--inflight` (default 2) synthetic 618x618 tiles per GPU, each resident in HBM as the job stores it (uint16 bands,
cloud / shadow mask, DEM) and each pushed through the WHOLE built hot path by ONE C-ABI call (ttc_predict_tile, no host round
trip) on its own HIP stream and context:

    uint16 decode + S1 dB -> bilinear 20 m->10 m -> cloud / shadow gap-fill (feather, aligned mosaic, per-date NNLS fit, blend)
    -> DSen2 super-resolution (31 windows x T dates, reference tiling) -> NaN repair, date screening, indices, 12xT temporal
    operator, medians -> 36 overlapping windows -> bi-ConvGRU + U-Net forward -> post-masks -> Gaussian overlap mosaic (uint8)
    [-> RCCL gather of the finished uint8 rasters to rank 0, batched, on a side stream, when N > 1]

Tiles differ from step to step (a pool of distinct seeds per rank: tile_id = k * world + rank), so the data-dependent branches of
the gap-fill see different inputs.  Tiles shard embarrassingly (one process per GPU, static assignment, weak scaling); the only
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
H16_MFMA_PEAK_TF = 2500.0        # v_mfma_f32_32x32x16_{f16,bf16}, dense
DTYPES = {
    "fp32": "f32 (fp32 MFMA, exact fp32 FMA chains)",
    "fp16": "fp16 hi+lo operand pairs, 3 MFMA products per term, f32 accumulate / GroupNorm / state",
    "bf16": "bf16 hi+lo operand pairs, 3 MFMA products per term, f32 accumulate / GroupNorm / state",
}


def conv_gates_flops(W, n_windows):
    """algorithmic FLOPs of ONE conv_gates launch: 3x3, 49 -> 64, W^2 px, both directions"""
    return 2.0 * 9 * 49 * 64 * W * W * (2 * n_windows)


def pmc_traffic(precision, win):
    """HBM bytes per conv_gates launch from the committed PMC passes (profiles/; rocprofv3 cannot run inside the timed
    process).  Only valid for the configuration the counters were collected on."""
    name = {"fp32": "r01_c_pmc_conv_gates.json", "fp16": "r02_pmc_conv_h16_gates.json"}.get(precision)
    p = os.path.join(ROOT, "profiles", name) if name else None
    if win != 172 or not p or not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def roofline(precision, win, gates_ms, gates_n):
    """dominant kernel family = the ConvGRU gates convolution (49 -> 64, both directions, 36 windows per launch)"""
    flops = conv_gates_flops(win, 36)
    ach = flops / (gates_ms * 1e-3) / 1e12 if gates_ms > 0 else 0.0
    if precision == "fp32":
        return {"kernel": "conv3x3_f32<CK=10,NCG=2,EPI_RAW> (ConvGRU gates, 49->64, both directions)",
                "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TF,
                "traffic": pmc_traffic(precision, win), "launch_ms": gates_ms, "launches_timed": gates_n, "flops_per_launch": flops}
    # 16-bit engines: ALGORITHMIC flops against the dense 16-bit MFMA peak; the three split products and the K padding
    # (49 -> 56 channels, 9 -> 10 tap halves) that the kernel actually issues are reported beside it
    nbytes = 72.0 * (56 * (win + 2) ** 2 * 4 + 64 * win * (win + 2) * 4)          # hi+lo blocked input planes + fp32 raw output
    issued = 3.0 * flops * (56.0 / 49) * (10.0 / 9)
    name = "conv3x3_h16<TERMS=3,NCG=2,EPI_RAW>"
    return {"kernel": name + " (ConvGRU gates, 49->64, both directions)", "bound": "mfma", "achieved": ach, "peak": H16_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": ach / H16_MFMA_PEAK_TF, "traffic": pmc_traffic(precision, win), "launch_ms": gates_ms,
            "launches_timed": gates_n, "flops_per_launch": flops,
            "mfma_issue_frac": issued / (gates_ms * 1e-3) / (H16_MFMA_PEAK_TF * 1e12) if gates_ms > 0 else 0.0,
            "hbm_frac": nbytes / (gates_ms * 1e-3) / (HBM_PEAK_GBS * 1e9) if gates_ms > 0 else 0.0, "bytes_per_launch": nbytes}


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
    q = TILE // 2                                       # gap-fill on a quarter tile (x4)
    random.seed(0)
    t0 = time.time()
    _, qi, _ = G.remove_cloud_and_shadows(s2[:, :q, :q].copy(), probs[:, :q, :q].copy(), np.zeros((q, q), bool))
    tm["gapfill"] = 4.0 * (time.time() - t0)
    interp = np.zeros(probs.shape, np.float32); interp[:, :q, :q] = qi
    t0 = time.time()                                    # DSen2 on 4 of the 31 windows (x 31/4), all T dates
    for k in range(4):
        win = np.pad(s2[:, 110 * k:110 * k + 110, :110], ((0, 0), (4, 4), (4, 4), (0, 0)), "reflect")
        ds(win, win[..., 4:])
    tm["dsen2"] = (time.time() - t0) * 31.0 / 4.0
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
    ap.add_argument("--precision", choices=list(DTYPES), default="fp32",
                    help="conv engines: exact fp32 MFMA chains (BASELINE configs[1], default), fp16 / bf16 hi+lo operand pairs on the "
                         "16-bit engine (configs[4] / [3])")
    ap.add_argument("--inflight", type=int, default=2, help="tiles in flight per GPU per step, each on its own HIP stream + context")
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic tiles per rank, visited round-robin")
    ap.add_argument("--gather-batch", type=int, default=64, help="finished rasters per RCCL gather (N > 1)")
    ap.add_argument("--detect", action="store_true",
                    help="also run the multi-temporal cloud/shadow DETECTION (cloud_removal.py:1215-1677) inside the step and "
                         "gap-fill with ITS mask instead of the given one")
    ap.add_argument("--preprocess-only", action="store_true",
                    help="BASELINE configs[2]: decode + bilinear + gap-fill + temporal stage + window assembly only (no DSen2, no "
                         "model), reported against the HBM roofline (404.8 MB algorithmic bytes per T=12 tile, SURVEY 8d)")
    ap.add_argument("--tiles", type=int, default=256, help="tiles of the --preprocess-only run")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational second measurement with the other precision")
    ap.add_argument("--no-dprob", action="store_true", help="skip max |dprob| (HIP vs the oracle on windows of the bench's own tile)")
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
    dev = f"cuda:{local}"
    size = args.win - 14
    weights = Wt.synth_weights(0)

    def make_sessions(precision):
        return [job.TTCSession(weights, win_in=args.win, length=args.length, max_windows=36, device=local, precision=precision)
                for _ in range(args.inflight)]

    # ---- the tile pool: tile_id = k * world + rank, seed 1234 + tile_id; raw arrays as stored (uint16, tof_downloading.py:51-61)
    def u16(a):
        return np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)

    def make_tile(tile_id):
        s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234 + tile_id, T=args.dates, H=TILE, W=TILE)
        _, _, _, s1, dem = synth.synth_tile(seed=1234 + tile_id, T=2, H=TILE, W=TILE)
        host = (u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), probs, np.asarray(dates), u16(s1), dem)
        d = {"s2_10": torch.from_numpy(host[0].view(np.int16)).to(dev), "s2_20": torch.from_numpy(host[1].view(np.int16)).to(dev),
             "mask": torch.from_numpy(probs).to(dev), "dates": torch.tensor([int(v) for v in dates], dtype=torch.int32, device=dev),
             "s1": torch.from_numpy(host[4].view(np.int16)).to(dev), "dem": torch.from_numpy(dem).to(dev)}
        d["dem_m"] = d["dem"] * 12.0            # metres for the detector's elevation rules (the synthetic DEM is in units of 90 m)
        return host, d
    pool = [make_tile(k * world + rank) for k in range(max(1, args.pool))]
    host_tile = pool[0][0]
    flags = 0
    if args.detect:
        flags |= 1
    if args.preprocess_only:
        flags |= 2 | 4

    prio = os.environ.get("TTC_BENCH_PRIO")        # probe: "1" = slot 0 on a high-priority stream, the others default
    streams = [torch.cuda.Stream(device=local, priority=(-1 if (prio and i == 0) else 0)) for i in range(args.inflight)]
    side = torch.cuda.Stream(device=local)
    B = max(1, min(args.gather_batch, args.inflight * max(1, args.steps)))
    B -= B % args.inflight if B > args.inflight else 0
    rings = [torch.empty((B, TILE, TILE), dtype=torch.uint8, device=dev) for _ in range(2)]        # double-buffered raster batches
    status = torch.zeros((2, B, 4), dtype=torch.int32, device=dev)
    gather_bufs = [[torch.empty((B, TILE, TILE), dtype=torch.uint8, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
                   for _ in range(2)]
    state = {"pos": 0, "ring": 0, "tile": 0, "gathers": 0, "failed": 0}
    gathered = [None, None]                      # event: the last gather that read ring r has finished

    def flush(n_valid):
        """hand the finished batch to the gather on the side stream; compute continues into the other ring"""
        r = state["ring"]
        if world > 1:
            for st in streams:
                side.wait_stream(st)
            with torch.cuda.stream(side):
                shard.gather_rasters(rings[r], rank, world, 0, gather_bufs[r])        # one collective per B rasters (RCCL over xGMI)
                gathered[r] = side.record_event()
            state["gathers"] += 1
        state["ring"] ^= 1
        state["pos"] = 0
        if gathered[state["ring"]] is not None:  # the ring we are about to overwrite was handed to a gather two flushes ago
            for st in streams:
                st.wait_event(gathered[state["ring"]])

    def step(sessions):
        for slot, (sess, st) in enumerate(zip(sessions, streams)):
            tile = pool[state["tile"] % len(pool)][1]
            state["tile"] += 1
            r, p = state["ring"], state["pos"]
            with torch.cuda.stream(st):
                try:
                    t_host = time.perf_counter()
                    sess.ctx.predict_tile_raw(tile["s2_10"], tile["s2_20"], tile["s1"], tile["dem"], tile["mask"], tile["dates"],
                                              job.min_all, job.max_all, size, dem_m=tile["dem_m"], flags=flags,
                                              out=None if args.preprocess_only else rings[r][p], status=status[r, p])
                    state["host_s"] = state.get("host_s", 0.0) + time.perf_counter() - t_host
                    state["host_n"] = state.get("host_n", 0) + 1
                except RuntimeError as e:        # a failed tile must not poison the batch: record it and go on
                    state["failed"] += 1
                    print(f"[bench] rank {rank}: tile {state['tile'] - 1} failed: {e}", file=sys.stderr)
            state["pos"] += 1
            if state["pos"] == B:
                flush(B)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(sessions, steps):
        ctx = sessions[0].ctx
        state.update(pos=0, ring=0, gathers=0)
        sync()
        for _ in range(args.warmup):
            step(sessions)
        if state["pos"]:
            flush(state["pos"])
        state["gathers"] = 0
        ctx.timing(2)                 # HIP events around the conv-engine launches only (on the launch stream)
        ctx.kernel_ms(None)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(sessions)
        if state["pos"]:
            flush(state["pos"])
        sync()
        dt = time.perf_counter() - t0
        if os.environ.get("TTC_BENCH_HOSTTIME") and rank == 0 and state.get("host_n"):
            print(f"[bench] host time inside ttc_predict_tile: {state['host_s'] / state['host_n'] * 1e3:.3f} ms per call over {state['host_n']} calls "
                  f"(enqueue only, no synchronisation); wall {dt / (steps * len(sessions)) * 1e3:.3f} ms per tile", file=sys.stderr)
        gates_ms, gates_n = ctx.kernel_ms("conv_gates")
        ctx.timing(0)
        bad = int(((status[..., 0] != 0) | (status[..., 2] != 0)).sum().item())
        return shard.max_over_ranks(dt, dev, world), gates_ms, gates_n, bad

    def max_dprob(sess):
        """HIP (this precision, through the C ABI) vs the fp32 torch oracle on windows of the bench's own tile: the model inputs the
        tile path assembled for tile 0, windows 0, 14 and 35"""
        from oracle import restate_model as M
        tile = pool[0][1]
        _, _, frames, _ = sess.ctx.predict_tile_raw(tile["s2_10"], tile["s2_20"], tile["s1"], tile["dem"], tile["mask"], tile["dates"],
                                                    job.min_all, job.max_all, size, dem_m=tile["dem_m"], flags=(flags & 1) | 2, want_inputs=True)
        torch.cuda.synchronize()
        x = frames[[0, 14, 35]][:, :, :, 1:-1, 1:-1].permute(0, 1, 3, 4, 2).contiguous()      # [3, L+1, W, W, 17]
        hip = sess.ctx.forward_windows(x).cpu().numpy()
        ref = M.TreeCoverNet(weights, dtype=torch.float32)(x.cpu().numpy())[..., 0]
        return float(np.abs(hip.astype(np.float64) - ref).max())

    sessions = make_sessions(args.precision)
    if args.preprocess_only:
        steps = max(1, args.tiles // args.inflight)
        dt, _, _, bad = measure(sessions, steps)
        if rank == 0:
            n_tiles = world * steps * args.inflight
            alg = (4.0 * args.dates * 15 + 4.0 * (args.length + 1) * 17) * TILE * TILE           # raw in + model input out
            gbs = n_tiles / world * alg / dt / 1e9
            print(json.dumps({
                "metric": "10m pixels/s tree-cover preprocessing only", "value": n_tiles * TILE * TILE / dt, "unit": "px/s", "n_gpus": world,
                "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE.json configs[2]: preprocessing only on {n_tiles} synthetic 618x618 T={args.dates} tiles "
                                       f"({args.inflight} in flight per GPU): uint16 decode + S1 dB, bilinear 20 m->10 m, cloud gap-fill, NaN repair + "
                                       f"date screening, indices + 12xT temporal operator + medians, window assembly + normalisation (L={args.length})",
                           "tiles": n_tiles, "ms_per_tile": dt / (steps * args.inflight) * 1e3, "tiles_flagged_for_staged_path": bad},
                "roofline": {"kernel": "whole preprocessing chain (per tile)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": None, "bytes_per_tile": alg,
                             "note": "algorithmic bytes = raw [T,15,618,618] f32-equivalent in + model input [L+1,17,618,618] out (SURVEY 8d); "
                                     "north_star target 0.40"}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    dt, gates_ms, gates_n, bad = measure(sessions, args.steps)
    iso_ms = None
    if args.inflight > 1 and rank == 0:         # the same kernel without a second tile competing for the CUs (informational)
        c0 = sessions[0].ctx
        c0.timing(2); c0.kernel_ms(None)
        tile = pool[0][1]
        for _ in range(3):
            c0.predict_tile_raw(tile["s2_10"], tile["s2_20"], tile["s1"], tile["dem"], tile["mask"], tile["dates"], job.min_all, job.max_all,
                                size, dem_m=tile["dem_m"], flags=flags)
        torch.cuda.synchronize()
        iso_ms, _ = c0.kernel_ms("conv_gates")
        c0.timing(0)
    dprob = None
    if rank == 0 and not args.no_dprob:
        dprob = max_dprob(sessions[0])
    if world > 1:
        dist.barrier()
    alt = None
    if world == 1 and not args.no_alt:
        other = "fp16" if args.precision == "fp32" else "fp32"
        for sx in sessions:
            sx.close()
        alt_sessions = make_sessions(other)
        dt2, g2, _, _ = measure(alt_sessions, args.steps)
        alt = {"precision": other, "dtype": DTYPES[other], "value": args.inflight * TILE * TILE * args.steps / dt2, "unit": "px/s",
               "ms_per_step": dt2 / args.steps * 1e3, "conv_gates_launch_ms": g2,
               "max_dprob": None if args.no_dprob else max_dprob(alt_sessions[0]),
               "note": "same step with the other conv engine; informational, not the headline value"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "10m pixels/s tree-cover inference", "value": world * args.inflight * TILE * TILE * args.steps / dt, "unit": "px/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPES[args.precision], "data": "synthetic",
            "max_dprob": dprob,
            "config": {
                "workload": f"{args.inflight} x 618x618 tile per GPU per step, T={args.dates} dates, 36 overlapping {args.win}x{args.win} "
                            f"windows (out {size}), L={args.length}, {args.precision} (BASELINE.json configs[1])",
                "stages": ["u16_decode+s1_db", "bilinear_20m"] + (["cloud_shadow_detection"] if args.detect else []) +
                          ["cloud_gapfill(feather+aligned_mosaic+NNLS fit+blend, expected-multiplicity sampler)",
                           "dsen2_superresolve(31 windows x T)", "nan_repair+date_screening+temporal_operator+indices+medians",
                           "window_assembly+normalise", "biConvGRU+UNet forward", "post_masks", "gaussian_mosaic"]
                          + (["rccl_gather_u8(batched, side stream)"] if world > 1 else []),
                "not_in_timed_region": ["H2D of the raw tile (inputs resident in HBM as stored: uint16 bands, f32 mask / DEM)",
                                        ("-" if args.detect else "cloud/shadow DETECTION (SURVEY 8f-1, built: --detect): the mask is an input")],
                "entry": "one ttc_predict_tile call per tile (no host round trip), one HIP stream + context per tile in flight",
                "weights": "synthetic seed 0 (ConvGRU/U-Net weights absent from the reference checkout); DSen2 real",
                "tiles_per_step_per_gpu": args.inflight, "streams_per_gpu": args.inflight, "distinct_tiles_per_gpu": len(pool),
                "tile_ids": "k * world + rank", "gather_batch": B if world > 1 else None, "gathers_timed": state["gathers"] if world > 1 else None,
                "tiles_flagged_for_staged_path": bad, "tiles_failed": state["failed"],
                "max_dprob_sample": "HIP vs fp32 oracle on windows 0, 14, 35 of tile 0 (model inputs as the tile path assembled them)",
                "win_in": args.win, "length": args.length, "dates": args.dates,
            },
            "roofline": roofline(args.precision, args.win, gates_ms, gates_n),
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
