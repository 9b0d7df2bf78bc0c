"""GPU-box probe: per-stage device+host time of one bench step (wall clock with a sync per stage) and the kernel
families inside (library HIP-event timers, level 1)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttc  # noqa
from ttc import job, synth, weights as Wt

TILE = 618
PREC = os.environ.get('TTC_PREC', 'fp32')
sess = job.TTCSession(Wt.synth_weights(0), win_in=172, length=4, max_windows=36, precision=PREC)
ctx = sess.ctx
s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234, T=12, H=TILE, W=TILE)
_, _, _, s1, dem = synth.synth_tile(seed=1234, T=2, H=TILE, W=TILE)
d10 = torch.from_numpy(np.ascontiguousarray(s2[..., :4])).cuda()
d20 = torch.from_numpy(np.ascontiguousarray(s2[:, ::2, ::2, 4:])).cuda()
dprobs, ds1, ddem = torch.from_numpy(probs).cuda(), torch.from_numpy(s1).cuda(), torch.from_numpy(dem).cuda()


def stages(tm):
    def lap(name, t0):
        torch.cuda.synchronize(); tm[name] = tm.get(name, 0) + time.perf_counter() - t0
    t0 = time.perf_counter(); s2d = ctx.upsample_20m(d10, d20); lap("upsample", t0)
    t0 = time.perf_counter(); dint, _, _ = ctx.remove_cloud_and_shadows(s2d, dprobs, None, None); lap("gapfill", t0)
    t0 = time.perf_counter(); ctx.superresolve_tile(s2d, quirks=True); lap("superresolve", t0)
    t0 = time.perf_counter(); f32, u8 = job.predict_tile(s2d, dates, dint, ds1, ddem, sess, size=158, to_host=False); lap("predict_tile", t0)


for _ in range(3):
    stages({})
tm = {}
K = 10
for _ in range(K):
    stages(tm)
for k, v in tm.items():
    print(f"{k:14s} {v / K * 1e3:8.3f} ms")
print("sum", sum(tm.values()) / K * 1e3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    s2d = ctx.upsample_20m(d10, d20); dint, _, _ = ctx.remove_cloud_and_shadows(s2d, dprobs, None, None)
    ctx.superresolve_tile(s2d, quirks=True); job.predict_tile(s2d, dates, dint, ds1, ddem, sess, size=158, to_host=False)
torch.cuda.synchronize(); print("async step", (time.perf_counter() - t0) / K * 1e3, "ms")
ctx.timing(1)
stages({}); stages({})
names = sys.argv[1:] or []
import ctypes as C
for k in ["upsample_20m", "feather", "aligned_mosaic", "gapfill_dates", "clouds_in_mosaic", "dsen2_gather", "dsen2_border", "dsen2_conv",
          "dsen2_scatter", "missing_counts", "fix_missing", "tile_temporal", "tile_s1", "assemble", "bright", "post",
          "frames_from_nhwc", "frames_to_b16", "conv_gates", "conv_cand", "gn_finalize", "gru_apply1", "gru_apply2", "conv_median", "conv_concat",
          "conv1", "conv2", "up2", "up2_out", "up3", "out_conv", "block_finalize", "head", "mosaic"] + names:
    try:
        ms, n = ctx.kernel_ms(k)
    except Exception as e:
        continue
    if n:
        print(f"  {k:20s} avg {ms:8.3f} ms x {n / 2:5.1f}/step = {ms * n / 2:8.3f} ms")
