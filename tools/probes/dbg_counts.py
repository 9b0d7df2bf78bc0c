import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ttc
from ttc import job, synth, weights as Wt
sess = job.TTCSession(Wt.synth_weights(0), win_in=172, length=4, max_windows=36)
ctx = sess.ctx
T, X = 9, 618
s2, dates, probs, _ = synth.synth_gapfill_scene(seed=1234, T=T, H=X, W=X)
_, _, _, s1, dem = synth.synth_tile(seed=1234, T=2, H=X, W=X)
s2[3, 100:400, :, :] = 0.0
u16 = lambda a: np.trunc(np.clip(a, 0, 1) * 65535).astype(np.uint16)
s2_10, s2_20, s1u = u16(s2[..., :4]), u16(s2[:, ::2, ::2, 4:]), u16(s1)
d10 = torch.from_numpy(s2_10.view(np.int16)).cuda(); d20 = torch.from_numpy(s2_20.view(np.int16)).cuda()
f10, f20 = ctx.to_float32(d10), ctx.to_float32(d20)
s2d = ctx.upsample_20m(f10, f20)
print("after upsample counts", ctx.tile_missing_counts(s2d))
dint, _, _ = ctx.remove_cloud_and_shadows(s2d, probs, None, None)
print("after gapfill counts", ctx.tile_missing_counts(s2d), "interp mean per date", dint.mean(dim=(1,2)).cpu().numpy().round(3))
ctx.superresolve_tile(s2d, quirks=True)
print("after dsen2 counts", ctx.tile_missing_counts(s2d))
u8, f32, frames, status = ctx.predict_tile_raw(s2_10, s2_20, s1u, dem, probs, dates, job.min_all, job.max_all, 158)
torch.cuda.synchronize(); print("status", status.cpu().numpy())
