import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ttc
from ttc import job, synth, weights as Wt
sess = job.TTCSession(Wt.synth_weights(0), win_in=44, length=4, dsen2_weights=None)
img, dem, forest, core, near = synth.synth_detection_scene(5, 12, 618, 618)
dimg = torch.from_numpy(img).cuda(); ddem = torch.from_numpy(dem).cuda()
for urban in (None, (core, near)):
    for _ in range(2):
        sess.ctx.identify_clouds_shadows(dimg, ddem, forest, urban)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        c, f = sess.ctx.identify_clouds_shadows(dimg, ddem, forest, urban)
    torch.cuda.synchronize()
    print("urban" if urban else "no urban", (time.perf_counter() - t0) / 5 * 1e3, "ms; cloud fraction per date", c.mean(dim=(1, 2)).cpu().numpy().round(3))
if len(sys.argv) > 1:
    from oracle import restate_clouds as C
    t0 = time.time(); wc, wf = C.identify_clouds_shadows(img.copy(), dem.copy(), forest, (core, near)); print("oracle s", time.time() - t0)
    print("diff clouds", ((c.cpu().numpy() > 0) != (wc > 0)).mean(), "fcps", ((f.cpu().numpy() > 0) != wf).mean())
