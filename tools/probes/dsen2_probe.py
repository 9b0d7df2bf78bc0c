"""GPU-box probe: DSen2 super-resolution of one T = 12 tile, device time per conv launch and in total, per precision."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttc import job
for prec in sys.argv[1:] or ["fp32", "fp16", "bf16"]:
    sess = job.TTCSession(None, win_in=44, length=2, max_windows=1, precision=prec)
    rng = np.random.default_rng(0)
    d = torch.from_numpy((rng.random((12, 618, 618, 10)) * 0.6).astype(np.float32)).cuda()
    for _ in range(2): sess.ctx.superresolve_tile(d, quirks=True)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): sess.ctx.superresolve_tile(d, quirks=True)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    sess.ctx.timing(True)
    for _ in range(3): sess.ctx.superresolve_tile(d, quirks=True)
    ms, n = sess.ctx.kernel_ms("dsen2_conv")
    print(f"{prec}: tile {dt*1e3:.2f} ms; dsen2_conv avg {ms:.3f} ms x {n/3:.0f} = {ms*n/3:.2f} ms per tile")
    sess.close()
