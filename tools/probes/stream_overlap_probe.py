"""Do kernels of different HIP streams overlap on this box?  torch.cuda._sleep (one spinning thread) on K streams, with and without pinned H2D copies
between them; wall time serial vs concurrent.  usage: python tools/probes/stream_overlap_probe.py"""
import time
import torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
CYC = 2_000_000                        # ~1 ms of spinning


def run(k, n, copy_mb=0, blocking_src=False):
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
    src = [torch.empty(copy_mb << 20, dtype=torch.uint8).pin_memory() if copy_mb else None for _ in range(k)]
    dst = [torch.empty(copy_mb << 20, dtype=torch.uint8, device=dev) if copy_mb else None for _ in range(k)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for s in range(k):
            with torch.cuda.stream(streams[s]):
                if copy_mb:
                    dst[s].copy_(src[s], non_blocking=True)
                torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


run(1, 3)
one = run(1, 20)
print(f"1 stream x 20 sleeps: {one:.1f} ms")
for k in (2, 3, 4, 6):
    print(f"{k} streams x 20 sleeps each: {run(k, 20):.1f} ms (serial would be {one * k:.1f})")
for mb in (36, 72):
    one_c = run(1, 20, mb)
    print(f"1 stream x 20 x (H2D {mb} MB pinned + sleep): {one_c:.1f} ms")
    print(f"3 streams x 20 x (H2D {mb} MB pinned + sleep): {run(3, 20, mb):.1f} ms (serial would be {3 * one_c:.1f})")
