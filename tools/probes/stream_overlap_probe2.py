"""H2D copies vs kernels on other streams: who blocks whom?  usage: python tools/probes/stream_overlap_probe2.py"""
import os
import time
import torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
CYC = 2_000_000
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"), " GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
MB = 72
src = [torch.empty(MB << 20, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(MB << 20, dtype=torch.uint8, device=dev) for _ in range(4)]
cs = torch.cuda.Stream(device=dev)
ks = [torch.cuda.Stream(device=dev) for _ in range(3)]


def t(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    h = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return h, (time.perf_counter() - t0) * 1e3


def copies(n=20):
    with torch.cuda.stream(cs):
        for i in range(n):
            dst[0].copy_(src[0], non_blocking=True)


def sleeps(n=20):
    for i in range(n):
        for s in ks:
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)


t(copies); t(sleeps)
print("20 copies of %d MB on one stream: host %.1f ms, total %.1f ms" % ((MB,) + t(copies)))
print("3 x 20 sleeps on three streams:   host %.1f ms, total %.1f ms" % t(sleeps))
print("both, copies enqueued first:      host %.1f ms, total %.1f ms" % t(lambda: (copies(), sleeps())))
print("both, sleeps enqueued first:      host %.1f ms, total %.1f ms" % t(lambda: (sleeps(), copies())))


def inter():
    for i in range(20):
        with torch.cuda.stream(cs):
            dst[0].copy_(src[0], non_blocking=True)
        for s in ks:
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)


print("interleaved enqueue (copy stream independent of the sleep streams): host %.1f ms, total %.1f ms" % t(inter))


def dep():
    for i in range(20):
        for j, s in enumerate(ks):
            with torch.cuda.stream(cs):
                dst[j].copy_(src[j], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(cs)
            with torch.cuda.stream(s):
                s.wait_event(ev)
                torch.cuda._sleep(CYC)


print("copy stream + event -> sleep stream (3 x 20): host %.1f ms, total %.1f ms" % t(dep))


def same():
    for i in range(20):
        for j, s in enumerate(ks):
            with torch.cuda.stream(s):
                dst[j].copy_(src[j], non_blocking=True)
                torch.cuda._sleep(CYC)


print("copy in the sleep's own stream (3 x 20):     host %.1f ms, total %.1f ms" % t(same))
# a device-side gather kernel reading the pinned buffer directly (zero-copy over PCIe) instead of hipMemcpyAsync
hsrc = [torch.empty(MB << 20, dtype=torch.uint8).pin_memory() for _ in range(3)]


def zc():
    for i in range(20):
        for j, s in enumerate(ks):
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)


print("reference: sleeps only (3 x 20): host %.1f ms, total %.1f ms" % t(zc))
