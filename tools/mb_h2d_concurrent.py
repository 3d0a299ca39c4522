#!/usr/bin/env python3
"""Host->device copies of 256 MB pinned buffers: one stream alone against two / four streams at once on ONE GPU (the
situation of `bench.py --in-process 2 --single-device`, where two device slots share one PCIe link).
usage: python tools/mb_h2d_concurrent.py"""
import threading, time, torch
N = 256 << 20
def run(k, reps=12):
    host = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(k)]
    dev = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def work(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                dev[i].copy_(host[i], non_blocking=True)
            streams[i].synchronize()
    for i in range(k):
        work(i)                                   # warm
    torch.cuda.synchronize()
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%d stream(s): %.1f GB/s aggregate" % (k, k * reps * N / dt / 1e9), flush=True)
for k in (1, 2, 4, 1):
    run(k)
