import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, synth, reductive_amd
M, K, dsub = 15, 256, 20
q = synth.normalish(43, (M, K, dsub))
pq = reductive_amd.Pq(None, q)
qt = torch.from_numpy(q).cuda()
for n in (15_000_000, 17_000_000, 20_000_000):
    g = torch.Generator(device="cuda").manual_seed(42)
    codes = torch.randint(0, K, (n, M), device="cuda", dtype=torch.uint8, generator=g)
    out = torch.empty((n, M * dsub), device="cuda", dtype=torch.float32)
    pq.reconstruct_batch_device(codes, out=out); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): pq.reconstruct_batch_device(codes, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # correctness on the tail rows (beyond 2^24) and head rows
    idx = torch.cat([torch.arange(0, 4096), torch.arange(n - 4096, n)]).cuda()
    ref = torch.cat([qt[m][codes[idx, m].long()] for m in range(M)], dim=1)
    ok = torch.equal(ref, out[idx])
    hist = torch.bincount(codes[-1000000:].flatten().int(), minlength=256)
    print(n, "ms", ms, "TB/s", n * 1215 / ms / 1e9, "correct", ok, "hist min/max of last 1M rows", int(hist.min()), int(hist.max()))
    del codes, out
