import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import reductive_amd as ra
from reductive_amd.pq import kmeans_iterations
M, K, DSUB = 15, 256, 20
for rows in (16384, 65536):
    g = torch.Generator(device="cuda").manual_seed(42)
    src = torch.randn((rows, M * DSUB), device="cuda", generator=g)
    pick = torch.arange(K, device="cuda") * (rows // K)
    q0 = np.stack([src[(pick + 7 * m) % rows, m * DSUB:(m + 1) * DSUB].cpu().numpy() for m in range(M)])
    for it in (0, 1, 5, 20):
        q = q0 if it == 0 else kmeans_iterations(q0, src, n_iterations=it, want_loss=False)[0]
        pq = ra.Pq(None, q)
        codes = pq.quantize_batch_device(src)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            pq.quantize_batch_device(src, out=codes)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20
        cnt = np.stack([np.bincount(codes[:, m].cpu().numpy(), minlength=K) for m in range(M)])
        # rows that coincide exactly with their centroid
        rec = pq.reconstruct_batch_device(codes)
        same = int((rec == src).reshape(rows, M, DSUB).all(-1).sum())
        print("rows", rows, "after", it, "iters: encode %.3f ms" % (dt * 1e3), "empty", int((cnt == 0).sum()), "singletons", int((cnt == 1).sum()), "rows equal to their centroid", same, flush=True)
