"""plain rotation x.P on the GPU: k_rotate_pblock8 (32x32x2) vs k_rotate_pblock9 (16x16x4) over d (runs on the GPU box)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import reductive_amd as ra
n = 2_000_000
for d in (48, 80, 96, 100, 128, 144, 160, 208, 256, 272, 300, 336, 400, 512, 600):
    x = torch.randn((n, d), device="cuda")
    P = np.linalg.qr(np.random.RandomState(d).randn(d, d))[0].astype(np.float32)
    res = []
    for v in (8, 9):
        ra.set_rotation_variant(v)
        for _ in range(2): ra.rotate(x, P)
        best = 1e9
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ra.rotate(x, P)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
        res.append(best)
    ra.set_rotation_variant(0)
    print("d=%d  v8 %.3f ms  v9 %.3f ms  v9/v8 %.3f  (%.1f / %.1f TFLOP/s)" % (d, res[0], res[1], res[1] / res[0], 2 * d * d * n / res[0] / 1e9, 2 * d * d * n / res[1] / 1e9))
