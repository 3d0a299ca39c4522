"""reconstruct_batch over sub-vector shapes: ms per 10 M codes rows and fraction of the HBM peak (8 TB/s) on (codes + rows written)."""
import sys
import numpy as np, torch
import reductive_amd as ra

n = 10_000_000
rng = np.random.default_rng(1)
SHAPES = [(300, 15, 256), (320, 160, 256), (300, 150, 256), (300, 75, 256), (300, 30, 256), (300, 300, 256), (128, 16, 256), (128, 64, 256),
                (128, 128, 256), (768, 48, 256), (768, 96, 256), (20, 10, 128), (128, 16, 16), (64, 32, 128), (96, 12, 256)]
if len(sys.argv) > 3:
    SHAPES = [tuple(int(v) for v in sys.argv[1:4])]      # one shape: python tools/rec_shapes.py <d> <M> <K>
for d, M, K in SHAPES:
    rows = n if d <= 300 else 4_000_000
    pq = ra.Pq(None, rng.standard_normal((M, K, d // M), dtype=np.float32))
    codes = torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8)
    out = torch.empty((rows, d), device="cuda", dtype=torch.float32)
    for _ in range(2):
        pq.reconstruct_batch_device(codes, out=out, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pq.reconstruct_batch_device(codes, out=out, check=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("d=%d M=%d K=%d rows=%d  %.3f ms  %.3e rows/s  %.3f of HBM  %s" % (d, M, K, rows, ms, rows / ms * 1e3, rows * (4 * d + M) / ms * 1e3 / 8e12, ra.launch_log()[-1:]), flush=True)
    del codes, out, pq
    torch.cuda.empty_cache()
