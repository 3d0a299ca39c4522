import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, synth, reductive_amd
q = synth.normalish(43, (15, 256, 20))
pq = reductive_amd.Pq(None, q)
x = np.random.default_rng(1).standard_normal((4_000_000, 300), dtype=np.float32)
pq.quantize_batch(x[:100000])
for n in (1_000_000, 4_000_000):
    t = time.perf_counter(); c = pq.quantize_batch(x[:n]); dt = time.perf_counter() - t
    print("host-resident encode", n, "rows:", n / dt, "vec/s", n * 1200 / dt / 1e9, "GB/s")
t = time.perf_counter(); r = pq.reconstruct_batch(c[:2_000_000]); dt = time.perf_counter() - t
print("host-resident reconstruct 2M rows:", 2e6 / dt, "vec/s")
