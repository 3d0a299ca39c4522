import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import synth, reductive_amd
for (d, M, K) in ((128, 16, 16), (300, 15, 16), (128, 16, 32)):
    q = synth.normalish(43, (M, K, d // M))
    for variant in (6, 4):
        pq = reductive_amd.Pq(None, q); pq.set_encode_variant(variant)
        for rows in (65536, 10_000_000):
            x = torch.randn((rows, d), device='cuda'); out = torch.empty((rows, M), device='cuda', dtype=torch.uint8)
            reps = max(3, 400_000_000 // rows // 8)
            pq.quantize_batch_device(x, out=out); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps): pq.quantize_batch_device(x, out=out)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            print(d, M, K, 'variant', variant, 'rows', rows, '%.3e vec/s' % (rows / ms * 1e3), pq.last_encode_kernel())
