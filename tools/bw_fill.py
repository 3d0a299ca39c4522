import torch, time
for gb in (12, 36, 72, 120):
    n = gb * (1 << 30) // 4
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    t.zero_(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): t.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(gb, "GB fill:", gb * (1 << 30) / ms / 1e9, "TB/s")
    del t
