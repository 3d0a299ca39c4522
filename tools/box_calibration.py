#!/usr/bin/env python3
"""Per-box calibration for the 100 M-code reconstruct (the 4.7 <-> 6.05 TB/s "box to
box" swing).  On ONE box, in ONE process, back to back:
  * torch fill_ (pure streaming store) and copy_ (load + store) rates on a 120 GB / 60 GB buffer,
  * k_reconstruct on 10 M and 100 M code rows: every launch's duration (min / median / max),
  * sclk / mclk / fclk / power sampled through rocm-smi while the 100 M launches run.
One JSON line: if the box's plain store rate moves with the reconstruct rate, the swing is the box
(HBM / fabric clocks of that device), not the kernel; a within-box spread would be the kernel's.
usage: python tools/box_calibration.py > gpurun_out/<tag>/box.json"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, reductive_amd

out = {"device": torch.cuda.get_device_name(0)}
def timed(fn, reps):
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    return ms
n = 120 * (1 << 30) // 4
t = torch.empty(n, dtype=torch.float32, device="cuda"); t.zero_(); torch.cuda.synchronize()
ms = timed(lambda: t.fill_(1.0), 5)
out["fill_120GB_TBps"] = [round(n * 4 / m / 1e9, 3) for m in ms]
h = n // 2
ms = timed(lambda: t[:h].copy_(t[h:]), 5)
out["copy_60GB_TBps_rw"] = [round(2 * h * 4 / m / 1e9, 3) for m in ms]
del t; torch.cuda.empty_cache()

M, K, dsub = 15, 256, 20
pq = reductive_amd.Pq(None, synth.normalish(43, (M, K, dsub)))
smi = []
def sampler(stop):
    while not stop.is_set():
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            smi.append(" ".join(l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power"))))
        except Exception as e:
            smi.append("err %s" % e)
        time.sleep(0.3)
for rows in (10_000_000, 100_000_000):
    g = torch.Generator(device="cuda").manual_seed(42)
    codes = torch.randint(0, K, (rows, M), device="cuda", dtype=torch.uint8, generator=g)
    dst = torch.empty((rows, M * dsub), device="cuda", dtype=torch.float32)
    pq.reconstruct_batch_device(codes, out=dst, check=False); torch.cuda.synchronize()
    stop = threading.Event()
    th = threading.Thread(target=sampler, args=(stop,)) if rows == 100_000_000 else None
    if th: th.start()
    ms = timed(lambda: pq.reconstruct_batch_device(codes, out=dst, check=False), 30)
    if th: stop.set(); th.join()
    tb = sorted(rows * 1215 / m / 1e9 for m in ms)
    out["reconstruct_%dM_TBps" % (rows // 1_000_000)] = {"min": round(tb[0], 3), "median": round(tb[len(tb) // 2], 3), "max": round(tb[-1], 3),
                                                         "launch_ms": [round(m, 3) for m in ms]}
    del codes, dst; torch.cuda.empty_cache()
out["smi_during_100M"] = smi[:12]
print(json.dumps(out))
