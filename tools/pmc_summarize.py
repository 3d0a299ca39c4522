#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/pmc_collect.sh into the profiles/pmc_traffic.json format that
bench.py reads: per workload the dominant kernel (longest total time in the stats pass), its average
duration, and HBM bytes per launch = FETCH_SIZE x 1024 x 2 (gfx950: FETCH_SIZE tallies 128-B requests
at 64 B, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE x 1024, each the mean over the kernel's dispatches of
its own pass.  Every entry carries the sha256 prefix of the kernel sources it was measured on; bench.py
refuses an entry whose hash differs from the build it runs.
usage: pmc_summarize.py <out_dir> <workload> ..."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
csv.field_size_limit(1 << 30)

def stats(d):
    best = None
    for p in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "pqhip::" not in r["Name"]:
                continue
            if best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"]):
                best = r
    return best

def timed_launches(d, kernel, n_timed):
    """durations (ns) of the LAST n_timed dispatches of `kernel` in the kernel trace of the stats pass: the timed
    steps of the bench run, without the warm-up launches (the first of which is cold)"""
    rows = []
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Kernel_Name"] == kernel:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    return [dur for _, dur in rows][-n_timed:]


def clock_mhz(path):
    """mean of the 'clock NNNN MHz' figures the diagnostic build printed (in-kernel s_memtime / s_memrealtime)"""
    import re
    try:
        v = [float(m) for m in re.findall(r"clock (\d+) MHz", open(path).read())]
    except OSError:
        return None
    v = [x for x in v if x > 0]
    return sum(v[-20:]) / len(v[-20:]) if v else None


def counter(d, name, kernel):
    vals = []
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name and r["Kernel_Name"] == kernel:
                vals.append(float(r["Counter_Value"]))
    return vals

def all_kernels(d):
    """name -> (calls, average ns) of every pqhip kernel of the stats pass"""
    out = {}
    for p in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "pqhip::" in r["Name"]:
                out[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    return out

def step_total(d, name, kernels, n_steps):
    """sum of a counter over every dispatch of the per-step kernels, per bench step (warm-up steps run the same
    launches, so the pass total divides by warm-up + timed steps)"""
    tot = 0.0
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name and r["Kernel_Name"] in kernels:
                tot += float(r["Counter_Value"])
    return tot / n_steps

out_dir, wls = sys.argv[1], sys.argv[2:]
entries = {}
for w in wls:
    try:
        b = json.loads([l for l in open(os.path.join(out_dir, w + ".bench.json")) if l.startswith("{")][-1])
    except Exception as e:
        print("no bench line for", w, e, file=sys.stderr); continue
    s = stats(os.path.join(out_dir, w, "stats"))
    if s is None:
        continue
    kern = s["Name"]
    f = counter(os.path.join(out_dir, w, "fetch"), "FETCH_SIZE", kern)
    wr = counter(os.path.join(out_dir, w, "write"), "WRITE_SIZE", kern)
    if not f or not wr:
        print("no counters for", w, file=sys.stderr); continue
    # skip the warm-up dispatches: keep the timed steps
    per_step_launches = max(1, int(s["Calls"]) // (b["steps"] + b["warmup"]))
    n_timed = b["steps"] * per_step_launches
    f, wr = f[-n_timed:], wr[-n_timed:]
    tl = timed_launches(os.path.join(out_dir, w, "stats"), kern, n_timed)
    c = b["config"]
    fetch_b = sum(f) / len(f) * 1024 * 2
    write_b = sum(wr) / len(wr) * 1024
    wl_name = {"reconstruct100": "reconstruct", "adc_scan8": "adc_scan", "smallk": "encode", "testshape": "encode", "halfdim": "encode", "halfdim_rec": "reconstruct", "lookup100": "lookup",
               "opq_train_fast": "opq_train"}.get(w, w)
    key = "%s@%d@d%d_m%d_k%d" % (wl_name, c["rows_per_gpu"], c["d"], c["M"], c["K"])
    if b.get("queries_per_scan", 1) > 1:
        key += "_q%d" % b["queries_per_scan"]
    if w == "lookup100":
        key += "_codes100000000"
    if w == "opq_train_fast":
        key += "_fastcross"
    alg = b["roofline"].get("algorithmic_bytes_per_vector", 0) * c["rows_per_gpu"]
    entries[key] = {"workload": wl_name, "rows": c["rows_per_gpu"], "kernel": kern[:120], "calls_in_stats_pass": int(s["Calls"]),
                    "avg_launch_ms_stats_pass": float(s["AverageNs"]) / 1e6,
                    "avg_launch_ms_timed_launches": (sum(tl) / len(tl) / 1e6) if tl else None,
                    "min_max_launch_ms_timed_launches": [min(tl) / 1e6, max(tl) / 1e6] if tl else None,
                    "timed_launches": len(tl),
                    "bench_avg_launch_ms_hip_events": b["roofline"]["avg_launch_ms"],
                    "bench_ms_per_step_same_run": b["ms_per_step"],
                    "in_kernel_clock_mhz_diag_pass": clock_mhz(os.path.join(out_dir, w + ".clock.err")),
                    "fetch_bytes": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
                    "algorithmic_bytes": alg, "ratio": (fetch_b + write_b) / alg if alg else None,
                    "dispatches_averaged": len(f), "source_hash": bench.source_hash()}
    # a step that is several launches (rotation + encode per chunk): the record a bench line needs is the whole
    # step's traffic, so sum every kernel that runs a whole number of times per step
    n_steps = b["steps"] + b["warmup"]
    ks = {k: v for k, v in all_kernels(os.path.join(out_dir, w, "stats")).items() if v[0] >= n_steps and v[0] % n_steps == 0}
    if len(ks) > 1 or int(s["Calls"]) != n_steps:
        sf = step_total(os.path.join(out_dir, w, "fetch"), "FETCH_SIZE", ks, n_steps) * 1024 * 2
        sw = step_total(os.path.join(out_dir, w, "write"), "WRITE_SIZE", ks, n_steps) * 1024
        e = entries[key]
        e["dominant_kernel_bytes_per_launch"] = e["hbm_bytes_per_launch"]
        e["launches_per_step"] = {k[:80]: v[0] // n_steps for k, v in ks.items()}
        e["kernel_ms_per_step"] = sum(v[0] // n_steps * v[1] for v in ks.values()) / 1e6
        e["fetch_bytes"], e["write_bytes"], e["hbm_bytes_per_launch"] = sf, sw, sf + sw
        e["ratio"] = (sf + sw) / alg if alg else None
        e["note"] = "one bench step = several launches; bytes are the sum over all of them"
print(json.dumps({"_comment": "HBM bytes per launch of each workload's dominant kernel from rocprofv3 --pmc passes (tools/pmc_collect.sh; FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE x1024 x2 per the gfx950 correction in MI355X_MICROARCH.md 'HBM', WRITE_SIZE x1024). bench.py reports an entry only when workload, size and source_hash match the build it runs.",
                  "entries": entries}, indent=1))
