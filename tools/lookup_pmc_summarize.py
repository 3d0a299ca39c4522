#!/usr/bin/env python3
"""Per-launch means of the counters tools/lookup_pmc.sh collected for the lookup kernel (k_reconstruct<.., SEL>), next to
the bench line of the same size.  usage: lookup_pmc_summarize.py <out_dir>"""
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)
out = {}
for n in (10000000, 100000000):
    rec = {}
    try:
        b = json.loads([l for l in open(os.path.join(sys.argv[1], "lookup_%d.bench.json" % n)) if l.startswith("{")][-1])
        rec["bench"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "frac": b["roofline"]["frac"], "kernel": b["roofline"]["kernel"]}
    except Exception as e:
        rec["bench_error"] = str(e)
    counters = {}
    for p in glob.glob(os.path.join(sys.argv[1], "lookup_%d" % n, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_reconstruct" in r["Kernel_Name"]:
                counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    rec["per_launch"] = {k: sum(v[-5:]) / len(v[-5:]) for k, v in sorted(counters.items())}
    c = rec["per_launch"]
    if "TCP_UTCL1_REQUEST_sum" in c and "TCP_UTCL1_TRANSLATION_MISS_sum" in c and c["TCP_UTCL1_REQUEST_sum"]:
        rec["utcl1_miss_rate"] = c["TCP_UTCL1_TRANSLATION_MISS_sum"] / c["TCP_UTCL1_REQUEST_sum"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
        rec["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in c and c.get("TCP_TCC_READ_REQ_sum") and "TCP_TCC_READ_REQ_LATENCY_sum" in c:
        rec["mean_l1_to_l2_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
    out["lookup_codes_%d" % n] = rec
print(json.dumps(out, indent=1))
