#!/usr/bin/env python3
"""Merges the pmc_traffic.json of several tools/pmc_collect.sh runs (gpurun_out/<tag>/) into profiles/pmc_traffic.json and copies
the newest rocprofv3 kernel-stats CSV of every workload to profiles/<prefix>_<workload>_kernel_stats.csv.
usage: tools/pmc_merge.py <prefix> <tag> [<tag> ...]"""
import glob, json, os, shutil, sys
prefix, tags = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
merged = None
for t in tags:
    j = json.load(open(os.path.join(root, "gpurun_out", t, "pmc_traffic.json")))
    if merged is None:
        merged = j
    else:
        merged["entries"].update(j["entries"])
    for w in sorted(set(p.split(os.sep)[-4] for p in glob.glob(os.path.join(root, "gpurun_out", t, "*", "stats", "*", "*_kernel_stats.csv")))):
        newest = max(glob.glob(os.path.join(root, "gpurun_out", t, w, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        shutil.copy(newest, os.path.join(root, "profiles", "%s_%s_kernel_stats.csv" % (prefix, w)))
json.dump(merged, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(len(merged["entries"]), "entries;", sorted(set(e["source_hash"] for e in merged["entries"].values())))
