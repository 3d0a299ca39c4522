#!/usr/bin/env python3
"""print average duration (ms) of kernels whose name contains the given substring, from a
rocprofv3 --stats output directory"""
import csv, glob, sys
pat, d = sys.argv[1], sys.argv[2]
for p in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if pat in r["Name"]:
            print("%-50s calls=%s avg_ms=%.3f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
