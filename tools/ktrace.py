#!/usr/bin/env python3
"""print the duration (ms) of every launch of kernels matching a substring, in launch order,
from a rocprofv3 kernel_trace.csv"""
import csv, sys
pat, path = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
