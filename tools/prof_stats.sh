#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats only, for any bench arguments.
# usage: tools/prof_stats.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
D=$(mktemp -d /tmp/prof_XXXXXX)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/stats.err
find $D -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $D -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
cat $OUT/bench.json
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-60s calls=%-5s avg_ms=%9.4f  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
