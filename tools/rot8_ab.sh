#!/bin/bash
# same-box A/B of rotation-kernel variants inside the OPQ encode bench, with in-kernel stamps and rocprof kernel stats
OUT=gpurun_out/$1; mkdir -p $OUT
for v in "PQHIP_X=1" "PQHIP_DEBUG_ROT8_STATIC=1" "PQHIP_DEBUG_NO_GEMM8=1"; do
  echo "== $v"
  env $v PQHIP_DEBUG_ROT_STAMP=1 python bench.py --workload opq_encode --no-cpu-baseline --no-sub-configs --steps 1 --warmup 1 2>&1 | grep stamps | sed -n 11,12p
done
for i in 1 2; do
for v in "PQHIP_X=1" "PQHIP_DEBUG_ROT8_STATIC=1" "PQHIP_DEBUG_NO_GEMM8=1"; do
  env $v tools/prof_stats.sh $1/prof_${v}_$i --workload opq_encode --no-sub-configs --steps 4 --warmup 1 > /dev/null 2>&1
  python3 - $OUT/prof_${v}_$i "$v" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
r = json.load(open(sys.argv[1] + "/bench.json"))
out = [sys.argv[2], "step %.2f ms" % r["roofline"]["avg_launch_ms"]]
for x in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:2]:
    out.append("%s avg %.4f max %.4f" % (x["Name"][12:32], float(x["AverageNs"]) / 1e6, float(x["MaxNs"]) / 1e6))
print("  ".join(out))
PY
done; done
