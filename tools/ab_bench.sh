#!/bin/bash
# Same-box A/B of one bench workload under two environments (rule 24: interleaved rounds, one box).
# usage: tools/ab_bench.sh <outdir> <workload> <rounds> "<env A>" "<env B>" [extra bench args]
OUT=$1; WL=$2; N=$3; EA=$4; EB=$5; shift 5
mkdir -p $OUT
for i in $(seq 1 $N); do
  env $EA python bench.py --workload $WL --no-cpu-baseline --no-sub-configs --steps 10 "$@" > $OUT/${WL}_A_$i.json 2> $OUT/${WL}_A_$i.err
  env $EB python bench.py --workload $WL --no-cpu-baseline --no-sub-configs --steps 10 "$@" > $OUT/${WL}_B_$i.json 2> $OUT/${WL}_B_$i.err
done
python - $OUT $WL <<'PY'
import json, sys, glob
out, wl = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob("%s/%s_[AB]_*.json" % (out, wl))):
    try:
        r = json.load(open(f))
        print(f, "%.4g" % r["value"], "frac %.3f" % r["roofline"]["frac"], "avg %.3f ms" % r["roofline"]["avg_launch_ms"], r["roofline"].get("launch_ms", [])[:5])
    except Exception as e:
        print(f, "FAILED", e)
PY
