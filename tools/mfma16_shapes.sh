#!/bin/bash
# where does k_encode_mfma16 beat k_encode_mfma_lds3?  one line per (shape, kernel), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
while read d m k rows; do
  for env in "PQHIP_X=1" "PQHIP_DEBUG_NO_MFMA16=1"; do
  out=$(env $env timeout -k 10 120 python3 $R/bench.py --d $d --m $m --k $k --rows $rows --steps 5 --warmup 2 --no-cpu-baseline --variant ${VARIANT:-0} 2>/dev/null)
  python3 - "$d" "$m" "$k" "$rows" <<PY
import json, sys
r = json.loads('''$out''')
print("d=%s M=%s K=%s rows=%s  %.3e vec/s  %.2f ms  mfma_frac=%.3f  %s" % (*sys.argv[1:5], r["value"], r["ms_per_step"], r["roofline"].get("mfma_frac", r["roofline"]["frac"]), r["encode_kernel"]))
PY
  done
done <<LIST
288 24 256 10000000
288 12 256 10000000
336 12 256 10000000
300 15 128 10000000
768 48 128 4000000
256 8 256 10000000
288 24 128 10000000
336 12 128 10000000
LIST
