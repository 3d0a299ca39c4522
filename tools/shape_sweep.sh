#!/bin/bash
# encode throughput over the PQ configuration space (runs on the GPU box via gpurun)
# usage: tools/shape_sweep.sh  -> one line per shape: d M K rows vec/s ms frac-of-MFMA-peak kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
while read d m k rows; do
  out=$(timeout -k 10 120 python3 $R/bench.py --d $d --m $m --k $k --rows $rows --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null)
  python3 - "$d" "$m" "$k" "$rows" <<PY
import json, sys
r = json.loads('''$out''')
print("d=%s M=%s K=%s rows=%s  %.3e vec/s  %.2f ms  mfma_frac=%.3f  hbm=%.0f GB/s  %s" % (*sys.argv[1:5], r["value"], r["ms_per_step"], r["roofline"].get("mfma_frac", r["roofline"]["frac"]), r["roofline"]["hbm_gbs"], r["encode_kernel"]))
PY
done <<LIST
300 15 256 10000000
300 150 256 4000000
300 75 256 4000000
300 30 256 10000000
300 10 256 10000000
300 20 256 10000000
300 60 256 4000000
300 100 256 4000000
768 48 256 4000000
768 96 256 4000000
768 24 256 4000000
768 16 256 4000000
768 12 256 4000000
300 6 256 4000000
300 15 1024 4000000
128 16 256 10000000
128 8 256 10000000
128 64 16 10000000
300 15 16 10000000
300 15 64 10000000
96 12 256 10000000
LIST
