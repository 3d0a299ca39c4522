#!/bin/bash
# small codebooks (K <= 16): pair kernel (variant 7, the auto choice) vs the VALU kernel (6) vs the default MFMA kernel (4), one box
for s in "128 16 16" "128 8 16" "128 32 16" "128 64 16" "768 48 16" "300 75 16"; do set -- $s
  for var in 7 6 4; do
    python bench.py --d $1 --m $2 --k $3 --variant $var --steps 10 --warmup 2 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('d=$1 M=$2 K=$3 variant=$var', r['encode_kernel'], '%.3e vec/s' % r['value'], 'hbm_frac %.3f' % ro['hbm_frac'], 'ms %.3f' % ro['avg_launch_ms'])
except Exception as e: print('d=$1 M=$2 K=$3 variant=$var failed', e)"
  done
done
