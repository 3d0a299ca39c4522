#!/bin/bash
# 16-float sub-vectors, K <= 32: k_encode_small16 (variant 10) against auto / pair / MFMA kernels
for s in "768 48 16" "128 8 16" "256 16 16" "1024 64 16" "768 48 32" "128 8 32" "512 32 16"; do set -- $s; for v in 10 0 7 4 9; do python bench.py --d $1 --m $2 --k $3 --variant $v --rows 4000000 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip()
if not t: print('d=$1 M=$2 K=$3 variant=$v unsupported'); sys.exit(0)
r=json.loads(t.splitlines()[-1]); ro=r['roofline']; print('d=$1 M=$2 K=$3 variant=$v', ro.get('kernel'), '%.3e vec/s' % r['value'], 'hbm_frac %.3f' % ro.get('hbm_frac', ro['frac']))"; done; done
