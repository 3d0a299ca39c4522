#!/bin/bash
# K <= 32 with sub-vectors of 4 / 8 floats: k_encode_small16 (variant 10) against the other kernels that serve these shapes
# (6 = scalar-path VALU kernel, 7 = pair kernel, 4 = MFMA 32x32x2, 9 = MFMA 16x16x4), one bench run each; prints vectors/s
for s in "128 16 16" "128 32 16" "300 75 16" "256 32 16" "768 96 16" "64 8 16" "128 16 32" "128 32 32" "300 75 32" "768 96 32" "128 16 24" "128 16 8"; do set -- $s; for v in 10 6 7 4 9; do python bench.py --d $1 --m $2 --k $3 --variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip()
if not t: print('d=$1 M=$2 K=$3 variant=$v unsupported'); sys.exit(0)
r=json.loads(t.splitlines()[-1]); ro=r['roofline']; print('d=$1 M=$2 K=$3 variant=$v', ro.get('kernel'), '%.3e vec/s' % r['value'], 'hbm_frac %.3f' % ro.get('hbm_frac', ro['frac']))"; done; done
