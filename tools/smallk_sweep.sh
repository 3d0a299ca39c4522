#!/bin/bash
# small-codebook shapes: auto dispatch against the forced kernels (2 = MFMA + VALU argmin, 4 = MFMA + LDS argmin,
# 6 = VALU kernel with scalar-path centroids), one bench run each
for s in "128 16 16" "128 16 32" "128 16 64" "300 15 16" "300 15 32" "300 15 64" "768 48 16" "768 48 64"; do set -- $s; for v in 0 2 4 6; do python bench.py --d $1 --m $2 --k $3 --variant $v --steps 5 --warmup 1 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('d=$1 M=$2 K=$3 variant=$v', r['encode_kernel'], '%.3e vec/s' % r['value'], 'hbm_frac %.3f mfma_frac %.3f' % (ro['hbm_frac'], ro.get('mfma_frac', ro['frac'])))"; done; done
