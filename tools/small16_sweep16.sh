#!/bin/bash
# 12- to 32-float sub-vectors, K <= 32 (4-bit / 5-bit PQ): k_encode_small16 (variant 10) against auto / pair / MFMA kernels
# usage: tools/small16_sweep16.sh [d:M:K ...]
for s in ${@:-768:48:16 128:8:16 256:16:16 1024:64:16 768:48:32 128:8:32 512:32:16 1024:32:16 256:8:16 768:24:16 1024:32:32 300:15:16 300:25:16 768:32:16 300:15:32 240:12:16}; do IFS=: read d m k <<< "$s"; for v in 10 0 7 4 9; do python bench.py --d $d --m $m --k $k --variant $v --rows 4000000 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read().strip()
if not t: print('d=$d M=$m K=$k variant=$v unsupported'); sys.exit(0)
r=json.loads(t.splitlines()[-1]); ro=r['roofline']; print('d=$d M=$m K=$k variant=$v', ro.get('kernel'), '%.3e vec/s' % r['value'], 'hbm_frac %.3f' % ro.get('hbm_frac', ro['frac']))"; done; done
