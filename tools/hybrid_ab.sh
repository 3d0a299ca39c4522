#!/bin/bash
# many-subquantizer shapes: hybrid lane-local + LDS-atomic argmin (libpqhip.so) vs one atomic per distance (libpqhip_timing.so built
# with -DENC_HYBRID_OFF=1), both forced with variant 4, and the VALU-argmin kernel (variant 2); one box
for s in "300 150 256" "300 100 256" "300 75 256" "300 60 256" "300 50 256" "128 16 256" "768 96 256" "20 10 128"; do set -- $s
  for mode in "hybrid:PQHIP_X=1:4" "atomic16:PQHIP_LIB=$PWD/reductive_amd/libpqhip_timing.so:4" "valu:PQHIP_X=1:2" "auto:PQHIP_X=1:0"; do
    IFS=: read name envv var <<< "$mode"
    env $envv python bench.py --d $1 --m $2 --k $3 --variant $var --steps 8 --warmup 2 --no-cpu-baseline --no-sub-configs 2>/dev/null | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('d=$1 M=$2 K=$3 $name', r['encode_kernel'], '%.3e vec/s' % r['value'], 'mfma_frac %.3f' % ro.get('mfma_frac', ro['frac']), 'hbm_frac %.3f' % ro['hbm_frac'])
except Exception as e: print('d=$1 M=$2 K=$3 $name failed')"
  done
done
