#!/bin/bash
# same-box A/B: the 16x16x4 encode kernel (default) vs k_encode_mfma_lds3 (PQHIP_DEBUG_NO_MFMA16=1)
tag=${1:-m16ab}; out=gpurun_out/$tag; mkdir -p $out
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  ms', round(r['ms_per_step'],3), 'value %.4g' % r['value'], 'frac', round(r['roofline']['frac'],4), r.get('encode_kernel'))"; }
for round in 1 2; do
  for wl in "encode" "encode_d768 --rows 12500000" "kmeans" "opq_encode"; do
    for env in "PQHIP_X=1" "PQHIP_DEBUG_NO_MFMA16=1"; do
      echo "== $wl $env round $round" | tee -a $out/log.txt
      env $env python bench.py --workload $wl --no-cpu-baseline --no-sub-configs --steps 10 --warmup 3 2>/dev/null | line | tee -a $out/log.txt
    done
  done
done
for env in "PQHIP_X=1" "PQHIP_DEBUG_NO_MFMA16=1"; do
  env $env PQHIP_DEBUG_ENC_STAMP=1 python bench.py --workload encode --no-cpu-baseline --no-sub-configs --steps 3 --warmup 2 2>&1 | grep "encode stamps" | tail -2 | tee -a $out/log.txt
done
