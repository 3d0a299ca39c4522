#!/bin/bash
# encode-kernel clock and cycles (in-kernel stamps) when it follows rotation v8 / v6 in the OPQ chunk loop
for v in "PQHIP_X=1" "PQHIP_DEBUG_NO_GEMM8=1" "PQHIP_X=1" "PQHIP_DEBUG_NO_GEMM8=1"; do
  echo "== $v"
  env $v PQHIP_DEBUG_ENC_STAMP=1 python bench.py --workload opq_encode --no-cpu-baseline --no-sub-configs --steps 1 --warmup 1 2>&1 | grep "encode stamps" | sed -n 10,13p
done
