#!/bin/bash
# Runs on the GPU box (via gpurun): for every bench workload whose dominant kernel carries a roofline
# record, one `rocprofv3 --kernel-trace --stats` pass and two SEPARATE counter passes (FETCH_SIZE,
# WRITE_SIZE; never combined with other trace domains), then tools/pmc_summarize.py turns them into
# profiles/pmc_traffic.json entries stamped with the kernel-source hash bench.py checks.
# Round 4: every pass runs the DRIVER's command (`--steps 20 --warmup 5`), one run per
# workload, and the summary averages the dominant kernel over the 20 timed launches only (per-dispatch records
# of the kernel trace), next to the HIP-event mean bench.py measured in that very run; for the encode workloads
# a fourth run on the diagnostic build (libpqhip_diag.so, if present) records the in-kernel clock.
# usage: tools/pmc_collect.sh <tag> [workload ...]      (default: the four single-GPU BASELINE configs + lookup + adc_scan)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
WL=${*:-"encode opq_encode reconstruct reconstruct100 encode_d768 lookup lookup100 adc_scan adc_scan8 opq_reconstruct kmeans opq_train opq_train_fast smallk testshape halfdim halfdim_rec"}
for W in $WL; do
  case $W in
    reconstruct100) ARGS="--workload reconstruct --rows 100000000";;
    adc_scan8) ARGS="--workload adc_scan --queries 8";;
    lookup100) ARGS="--workload lookup --lookup-codes 100000000";;
    opq_train_fast) ARGS="--workload opq_train --fast-cross";;
    smallk) ARGS="--workload encode --d 128 --m 16 --k 16";;
    testshape) ARGS="--workload encode --d 20 --m 10 --k 128";;
    halfdim) ARGS="--workload encode --d 300 --m 150 --k 256";;
    halfdim_rec) ARGS="--workload reconstruct --d 300 --m 150 --k 256 --rows 10000000";;
    *) ARGS="--workload $W";;
  esac
  CMD="$R/bench.py $ARGS --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sub-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/stats -- python3 $CMD > $OUT/$W.bench.json 2> $OUT/$W.stats.err || echo "stats pass failed: $W"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$W/fetch -- python3 $CMD > /dev/null 2> $OUT/$W.fetch.err || echo "fetch pass failed: $W"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$W/write -- python3 $CMD > /dev/null 2> $OUT/$W.write.err || echo "write pass failed: $W"
  case $W in
    encode|encode_d768|opq_encode)
      if [ -f $R/reductive_amd/libpqhip_diag.so ]; then
        PQHIP_LIB=$R/reductive_amd/libpqhip_diag.so PQHIP_DEBUG_ENC_STAMP=1 PQHIP_DEBUG_FUSED_STAMP=1 python3 $CMD > /dev/null 2> $OUT/$W.clock.err || echo "clock pass failed: $W"
      fi;;
  esac
  echo "done $W"
done
python3 $R/tools/pmc_summarize.py $OUT $WL > $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json | head -c 3000
