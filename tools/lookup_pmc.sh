#!/bin/bash
# Counters behind the lookup workload's roofline fraction ("TLB reach / DRAM page locality is the suspect"
# was a suspicion, no counter).  One rocprofv3 --pmc pass per counter group (never combined with other trace domains), the
# lookup of 10 M random rows out of a resident code matrix of 10 M and of 100 M rows.  Summary: tools/lookup_pmc_summarize.py.
# usage (GPU box): tools/lookup_pmc.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for N in 10000000 100000000; do
  CMD="$R/bench.py --workload lookup --lookup-codes $N --steps 5 --warmup 2 --no-cpu-baseline --no-sub-configs"
  python3 $CMD > $OUT/lookup_$N.bench.json 2> $OUT/lookup_$N.err
  i=0
  for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "GRBM_GUI_ACTIVE SQ_WAVES_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/lookup_$N/p$i -- python3 $CMD > /dev/null 2> $OUT/lookup_$N.p$i.err || echo "pass $i failed ($C)"
  done
  echo "done $N"
done
python3 $R/tools/lookup_pmc_summarize.py $OUT > $OUT/lookup_pmc.json
cat $OUT/lookup_pmc.json
