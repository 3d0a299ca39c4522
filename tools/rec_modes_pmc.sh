#!/bin/bash
# counter passes (one small group per pass, --kernel-trace only) over tools/rec_modes.py
# usage: tools/rec_modes_pmc.sh <tag> rows...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum" \
         "TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_LEVEL_sum" \
         "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
         "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 $R/tools/rec_modes.py "$@" > $OUT/p$i.times 2> $OUT/p$i.err || echo "pass $i failed"
  echo "pass $i done"
done
