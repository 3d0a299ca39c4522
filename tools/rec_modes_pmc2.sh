#!/bin/bash
# per-instance (per L2 channel) counters over tools/rec_modes.py: which channels stall in the slow placement?
# usage: tools/rec_modes_pmc2.sh <tag> rows...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL" "TCC_REQ TCC_TAG_STALL" "TCC_HIT TCC_MISS" "TCC_EA0_WRREQ_LEVEL TCC_BUSY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 $R/tools/rec_modes.py "$@" > $OUT/p$i.times 2> $OUT/p$i.err || echo "pass $i failed"
  echo "pass $i done"
done
