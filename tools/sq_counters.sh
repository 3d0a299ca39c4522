#!/bin/bash
# Runs on the GPU box: SQ counters of one kernel under tools/sk_time.py (or any command after the tag), a few counters per pass.
# usage: [KFILTER=k_reconstruct] tools/sq_counters.sh <tag> <python script and args ...>   (KFILTER: substring of the kernel names kept, default k_encode)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$R
KF=${KFILTER:-k_encode}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/"$@" > $OUT/stats.out 2> $OUT/stats.err || echo "stats pass failed"
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -- python3 $R/"$@" > /dev/null 2> $OUT/pmc$i.err || echo "pmc pass $i failed ($SET)"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "$KF" not in k: continue
        key = (k.split("(")[0][:60], r["Counter_Name"])
        tot[key][0] += float(r["Counter_Value"]); tot[key][1] += 1
for (k, c), (v, n) in sorted(tot.items()):
    print("%-60s %-32s %14.0f per launch (%d launches)" % (k, c, v / n, n))
PY
grep -h "$KF" $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -5
