#!/bin/bash
# usage: pmc_quick.sh <tag> "<python command>" <kernel substring>  -- two SQ passes, averages per dispatch of the matching kernel
cd /tmp && export TMPDIR=/tmp
TAG=$1; CMD=$2; KERN=$3
OUT=/root/repo/gpurun_out/pmcq_$TAG
mkdir -p $OUT
i=0
# counter groups, one rocprofv3 pass each: BM_PMC_SETS="group 1;group 2" overrides the two SQ groups
DEFAULT_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
IFS=';' read -ra SETS <<< "${BM_PMC_SETS:-$DEFAULT_SETS}"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KERN" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$TAG $KERN:", " ".join("%s=%.4g" % (c, tot[c] / max(n[c], 1)) for c in sorted(tot)), "dispatches", max(n.values()) if n else 0)
PY
