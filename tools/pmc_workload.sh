#!/bin/bash
# usage: pmc_workload.sh <workload> <tag>  -- FETCH/WRITE/TCC counters of one bench workload (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
WL=$1; TAG=$2
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python /root/repo/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_paths<false>" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(tot): print(c, tot[c] / max(n[c], 1) * (1 if True else 1), "(per dispatch; %d dispatches)" % n[c])
PY
