#!/bin/bash
# usage: pmc.sh <tag>   -- separate --pmc passes (no tracing domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
TAG=$1
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE TCC_HIT TCC_MISS TCC_REQ" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_paths<false," not in k: continue
        tot[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
disp = 4
for c in sorted(tot): print(c, tot[c]["v"] / disp)
import json
json.dump({"workload": "config2", "per": "launch of bm::trace_paths<false>", **{("%s_KiB" % c if c in ("FETCH_SIZE", "WRITE_SIZE") else c): tot[c]["v"] / disp for c in sorted(tot)}}, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
