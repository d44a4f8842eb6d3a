#!/bin/bash
# instruction mix of every kernel of a step (separate PMC passes; run on the GPU box from the repo root)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > /tmp/pmc_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
names = ["SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_SMEM","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","GRBM_GUI_ACTIVE"]
print("%-48s" % "kernel (per launch)", " ".join("%12s" % n[-12:] for n in names))
for k in sorted(agg):
    print("%-48s" % k, " ".join("%12.0f" % (agg[k][n] / max(1, cnt[k][n])) for n in names))
PY
