#!/bin/bash
# SQ counters of the inflate kernels (two PMC passes; run on the GPU box from the repo root): where a step's cycles go
# usage: tools/pmc_inflate.sh [mbp] [wave|lanes]
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmci_$i
  BDX_KZ=${2:-wave} timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmci_$i -o p -- python $R/tools/bamdec_probe.py --mbp ${1:-4} --inflate-only > /tmp/pmci_$i.log 2>&1 < /dev/null
done
cd $R
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/pmci_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kz_" not in r["Kernel_Name"]: continue
        name = re.search(r"kz_\w+", r["Kernel_Name"]).group(0)
        agg[(name, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
print("inflate kernels, mean per launch (SQ cycle counters are quad-cycles summed over waves / SIMDs):")
for k in sorted(agg): print("  %-28s %-24s %16.0f  (%d launches)" % (k[0], k[1], agg[k] / cnt[k], cnt[k]))
PY
tail -3 /tmp/pmci_1.log
