#!/bin/bash
# rocprofv3 kernel statistics of the default bench run -> gpurun_out/kstats.csv (run on the GPU box from the repo root)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python $R/bench.py --steps ${1:-40} --warmup 5 --no-cpu-baseline --no-end-to-end > /tmp/prof_ks.log 2>&1
cd $R
mkdir -p gpurun_out
f=$(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/kstats.csv
python - <<PY
import csv
tot=0
for r in csv.DictReader(open("gpurun_out/kstats.csv")):
    print("%-70s %6s %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
PY
