#!/bin/bash
# rocprofv3 kernel statistics of tools/variant_bench.py with one measurement build: tools/variant_stats.sh variants/x.so [filter]
R=$(pwd)
L=$(realpath $1)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o v -- python $R/tools/variant_bench.py --lib $L --steps 40 > /tmp/prof_v.log 2>&1
cd $R
f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1)
python - "$f" "${2:-}" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-60s %6s %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
