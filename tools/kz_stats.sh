#!/bin/bash
# rocprofv3 kernel statistics of the inflate probe (tools/bamdec_probe.py --inflate-only) -> gpurun_out/kz_stats.txt
# usage (GPU box, repo root): tools/kz_stats.sh [mbp] [wave|lanes] [slice-gb]
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kz
BDX_KZ=${2:-lanes} timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kz -o kz -- python $R/tools/bamdec_probe.py --mbp ${1:-20} --inflate-only --slice-gb ${3:-1.5} > /tmp/prof_kz.log 2>&1
cd $R
mkdir -p gpurun_out
f=$(find /tmp/prof_kz -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -5 /tmp/prof_kz.log; exit 1; fi
{ grep "inflate kernel\|members" /tmp/prof_kz.log; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s  average %12.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
} | tee gpurun_out/kz_stats_${2:-lanes}_${1:-20}.txt
