#!/bin/bash
# idle time on the stream before every kernel of ONE context's run at a GPU's share of a genome (rocprofv3 kernel trace around
# tools/genome_ab.py; run on the GPU box from the repo root).  usage: gap_trace_genome.sh [--t]
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gapg
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gapg -o g -- python $R/tools/genome_ab.py --rounds 2 $1 > /tmp/prof_gapg.log 2>&1
cd $R
grep -E "best" /tmp/prof_gapg.log
f=$(find /tmp/prof_gapg -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if g < 500: gaps[n1.replace("bdx::", "")[:44]].append(g)   # (the pause between two runs is not a gap of the run)
print("%-46s %6s %9s %9s" % ("kernel (gap = its start - previous kernel's end)", "n", "median us", "p90 us"))
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
    v = sorted(v)
    tot += v[len(v) // 2]
    print("%-46s %6d %9.2f %9.2f" % (k, len(v), v[len(v) // 2], v[int(len(v) * 0.9)]))
print("sum of the medians: %.1f us" % tot)
PY
