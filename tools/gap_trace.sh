#!/bin/bash
# idle time on the stream before every kernel of a step (rocprofv3 kernel trace of the default bench run; run on the GPU box from the repo root)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -o g -- python $R/bench.py --steps ${1:-40} --warmup 5 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc --no-overlap > /tmp/prof_gap.log 2>&1
cd $R
f=$(find /tmp/prof_gap -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[n1.replace("bdx::", "")[:44]].append((s1 - e0) / 1e3)
print("%-46s %6s %9s %9s" % ("kernel (gap = its start - previous kernel's end)", "n", "median us", "p90 us"))
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
    v = sorted(v)
    print("%-46s %6d %9.2f %9.2f" % (k, len(v), v[len(v) // 2], v[int(len(v) * 0.9)]))
PY
