#!/bin/bash
# HBM traffic counters of every kernel of a step (one PMC pass per counter; run on the GPU box from the repo root)
# usage: tools/pmc_traffic.sh [step|genome]    step: configs[1]'s step alone (bench.py --no-genome); genome: one sharded run of the
#        116 M-record genome share (tools/genome_probe.py --skip-single)
R=$(pwd)
if [ "${1:-step}" = genome ]; then CMD="python $R/tools/genome_probe.py --skip-single --repeat 1"; else CMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc"; fi
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmct_$c -o p -- $CMD > /tmp/pmct_$c.log 2>&1 < /dev/null
done
cd $R
python - "$CMD" <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmct_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("bdx::", "")[:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
import sys
print("rocprofv3 --kernel-trace --pmc <counter> (one pass per counter) -- %s; mean Counter_Value per launch in KB." % sys.argv[1])
print("gfx950 correction (MI355X_MICROARCH.md, calibrated in r01_k1_pmc.txt): HBM bytes read = FETCH_SIZE x 2 KB; WRITE_SIZE as reported.")
print("%-50s %12s %12s %8s" % ("kernel", "FETCH_SIZE", "WRITE_SIZE", "launches"))
rows = sorted(agg, key=lambda k: -agg[k]["FETCH_SIZE"] / max(1, cnt[k]["FETCH_SIZE"]))
for k in rows:
    print("%-50s %12.1f %12.1f %8d" % (k, agg[k]["FETCH_SIZE"] / max(1, cnt[k]["FETCH_SIZE"]), agg[k]["WRITE_SIZE"] / max(1, cnt[k]["WRITE_SIZE"]), cnt[k]["FETCH_SIZE"]))
PY
