#!/bin/bash
# rocprofv3 kernel statistics of one bin/breakdancer-max run on a configs[1]-shaped BAM (run on the GPU box from the repo root)
R=$(pwd)
MBP=${1:-50}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cliprof && mkdir -p /tmp/cliprof && cd /tmp/cliprof
python - <<PY
import sys
sys.path.insert(0, "$R")
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
d = make_chromosome(length=int($MBP * 1e6), seed=1)
write_bam("syn.bam", d, ["chrS"], seed=3)
open("cfg", "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
PY
BDX_CLEAN_EXIT=1 BDX_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/cliprof/out -o p -- $R/bin/breakdancer-max cfg > /tmp/cliprof/stdout.txt 2> /tmp/cliprof/stderr.txt
f=$(find /tmp/cliprof/out -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-150; grep "bdx timing" /tmp/cliprof/stderr.txt
mkdir -p $R/gpurun_out && cp "$f" $R/gpurun_out/cli_kernel_stats.csv
# the launches in time order: start (ms after the first launch), duration, queue -- where the GPU waits for the host and vice versa
t=$(find /tmp/cliprof/out -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/cliprof/out -name "*memory_copy_trace.csv" | head -1)
python - "$t" "$m" <<'PY' | tee $R/gpurun_out/cli_timeline.txt
import csv, sys
rows = []
for path, kind in ((sys.argv[1], "k"), (sys.argv[2] if len(sys.argv) > 2 else "", "c")):
    if not path:
        continue
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name") or r.get("Direction") or "?"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        import re
        mm = re.search(r"(k[0-9zb]_?[a-z0-9_]*|scan_[a-z0-9_]*|finalize_kernel|__amd_rocclr_[a-zA-Z]*|MEMORY_COPY_[A-Z_]*)", name)
        rows.append((s, e, (mm.group(1) if mm else name)[:28], r.get("Queue_Id", r.get("Stream_Id", "")), kind))
rows.sort()
t0 = rows[0][0]
busy_end = t0
idle = 0
for s, e, n, q, kind in rows:
    if (e - s > 200000 and not n.startswith("MEMORY")) or n.startswith("kz") or n.startswith("kb_st"):
        print("%9.3f ms  +%8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
cp = [(s - t0) / 1e6 for s, e, n, q, kind in rows if n.startswith("MEMORY_COPY_HOST_TO_DEVICE") and e - s > 100000]
print("H2D piece copies start at (ms): " + " ".join("%.1f" % x for x in cp))
cd = sorted((e - s) / 1e6 for s, e, n, q, kind in rows if n.startswith("MEMORY_COPY_HOST_TO_DEVICE") and e - s > 100000)
if cd:
    print("H2D piece copies: %d, duration min %.3f / median %.3f / max %.3f ms, %.1f ms in all" % (len(cd), cd[0], cd[len(cd) // 2], cd[-1], sum(cd)))
print("span %.3f ms, GPU idle inside it %.3f ms, launches %d" % ((busy_end - t0) / 1e6, idle / 1e6, len(rows)))
PY
