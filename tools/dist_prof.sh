#!/bin/bash
# rocprofv3 launch / copy timeline of ONE sharded whole-genome run (tools/genome_probe.py --skip-single): every kernel and copy of the
# bdx_dist_run window in time order with the idle time in front of it.  Run on the GPU box from the repo root:  tools/dist_prof.sh [probe args]
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/distprof && mkdir -p /tmp/distprof && cd /tmp/distprof
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/distprof/out -o p -- python $R/tools/genome_probe.py --skip-single "$@" > /tmp/distprof/stdout.txt 2> /tmp/distprof/stderr.txt
grep '"ranks"' /tmp/distprof/stdout.txt | head -1 | cut -c1-900
t=$(find /tmp/distprof/out -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/distprof/out -name "*memory_copy_trace.csv" | head -1)
mkdir -p $R/gpurun_out
python - "$t" "$m" <<'PY' | tee $R/gpurun_out/dist_timeline.txt
import csv, re, sys
rows = []
for path, kind in ((sys.argv[1], "k"), (sys.argv[2] if len(sys.argv) > 2 else "", "c")):
    if not path:
        continue
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name") or r.get("Direction") or "?"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        mm = re.search(r"(k[0-9zb]+_[a-z0-9_]*|scan_[a-z0-9_]*|finalize[0-9a-z_]*|init_kernel|__amd_rocclr_[a-zA-Z]*|MEMORY_COPY_[A-Z_]*|nccl[A-Za-z_]*)", name)
        rows.append((s, e, (mm.group(1) if mm else name)[:34], kind))
rows.sort()
# the run's window: from the last k1_classify launch (the probe's one sharded run) to the end
k1 = [i for i, r in enumerate(rows) if r[2].startswith("k1_classify")]
i0 = k1[-1] if k1 else 0
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 300000 and not rows[i0 - 1][2].startswith("MEMORY_COPY_HOST_TO_DEVICE"):
    i0 -= 1
t0 = rows[i0][0]
busy_end = t0
print("%10s %9s %9s  %s" % ("start_us", "dur_us", "idle_us", "what"))
tot_busy = tot_idle = 0
for s, e, n, kind in rows[i0:]:
    idle = max(0, s - busy_end)
    print("%10.1f %9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle / 1e3, n))
    tot_idle += idle
    tot_busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
print("window %.1f us: busy %.1f us, idle %.1f us" % ((busy_end - t0) / 1e3, tot_busy / 1e3, tot_idle / 1e3))
PY
