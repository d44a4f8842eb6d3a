#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of kernels with KNOWN byte counts (tools/pmc_scatter_probe.hip): the counters' factor for scattered access.
# Run on the GPU box from the repo root; writes gpurun_out/pmc_scatter.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/pmc_scatter_probe $R/tools/pmc_scatter_probe.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -o p -- /tmp/pmc_scatter_probe > /tmp/pmcs_$c.log 2>&1 < /dev/null
done
mkdir -p $R/gpurun_out
python - <<'PY' | tee $R/gpurun_out/pmc_scatter.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcs_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 1 << 22
GiB = 1 << 30
known = {"stream_read16": ("1 GiB read, 16 B per lane", GiB, 0), "stream_write16": ("1 GiB written, 16 B per lane", 0, GiB),
         "gather8": ("n 8-byte loads, each from a random 64-byte line", n * 8, 0), "gather16": ("n 16-byte loads, random lines", n * 16, 0),
         "scatter4": ("n 4-byte stores to random lines", 0, n * 4), "scatter8": ("n 8-byte stores to random lines", 0, n * 8),
         "atomic_add4": ("n atomicAdd(u32) on random lines of 1 GiB", n * 4, n * 4), "atomic_cas8": ("n atomicCAS(u64) on random lines of 1 GiB", n * 8, n * 8),
         "atomic_cas8_small": ("n atomicCAS(u64) on a 4 MiB table", n * 8, n * 8)}
print("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each), median of 3 launches, KB as reported; n = %d" % n)
print("%-20s %-48s %12s %12s %14s %14s" % ("kernel", "what it moves", "FETCH_SIZE", "WRITE_SIZE", "FETCH B/elem", "WRITE B/elem"))
for k, (what, rb, wb) in known.items():
    f = sorted(agg[k]["FETCH_SIZE"])[len(agg[k]["FETCH_SIZE"]) // 2] if agg[k]["FETCH_SIZE"] else float("nan")
    w = sorted(agg[k]["WRITE_SIZE"])[len(agg[k]["WRITE_SIZE"]) // 2] if agg[k]["WRITE_SIZE"] else float("nan")
    per = (GiB // 16) if k.startswith("stream") else n
    print("%-20s %-48s %12.1f %12.1f %14.2f %14.2f" % (k, what, f, w, f * 1024 / per, w * 1024 / per))
PY
