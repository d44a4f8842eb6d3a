#!/bin/bash
# The round's profile set (run on the GPU box from the repo root): everything lands in gpurun_out/r04/
R=$(pwd)
mkdir -p $R/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc --no-overlap > /tmp/ks.log 2>&1 < /dev/null
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04/r04_kernel_stats.csv
cd $R
timeout 600 bash tools/cli_prof.sh 50 > gpurun_out/r04/r04_cli_prof.txt 2>&1 < /dev/null; cp gpurun_out/cli_kernel_stats.csv gpurun_out/r04/r04_cli_kernel_stats.csv 2>/dev/null
timeout 600 python tools/genome_probe.py > gpurun_out/r04/r04_genome_probe_after.txt 2>&1 < /dev/null
timeout 600 python tools/genome_probe.py --t >> gpurun_out/r04/r04_genome_probe_after.txt 2>&1 < /dev/null
timeout 600 bash tools/dist_prof.sh > gpurun_out/r04/r04_dist_timeline.txt 2>&1 < /dev/null
timeout 600 python tools/cli_probe_sharded.py > gpurun_out/r04/r04_cli_sharded.txt 2>&1 < /dev/null
ls -la gpurun_out/r04
