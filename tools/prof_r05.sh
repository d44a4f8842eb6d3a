#!/bin/bash
# The round's profile set (run on the GPU box from the repo root): everything lands in gpurun_out/r05/
R=$(pwd)
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc --no-overlap > /tmp/ks.log 2>&1 < /dev/null
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r05/r05_kernel_stats.csv
cd $R
# BAM -> table at a GPU's share of a genome: three runs, then one under rocprofv3 (kernel statistics + launch timeline)
timeout 900 python tools/genome_bam_probe.py 0.125 3 --prof > gpurun_out/r05/r05_cli_genome_prof.txt 2>&1 < /dev/null
cp gpurun_out/genome_kernel_stats.csv gpurun_out/r05/r05_cli_kernel_stats.csv 2>/dev/null
cp gpurun_out/genome_timeline.txt gpurun_out/r05/r05_cli_timeline.txt 2>/dev/null
for h in 0 1 2; do bin/bdx-feed-probe /dev/shm/bdx_genome/genome_0.125.bam 6 16 12 $h; done > gpurun_out/r05/r05_feed_probe.txt 2>&1
# the clustering path on the records of that genome share, HBM-resident: one context and the sharded run, kernel by kernel
cd /tmp && rm -rf /tmp/gp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o p -- python $R/tools/genome_probe.py --repeat 5 > $R/gpurun_out/r05/r05_genome_probe.txt 2>&1 < /dev/null
f=$(find /tmp/gp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r05/r05_genome_kernel_stats.csv
cd $R
timeout 600 bash tools/pmc_traffic.sh genome > gpurun_out/r05/r05_pmc_genome.txt 2>&1 < /dev/null
timeout 600 bash tools/dist_prof.sh > gpurun_out/r05/r05_dist_timeline.txt 2>&1 < /dev/null
ls -la gpurun_out/r05
