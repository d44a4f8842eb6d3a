#!/bin/bash
# The round's profile set (run on the GPU box from the repo root): everything lands in gpurun_out/r06/ (copy what is to be judged into profiles/)
R=$(pwd)
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# configs[1]'s timed steps, kernel by kernel
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-genome --no-pmc --no-overlap > /tmp/ks.log 2>&1 < /dev/null
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_kernel_stats.csv
# the clustering path on a GPU's share of a genome, HBM-resident: one context and the sharded run on one rank, kernel by kernel
rm -rf /tmp/gp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o p -- python $R/tools/genome_probe.py --repeat 5 > $O/r06_genome_probe.txt 2>&1 < /dev/null
f=$(find /tmp/gp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_genome_kernel_stats.csv
cd $R
# one context alone (bdx_run repeated on resident records), default options and -t, kernel by kernel
bash tools/single_trace.sh $O > $O/r06_single_context_trace.txt 2>&1 < /dev/null
cd $R
# run-time switches A/B on one box: the LDS-first join, the region table's forward, the walk's lanes
timeout 500 python tools/genome_ab.py --rounds 3 default ins_plain=1 ins_plain=2 walk_lanes=16 > $O/r06_genome_ab.txt 2>&1 < /dev/null
# HBM traffic per kernel (PMC, one pass per counter) of a genome-share run
timeout 700 bash tools/pmc_traffic.sh genome > $O/r06_pmc_genome.txt 2>&1 < /dev/null
# the multi-GPU protocol with the ranks as threads on this GPU: 2 and 8 ranks, default options and -t, against the single context
for a in "--ranks 2" "--ranks 8" "--ranks 2 --t" "--ranks 8 --t"; do
  echo "== genome_probe.py --repeat 2 $a" >> $O/r06_thread_ranks.txt
  timeout 300 python tools/genome_probe.py --repeat 2 $a 2>&1 | grep -E "^sharded ==|^\{\"ranks\"" | cut -c1-6000 >> $O/r06_thread_ranks.txt
done
# in-kernel clocks of the walk and table kernels at the genome share (measurement build, if it was linked: variants/libbdx_kprof.so)
[ -f variants/libbdx_kprof.so ] && timeout 300 python tools/kprof.py --genome 0.125 > $O/r06_kprof_genome.txt 2>&1 < /dev/null
# a BAM that looks like one: the inflate kernel alone on 2.5 GB of it, every byte against zlib, matches by where their source came from
BDX_KZ_PROF=/tmp/kzprof.bin timeout 400 python tools/bamdec_probe.py --mbp 40 --realistic --check-zlib --inflate-only --slice-gb 4 > $O/r06_realistic_inflate_probe.txt 2>&1 < /dev/null
timeout 300 python tools/bamdec_probe.py --mbp 40 --realistic --inflate-only --slice-gb 4 >> $O/r06_realistic_inflate_probe.txt 2>&1 < /dev/null
# the CLI on the genome-share BAMs (random-base level 1, reference-drawn level 6): three runs each, then one under rocprofv3
timeout 900 python tools/genome_bam_probe.py 0.125 3 --prof > $O/r06_cli_genome_prof.txt 2>&1 < /dev/null
cp gpurun_out/genome_kernel_stats.csv $O/r06_cli_kernel_stats.csv 2>/dev/null
cp gpurun_out/genome_timeline.txt $O/r06_cli_timeline.txt 2>/dev/null
timeout 900 python tools/genome_bam_probe.py 0.125 3 --realistic --prof > $O/r06_cli_realistic_prof.txt 2>&1 < /dev/null
cp gpurun_out/genome_kernel_stats.csv $O/r06_cli_realistic_kernel_stats.csv 2>/dev/null
cp gpurun_out/genome_timeline.txt $O/r06_cli_realistic_timeline.txt 2>/dev/null
# the feed when N GPUs' feeders share this host; the same table every time at the genome share, 1-3 ranks
timeout 600 python tools/feed_scaling.py /dev/shm/bdx_genome/genome_0.125.bam 4 > $O/r06_feed_scaling.txt 2>&1 < /dev/null
for g in "" "BDX_GPUS=0,0" "BDX_GPUS=0,0,0"; do timeout 600 python tools/determinism_probe.py 0.125 10 $g 2>&1 | grep -E "^records|rc [1-9]" >> $O/r06_determinism_probe.txt; done
ls -la $O
