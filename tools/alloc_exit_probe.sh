# where a genome-sized run's set-up goes: the decoder's marks and every allocation (run on the GPU box from the repo root)
python tools/genome_bam_probe.py 0.125 1 > /dev/null
cd /dev/shm/bdx_genome
for i in 1 2; do
BDX_BAMDEC_TRACE=1 BDX_ALLOC_TRACE=1 BDX_TIMING=1 BDX_FOREGROUND=1 /root/repo/bin/breakdancer-max genome_0.125.cfg 2> /root/repo/gpurun_out/alloc_trace.txt > /dev/null
done
grep -n "bamdec create\|bdx timing" /root/repo/gpurun_out/alloc_trace.txt | cut -c1-220
