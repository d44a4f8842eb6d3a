#!/bin/bash
# A/B of library builds under the CLI on one configs[1]-shaped BAM: tools/cli_ab.sh <mbp> <lib[:ENV=VAL,...]>...  (GPU box, repo root; the library
# is swapped in place and restored).  Per build three runs of bin/breakdancer-max (one process, exit included) with their BDX_TIMING lines -> gpurun_out/cli_ab.txt
MBP=${1:-50}; shift
mkdir -p gpurun_out /tmp/cli_ab
python - $MBP <<'PY'
import sys
sys.path.insert(0, ".")
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
write_bam("/tmp/cli_ab/syn.bam", make_chromosome(length=int(float(sys.argv[1]) * 1e6), seed=1), ["chrS"], seed=3)
open("/tmp/cli_ab/cfg", "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
PY
R=$(pwd)
cp breakdancer_amd/libbdx.so /tmp/libbdx_keep.so
for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    [ "$spec" != "$lib" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    cp $lib breakdancer_amd/libbdx.so
    echo "== $spec"
    for i in $(seq 1 ${RUNS:-3}); do
        sleep 0.5
        ( cd /tmp/cli_ab && s=$(date +%s%N) && env BDX_TIMING=1 BDX_FOREGROUND=1 $envs $R/bin/breakdancer-max cfg > out.txt 2> err.txt; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms, $(grep -vc '^#' out.txt) rows"; grep "device decode\|inside the decoder\|total=" err.txt | cut -c1-260 )
    done
done 2>&1 | tee gpurun_out/cli_ab.txt
cp /tmp/libbdx_keep.so breakdancer_amd/libbdx.so
