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
BDX_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cliprof/out -o p -- $R/bin/breakdancer-max cfg > /tmp/cliprof/stdout.txt 2> /tmp/cliprof/stderr.txt
f=$(find /tmp/cliprof/out -name "*kernel_stats.csv" | head -1)
head -25 "$f"
mkdir -p $R/gpurun_out && cp "$f" $R/gpurun_out/cli_kernel_stats.csv
