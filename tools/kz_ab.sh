#!/bin/bash
# A/B of inflate-kernel builds on one file: tools/kz_ab.sh <mbp> <variants/libbdx_x.so>...   (GPU box, repo root; the library is swapped in place
# and restored).  Per build: the kernel alone in launches of <= 1.5 GB and in one launch, and its own clocks (cycles per step) -> gpurun_out/kz_ab.txt
MBP=${1:-50}; shift
mkdir -p gpurun_out
python - $MBP <<'PY'
import sys
sys.path.insert(0, ".")
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
write_bam("/tmp/kz_ab.bam", make_chromosome(length=int(float(sys.argv[1]) * 1e6), seed=1), ["chrS"], seed=3, level=1)
PY
cp breakdancer_amd/libbdx.so /tmp/libbdx_keep.so
for lib in "$@"; do
    cp $lib breakdancer_amd/libbdx.so
    echo "== $lib"
    for sl in 1.5 8; do python tools/bamdec_probe.py --bam /tmp/kz_ab.bam --inflate-only --slice-gb $sl 2>/dev/null | grep "inflate kernel" | sed "s/^/slice $sl: /"; done
    BDX_KZ_PROF=/tmp/kzprof.bin python tools/bamdec_probe.py --bam /tmp/kz_ab.bam --inflate-only --slice-gb 8 2>/dev/null | grep "kernel clocks" | tail -1
done 2>&1 | tee gpurun_out/kz_ab.txt
cp /tmp/libbdx_keep.so breakdancer_amd/libbdx.so
