#!/bin/bash
# links a measurement build of libbdx.so: tools/link_variant.sh <name> <replacement k6 object>   (run from the repo root, after make)
set -e
C=breakdancer_amd/csrc
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libbdx_$1.so $C/k1_classify.o $C/k2_compact.o $C/k3_regions.o $C/k4_join.o $C/k5_poisson.o $2 $C/k7_exchange.o $C/bdx_api.o $C/bdx_walk.o $C/bdx_walk_reads.o
