#!/bin/bash
# links a measurement build of libbdx.so: tools/link_variant.sh <name> <replacement object>...   (run from the repo root, after make)
# a replacement object named like one of the library's (k1_classify*.o, k6_assemble*.o, ...) takes its place
set -e
C=breakdancer_amd/csrc
mkdir -p variants
name=$1; shift
objs=""
for o in k1_classify k2_compact k3_regions k4_join k5_poisson k6_assemble k7_exchange k9_shard kz_inflate kb_records kc_insert_stats bdx_api bdx_walk bdx_walk_reads; do
    use=$C/$o.o
    for r in "$@"; do case "$(basename $r)" in ${o%%_*}_*) use=$r;; esac; done
    objs="$objs $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libbdx_$name.so $objs
