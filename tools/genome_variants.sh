# A/B of feeder settings on the genome-share BAM (run on the GPU box from the repo root): libbdx.so rebuilt with other staging counts
python tools/genome_bam_probe.py 0.125 2 | tail -8
python tools/genome_bam_probe.py 0.125 2 BDX_BAM_PIECE_BYTES=16777216 | tail -8
python tools/genome_bam_probe.py 0.125 2 BDX_BAM_PIECE_BYTES=33554432 | tail -8
for n in 12; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Iinclude -DBDX_BAM_STAGING=$n -c breakdancer_amd/csrc/bdx_api.hip -o /tmp/bdx_api_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o breakdancer_amd/libbdx.so $(ls breakdancer_amd/csrc/*.o | grep -v bdx_api.o) /tmp/bdx_api_$n.o
  echo "== staging $n"
  python tools/genome_bam_probe.py 0.125 2 | tail -8
  python tools/genome_bam_probe.py 0.125 2 BDX_BAM_PIECE_BYTES=16777216 | tail -8
done
