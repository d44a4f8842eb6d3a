# A/B of decoder settings on the genome-share BAM (run on the GPU box from the repo root)
for v in "" "BDX_BAM_BATCH_ROUNDS=8" "BDX_BAM_PIECE_BYTES=16777216" "BDX_BAM_BATCH_ROUNDS=2" "BDX_BAM_AHEAD=2"; do
  echo "== $v"
  python tools/genome_bam_probe.py 0.125 4 $v | grep -E "^run|steady|inside the decoder|total=" | cut -c1-260
done
