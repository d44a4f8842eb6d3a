#!/bin/bash
# reader scaling: inflate / parse CPU sums of the CLI at several thread counts (BAM from tools/cli_profile.py in /tmp/bdx_cli_prof)
cd /tmp/bdx_cli_prof
for t in 8 16 32 64 96; do
  for madv in 0 1; do
    if [ $madv = 1 ]; then export BDX_BAM_NO_MADV=1; else unset BDX_BAM_NO_MADV; fi
    echo "threads $t no_madv $madv"
    BDX_THREADS=$t BDX_BAM_PROFILE=1 BDX_TIMING=1 /root/repo/bin/breakdancer-max cfg 2>&1 >/dev/null | tail -3
  done
done
