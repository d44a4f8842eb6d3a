#!/bin/bash
# per-kernel times of ONE context's runs at a GPU's share of a genome (default options, then -t): rocprofv3 --kernel-trace --stats around
# tools/genome_ab.py (bdx_run repeated on resident records).  usage: single_trace.sh <output directory>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${1:-$R/gpurun_out/single_trace}; case "$O" in /*) ;; *) O="$PWD/$O";; esac
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for t in default t; do
  flag=""; [ $t = t ] && flag="--t"
  rm -rf /tmp/ga_$t
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ga_$t -o p -- python $R/tools/genome_ab.py --rounds 2 $flag > "$O/single_$t.txt" 2>&1 < /dev/null
  f=$(find /tmp/ga_$t -name "p_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/single_${t}_kernel_stats.csv"
  grep -E "walk split|records|default" "$O/single_$t.txt"
  [ -n "$f" ] && python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:28]: print(r['Name'][:56].ljust(56), r['Calls'], r['AverageNs'], r['Percentage'])
" "$f"
done
