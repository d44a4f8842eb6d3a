"""usage: kernel_sum.py <rocprofv3 kernel_stats.csv of tools/single_trace.sh> -> the kernels' time per bdx_run (15 runs in that trace), to set beside the
unprofiled wall time of tools/genome_ab.py on the same box"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    calls=int(r['Calls']); avg=float(r['AverageNs'])
    if 'noop' in r['Name'] or 'rocclr' in r['Name'] or 'side_kernel' in r['Name'] or 'walk_big' in r['Name']: continue
    tot+=avg*calls/15.0 if calls>=14 else 0
print('kernel time per run (sum of averages x calls/15): %.1f us'%(tot/1e3))
