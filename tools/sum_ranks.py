"""stdin: tools/genome_probe.py --ranks N output -> the run's milliseconds, rank 0's and rank 1's phases above 0.02 ms, rank 0's own part"""
import sys,json
for line in sys.stdin:
    if not line.startswith('{"ranks"'): 
        if line.startswith('sharded =='): print('  ',line.strip())
        continue
    d=json.loads(line)
    p=d['phase_ms']
    r0=p[0]
    tot=d.get('second_run_seconds') or d['seconds']
    only=sum(v for k,v in r0.items() if k.startswith('rank0_only'))
    print('ranks',d['ranks'],'first %.3f ms second %.3f ms'%(d['seconds']*1e3,(d.get('second_run_seconds') or 0)*1e3),'walk_split',d.get('walk_split'))
    print('   rank0:',{k[:30]:v for k,v in r0.items() if v>0.02}, 'rank0_only sum %.3f'%only)
    if len(p)>1: print('   rank1:',{k[:30]:v for k,v in p[1].items() if v>0.02})
