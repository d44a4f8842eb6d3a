"""stdin: tools/genome_probe.py's output -> one line per run of the single context: milliseconds and the host's stage timings (bdx_get_timings)"""
import sys,json
for line in sys.stdin:
    if line.startswith('{"first_run_seconds"'):
        d=json.loads(line)
        for r in d["runs"]: print(round(r["seconds"]*1e3,3), {k:v for k,v in r["stage_ms"].items() if v})
        break
