import sys,json
for line in sys.stdin:
    if line.startswith('{"first_run_seconds"'):
        d=json.loads(line)
        for r in d["runs"]: print(round(r["seconds"]*1e3,3), {k:v for k,v in r["stage_ms"].items() if v})
        break
