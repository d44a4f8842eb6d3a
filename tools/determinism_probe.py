#!/usr/bin/env python3
"""Does bin/breakdancer-max print the same table every time?  The genome-share BAM at a small fraction, N runs, distinct outputs kept and
diffed (tools, not the product).  usage: determinism_probe.py [fraction] [runs] [KEY=VALUE ...] [-option ...]"""
import difflib, hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_genome_bam
fraction = float(sys.argv[1]) if len(sys.argv) > 1 else 0.004
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
extra = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a and not a.startswith("-"))
cli_args = [a for a in sys.argv[3:] if a.startswith("-")]   # (options for breakdancer-max: -t, -a, -h ...)
td = "/dev/shm/bdx_det"
os.makedirs(td, exist_ok=True)
bam, cfg, n = write_genome_bam(td, fraction)
seen = {}
for r in range(runs):
    p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max")] + cli_args + [cfg], cwd=td, env=dict(os.environ, **dict({"BDX_FOREGROUND": "1"}, **extra)), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = "\n".join(l for l in p.stdout.decode().splitlines() if not l.startswith("#Command") and not l.startswith("#Software"))
    h = hashlib.md5(text.encode()).hexdigest()[:10]
    seen.setdefault(h, [0, text, p.returncode, p.stderr.decode()])
    seen[h][0] += 1
print("records %d, runs %d %s %s: %s" % (n, runs, extra, cli_args, {h: (v[0], "rc %d" % v[2], "%d lines" % len(v[1].splitlines())) for h, v in seen.items()}))
keys = list(seen)
for h in keys:
    print("== stderr of a run with table %s:" % h)
    print(seen[h][3][-2500:])
if len(keys) > 1:
    a, b = seen[keys[0]][1].splitlines(), seen[keys[1]][1].splitlines()
    for l in list(difflib.unified_diff(a, b, lineterm="", n=0))[:14]:
        print(l[:260])
