#!/usr/bin/env python3
"""What bin/breakdancer-max allocates on the way from a configs[1] BAM to the table, and how long the process takes to exit after its
last line (one process: BDX_FOREGROUND=1).  usage: exit_probe.py [Mbp]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG_LINE
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
td = tempfile.mkdtemp(prefix="bdx_exit_", dir="/tmp")
d = make_chromosome(length=int(mbp * 1e6), seed=1)
write_bam(os.path.join(td, "syn.bam"), d, ["chrS"], seed=3)
open(os.path.join(td, "cfg"), "w").write(CFG_LINE % "syn.bam")
exe = os.path.join(ROOT, "bin", "breakdancer-max")
for env_extra in ({}, {"BDX_ALLOC_TRACE": "1"}):
    for rep in range(3):
        time.sleep(1.0)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "cfg"], cwd=td, env=dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **env_extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        err = p.stderr.decode()
        tot = [l for l in err.splitlines() if "total=" in l]
        inside = float(tot[0].split("total=")[1].split("s")[0]) if tot else float("nan")
        print("wall %.3f s, inside %.3f s, exit %.3f s%s" % (dt, inside, dt - inside, "  (alloc trace on)" if env_extra else ""))
    if env_extra:
        dev = pin = 0
        big = []
        for l in err.splitlines():
            if l.startswith("[bdx alloc]"):
                f = l.split()
                kind, size = f[2], int(f[3])
                if kind == "device": dev += size
                else: pin += size
                if size >= 64 << 20: big.append(l[:120])
        print("device bytes %.2f GB, pinned/other bytes %.2f GB; allocations >= 64 MB:" % (dev / 1e9, pin / 1e9))
        print("\n".join(big))
