#!/usr/bin/env python3
"""Generate the configs[1] chromosome as a BAM (or reuse it) and time bin/breakdancer-max on it with the reader's profile on."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
length = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
td = "/tmp/bdx_cli_prof"
os.makedirs(td, exist_ok=True)
bam = os.path.join(td, "syn.bam")
if not os.path.exists(bam):
    d = make_chromosome(length=length, seed=1)
    t = time.time(); write_bam(bam, d, ["chrS"], seed=3); print("bam written in %.1fs, %d bytes" % (time.time() - t, os.path.getsize(bam)))
open(os.path.join(td, "cfg"), "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
for env_extra in [dict(BDX_BAM_PROFILE="1")] + [dict(BDX_THREADS=str(t)) for t in sys.argv[2:]] + [dict()]:
    env = dict(os.environ, BDX_TIMING="1", **env_extra)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), "cfg"], cwd=td, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        best = min(best, time.time() - t0)
    err = p.stderr.decode().splitlines()
    print(env_extra, "wall %.3f" % best)
    for l in err[-9:] if "BDX_BAM_PROFILE" in env_extra else err[-2:]:
        print("   ", l)
