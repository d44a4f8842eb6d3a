#!/usr/bin/env python3
"""bin/breakdancer-max with BDX_GPUS on ONE indexed 24-chromosome BAM (hg38 x fraction, 30x): the BDX_TIMING breakdown of the sharded
run whose ranks decode their chromosomes on their own GPUs, beside the same file on one GPU.  Usage: cli_probe_sharded.py [fraction] [gpus]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG_LINE, HG38_MBP  # noqa: E402
from breakdancer_amd.bamwrite import write_bam  # noqa: E402
from breakdancer_amd.synth import make_genome  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0 / 64
gpus = sys.argv[2] if len(sys.argv) > 2 else "0,0"
td = tempfile.mkdtemp(prefix="bdx_shprobe_", dir="/tmp")
lengths = [int(m * 1e6 * frac) for m in HG38_MBP]
d = make_genome(lengths, coverage=30.0, seed=21, n_translocations=max(20, int(600 * frac * 64)))
write_bam(os.path.join(td, "genome.bam"), d, ["chr%d" % (i + 1) for i in range(len(lengths))], seed=5, index=True)
open(os.path.join(td, "gcfg"), "w").write(CFG_LINE % "genome.bam")
print("records", len(d["tid"]), "bam bytes", os.path.getsize(os.path.join(td, "genome.bam")), flush=True)
for label, env in (("sharded " + gpus, dict(BDX_GPUS=gpus)), ("one gpu", dict())):
    for rep in range(3):
        time.sleep(1.0)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), "gcfg"], cwd=td, env=dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
    print("== %s: %.3f s (last of three), rc %d, %d rows" % (label, dt, p.returncode, sum(1 for l in p.stdout.splitlines() if l and not l.startswith(b"#"))))
    lines = [l for l in p.stderr.decode().splitlines() if l.startswith("[bdx timing]")]
    dec = [l for l in lines if "device decode:" in l]
    print("\n".join(l[:330] for l in lines if "device decode:" not in l and "inside the decoder" not in l and "] rank " not in l))
    print("\n".join(l[:200] for l in lines if "] rank " in l))
    print("decoders: %d; first three:" % len(dec))
    print("\n".join(l[:330] for l in dec[:3]))
