"""bin/breakdancer-max on the configs[1] BAM (15 M records, 2 GB) with BDX_TIMING + BDX_BAMDEC_TRACE: where the time between the GPU context
and the first inflate launch goes (tools, not the product).  usage: python tools/cli_trace_probe.py [mbp] [runs]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
EXE = os.path.join(ROOT, "bin", "breakdancer-max")
with tempfile.TemporaryDirectory(prefix="bdx_cli_", dir="/dev/shm") as td:
    d = make_chromosome(length=int(mbp * 1e6), seed=1)
    write_bam(os.path.join(td, "syn.bam"), d, ["chrS"], seed=3)
    open(os.path.join(td, "cfg"), "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
    for r in range(runs):
        time.sleep(1.0)
        t0 = time.perf_counter()
        p = subprocess.run([EXE, "cfg"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **({"BDX_BAMDEC_TRACE": "1"} if r == runs - 1 else {})))
        dt = time.perf_counter() - t0
        print("== run %d: wall %.3f s, rc %d" % (r, dt, p.returncode))
        print(p.stderr.decode().strip())
