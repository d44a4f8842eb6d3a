"""bin/breakdancer-max on a configs[1]-shaped BAM with BDX_TIMING, device and host readers, a few settings (tools, not the product)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
EXE = os.path.join(ROOT, "bin", "breakdancer-max")
with tempfile.TemporaryDirectory(prefix="bdx_cli_") as td:
    d = make_chromosome(length=int(mbp * 1e6), seed=1)
    write_bam(os.path.join(td, "syn.bam"), d, ["chrS"], seed=3)
    open(os.path.join(td, "cfg"), "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
    for label, env in (("device", {}), ("device", {}), ("device piece 8MB", {"BDX_BAM_PIECE_BYTES": str(8 << 20)}), ("device piece 64MB", {"BDX_BAM_PIECE_BYTES": str(64 << 20)}),
                       ("host", {"BDX_DECODE": "host"})):
        t0 = time.perf_counter()
        p = subprocess.run([EXE, "cfg"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BDX_TIMING="1", **env))
        dt = time.perf_counter() - t0
        print("== %s: wall %.3f s, rc %d, %d rows" % (label, dt, p.returncode, len([l for l in p.stdout.decode().splitlines() if not l.startswith("#")])))
        print(p.stderr.decode().strip())
