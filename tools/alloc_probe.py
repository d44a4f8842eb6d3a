import os, subprocess, sys, tempfile, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
EXE = os.path.join(ROOT, "bin", "breakdancer-max")
with tempfile.TemporaryDirectory(prefix="bdx_cli_") as td:
    d = make_chromosome(length=int(50e6), seed=1)
    write_bam(os.path.join(td, "syn.bam"), d, ["chrS"], seed=3)
    open(os.path.join(td, "cfg"), "w").write("readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n")
    for label, env in (("plain", {}), ("plain", {}), ("plain", {})):
        time.sleep(1.0)
        t0 = time.perf_counter()
        p = subprocess.run([EXE, "cfg"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BDX_TIMING="1", **env))
        dt = time.perf_counter() - t0
        print("== %s: wall %.3f s, rc %d" % (label, dt, p.returncode))
        err = p.stderr.decode().strip().splitlines()
        big = []
        print("\n".join([l for l in err if "[bdx alloc]" not in l] + big[:80]))
