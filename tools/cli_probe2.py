"""bin/breakdancer-max on a two-BAM configuration (two libraries, same chromosome: a tumour / normal pair in miniature), device and host readers (tools, not the product)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakdancer_amd.bamwrite import write_bam
from breakdancer_amd.synth import make_chromosome
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
EXE = os.path.join(ROOT, "bin", "breakdancer-max")
with tempfile.TemporaryDirectory(prefix="bdx_cli2_") as td:
    lines = []
    for b, cov in enumerate((20.0, 10.0)):
        d = make_chromosome(length=int(mbp * 1e6), seed=1 + b, coverage=cov, name_base=b << 40)
        write_bam(os.path.join(td, "f%d.bam" % b), d, ["chrS"], rg="rg%d" % b, seed=3 + b)
        lines.append("readgroup:rg%d\tplatform:illumina\tmap:f%d.bam\treadlen:100.00\tlib:lib%d\tnum:10001\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n" % (b, b, b))
        print("file %d: %d records, %.0f MB" % (b, len(d["tid"]), os.path.getsize(os.path.join(td, "f%d.bam" % b)) / 1e6))
    open(os.path.join(td, "cfg"), "w").write("".join(lines))
    outs = {}
    for label, env in (("device", {}), ("host", {"BDX_DECODE": "host"}), ("device", {})):
        time.sleep(1.0)
        t0 = time.perf_counter()
        p = subprocess.run([EXE, "cfg"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BDX_TIMING="1", **env))
        dt = time.perf_counter() - t0
        rows = [l for l in p.stdout.decode().splitlines() if not l.startswith("#")]
        outs.setdefault(label, rows)
        print("== %s: wall %.3f s, rc %d, %d rows" % (label, dt, p.returncode, len(rows)))
        print("\n".join(l for l in p.stderr.decode().strip().splitlines() if "files decoded" in l or "reads=" in l))
    print("same table:", outs["device"] == outs["host"])
