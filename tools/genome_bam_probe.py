#!/usr/bin/env python3
"""BAM -> SV table at a GPU's share of a 30x genome: ONE indexed 24-chromosome, 4-library BAM (hg38 lengths x fraction; 1/8 = 116 M
records, ~16 GB), bin/breakdancer-max in one process.  Usage: genome_bam_probe.py [fraction] [runs] [KEY=VALUE ...] (extra environment
for the CLI); the BAM is kept in $BDX_PROBE_DIR (default /dev/shm/bdx_genome) between calls."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(td, fraction, realistic=False):
    from breakdancer_amd.bamwrite import write_genome_bam
    t0 = time.time()
    bam, cfg, n = write_genome_bam(td, fraction, realistic=realistic, tag="realistic" if realistic else "genome")   # (--realistic: reference-drawn bases, binned qualities, level 6)
    if time.time() - t0 > 1:
        print("genome: %d records, BAM of %.2f GB synthesised and written in %.1f s" % (n, os.path.getsize(bam) / 1e9, time.time() - t0), flush=True)
    return bam, cfg, n


if __name__ == "__main__":
    fraction = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    extra = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a)
    prof = "--prof" in sys.argv[3:]
    td = os.environ.get("BDX_PROBE_DIR", "/dev/shm/bdx_genome")
    bam, cfg, n = make(td, fraction, "--realistic" in sys.argv[3:])
    size = os.path.getsize(bam)
    for r in range(runs):
        env = dict(os.environ, BDX_TIMING="1", BDX_FOREGROUND="1", **extra)
        time.sleep(2.5)   # (untimed: the driver is still taking back the previous process's tens of GB)
        t0 = time.perf_counter()
        w0 = time.time()
        p = subprocess.run([os.path.join(ROOT, "bin", "breakdancer-max"), cfg], cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        w1 = time.time()
        rows = sum(1 for l in p.stdout.splitlines() if l and not l.startswith(b"#"))
        import re
        m0 = re.search(r"main\(\) entered at ([0-9.]+)", p.stderr.decode())
        m1 = re.search(r"_exit called at ([0-9.]+)", p.stderr.decode())
        if m0 and m1:
            print("      start -> main() %.3f s, main() -> _exit %.3f s, _exit -> process gone %.3f s" % (float(m0.group(1)) - w0, float(m1.group(1)) - float(m0.group(1)), w1 - float(m1.group(1))))
        import hashlib
        digest = hashlib.md5(b"\n".join(l for l in p.stdout.splitlines() if not l.startswith(b"#Command") and not l.startswith(b"#Software"))).hexdigest()[:12]
        print("run %d %s: rc %d, %.3f s, %.1f M read-pairs/s, %.2f GB/s of BAM, %d SV rows (md5 of the table %s)" % (r, extra, p.returncode, dt, n / 2 / dt / 1e6, size / dt / 1e9, rows, digest), flush=True)
        if r == runs - 1 or p.returncode:
            print("\n".join(l for l in p.stderr.decode().splitlines() if "bdx timing" in l or p.returncode))
    if prof:   # one more run under rocprofv3: kernel statistics and the timeline (gpurun_out/genome_kernel_stats.csv, genome_timeline.txt)
        import glob, shutil
        out = "/tmp/bdx_genome_prof"
        shutil.rmtree(out, ignore_errors=True)
        env = dict(os.environ, BDX_TIMING="1", BDX_CLEAN_EXIT="1", TMPDIR="/tmp", **extra)
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "p", "--",
                            os.path.join(ROOT, "bin", "breakdancer-max"), cfg], cwd=td, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        print("profiled run: rc %d" % p.returncode)
        print("\n".join(l for l in p.stderr.decode().splitlines() if "bdx timing" in l))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for f in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)[:1]:
            shutil.copy(f, os.path.join(ROOT, "gpurun_out", "genome_kernel_stats.csv"))
        tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), out, "2.0"], stdout=subprocess.PIPE).stdout.decode()
        open(os.path.join(ROOT, "gpurun_out", "genome_timeline.txt"), "w").write(tl)
        print(tl[-3500:])
