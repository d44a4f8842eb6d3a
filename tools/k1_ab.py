#!/usr/bin/env python3
"""K1's own time (HIP events around the kernel, bdx_set_stage_timing) with several builds of libbdx.so, interleaved on one box:
   python tools/k1_ab.py [--rounds 3] variants/libbdx_a.so variants/libbdx_b.so ...
Each build runs in its own process on configs[1] (15 M records, one library) and on the genome share (116 M records, 4 libraries in one file)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(lib_path, what):
    import numpy as np
    import breakdancer_amd._lib as lib
    lib.LIB_PATH = os.path.abspath(lib_path)
    import torch
    import breakdancer_amd.api as bda
    from breakdancer_amd.api import LibraryConfig, Options
    dev = torch.device("cuda", 0)
    if what == "configs1":
        from breakdancer_amd.synth import LIB_C2, make_chromosome
        d = make_chromosome(length=50_000_000, seed=1)
        libs, nbams, ntids = [LibraryConfig(**LIB_C2)], 1, 1
        n = len(d["pos"])
        tens = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
        bd = bda.BreakDancer(Options(), libs, nbams, ntids=ntids, max_read_window_size=200, device=0)
        bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
    else:
        from breakdancer_amd.synth import make_genome
        from tools.genome_probe import HG38_MBP, LIBS4
        lengths = [int(m * 1e6 * 0.125) for m in HG38_MBP]
        libs = [LibraryConfig(mean_insertsize=m, std_insertsize=sd, uppercutoff=m + 3 * sd, lowercutoff=m - 3 * sd, readlens=100.0, name="lib%d" % i)
                for i, (m, sd) in enumerate(LIBS4)]
        d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=5000)
        n = len(d["tid"])
        bd = bda.BreakDancer(Options(), libs, 1, ntids=len(lengths), max_read_window_size=200, device=0)
        bd.lib.bdx_reserve(bd.h, n)
        bd.push_reads(d)
        torch.cuda.synchronize()
        bd.run()
        bd.set_enqueue_ahead(0)
    for _ in range(5):
        bd.run()
    bd.set_stage_timing(True)
    k1, tot = [], []
    for _ in range(30):
        bd.run()
        t = bd.timings()
        k1.append(t["classify"]); tot.append(t["total"])
    print(json.dumps({"lib": os.path.basename(lib_path), "what": what, "reads": n, "k1_ms_avg": float(np.mean(k1)), "k1_ms_min": float(np.min(k1)),
                      "k1_algo_TBps": 28 * n / float(np.mean(k1)) / 1e9, "run_ms_avg": float(np.mean(tot))}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
        sys.exit(0)
    args = sys.argv[1:]
    rounds = 3
    if args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    for what in ("configs1", "genome"):
        for r in range(rounds):
            for l in args:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", l, what], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                out = [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
                print(out[-1] if out else "FAILED %s %s: %s" % (l, what, p.stderr.decode()[-400:]), flush=True)
