#!/usr/bin/env python3
"""One-off scale validation on a multi-chromosome, multi-library, two-file input with translocations (configs[2]/[3]/[4]
shapes at tens of millions of reads): product vs oracle, default options and -a -h."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_opts
from runner import compare, product_from_oracle
from test_gpu_configs import cfg_line, oracle_from_soa
from breakdancer_amd.synth import make_genome

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lengths = [int(60e6 * scale), int(45e6 * scale), int(30e6 * scale), int(20e6 * scale)]
libs = ((400.0, 30.0), (330.0, 25.0), (480.0, 45.0), (300.0, 25.0))
t0 = time.time()
d = make_genome(lengths, coverage=30.0, seed=77, libs=libs, lib_bam=(0, 0, 1, 1), n_translocations=int(4000 * scale))
print("generated", len(d["tid"]), "reads in %.1fs" % (time.time() - t0), flush=True)
cfg = "".join(cfg_line("rg%d" % i, "a.bam" if i < 2 else "b.bam", "lib%d" % i, m, s) for i, (m, s) in enumerate(libs))
SETS = (dict(), dict(cn_lib=1, print_af=1), dict(transchr_rearrange=1))
if len(sys.argv) > 2 and sys.argv[2] == "more":
    SETS = (dict(buffer_size=1), dict(buffer_size=3, min_read_pair=1), dict(min_read_pair=4), dict(chr_tid=1), dict(min_len=50, seq_coverage_lim=5),
            dict(illumina_long_insert=1), dict(max_sd=700), dict(fisher=1, score_threshold=0), dict(min_map_qual=10))
for kw in SETS:
    t0 = time.time()
    run = oracle_from_soa(d, cfg, ["a.bam", "b.bam"], make_opts(**kw), ["c1", "c2", "c3", "c4"])
    print(kw, "oracle %.1fs: regions %d svs %d" % (time.time() - t0, run.n_regions, run.n_svs), flush=True)
    bd = product_from_oracle(run)
    s = compare(run, bd)
    print("  product == oracle; device / host SVs, host groups:", bd.walk_split(), "total ms", bd.timings()["total"], flush=True)
    bd.close()
