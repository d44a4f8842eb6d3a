#!/usr/bin/env python3
"""K1/K2 timing when libraries and source files are interleaved inside every wave (configs[2]/[4] shape):
4 libraries over 2 BAMs, reads of all libraries mixed at random along one chromosome."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import breakdancer_amd as bda
from breakdancer_amd.api import LibraryConfig, Options
from breakdancer_amd.synth import make_chromosome, concat

L = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
libs = [LibraryConfig(400, 30, 490, 310, 100, bam_file_index=0), LibraryConfig(350, 40, 470, 230, 100, bam_file_index=0),
        LibraryConfig(500, 50, 650, 350, 100, bam_file_index=1), LibraryConfig(300, 25, 375, 225, 100, bam_file_index=1)]
parts = []
for i, l in enumerate(libs):
    parts.append(make_chromosome(length=L, coverage=7.5, seed=10 + i, lib=i, bam=l.bam_file_index, name_base=i << 40,
                                 mean=l.mean_insertsize, std=l.std_insertsize))
d = concat(parts)
order = np.argsort(d["pos"].astype(np.int64) * 2 + ((d["flag"] >> 4) & 1), kind="stable")
d = {k: v[order] for k, v in d.items()}
n = len(d["tid"])
for cn_lib in (False, True):
    bd = bda.BreakDancer(Options(CN_lib=cn_lib), libs, 2, max_read_window_size=100)
    bd.push_reads(d)
    for _ in range(3):
        bd.run()
    t = [bd.run().timings() for _ in range(10)]
    avg = {k: float(np.mean([x[k] for x in t])) for k in t[0]}
    s = bd.summary()
    print(json.dumps({"reads": n, "cn_lib": cn_lib, "stage_ms": avg, "k1_algo_GBps": 28 * n / avg["classify"] / 1e6, "svs": s["n_svs_printed"],
                      "regions": s["n_regions"]}))
    bd.close()
