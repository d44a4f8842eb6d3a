#!/usr/bin/env python3
"""Steady-state step time of a dense synthetic case (200 Mbp, 8 % discordant pairs, 60 M reads, ~166 k SV candidates) with
the general device walk off and on: python tools/dense_case_probe.py  (BDX_WALK_PROFILE=1 adds the host walk's share)."""
import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import breakdancer_amd as bda
from breakdancer_amd.api import LibraryConfig, Options
from breakdancer_amd.synth import LIB_C2, make_chromosome
d = make_chromosome(length=200_000_000, seed=21, discordant=0.08)
n = len(d["tid"])
for mode in ("0", "1"):
    os.environ["BDX_BIG_WALK"] = mode
    bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, max_read_window_size=200)
    bd.push_reads(d)
    for i in range(4):
        t0 = time.perf_counter(); bd.run(); dt = time.perf_counter() - t0
    print("big", mode, "run %.2f ms" % (dt * 1e3), "split", bd.walk_split(), {k: round(v, 2) for k, v in bd.timings().items()}, flush=True)
    bd.close()
