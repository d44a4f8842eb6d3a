#!/usr/bin/env python3
"""Timing (ii) of SURVEY.md 8(d) alone: pinned host SoA -> SV table at configs[1] size (bench.py's time_host_soa)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
import breakdancer_amd as bda
from breakdancer_amd.api import LibraryConfig, Options
from breakdancer_amd.synth import LIB_C2, make_chromosome
d = make_chromosome(length=int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, seed=1)
n = len(d["tid"])
for _ in range(2):
    r = bench.time_host_soa(bda, Options, LibraryConfig, LIB_C2, d, n, 0, torch)
    print("%.3f ms  %.3f G read-pairs/s  (%.1f GB/s at 25 B/read)" % (r["seconds"] * 1e3, r["value"] / 1e9, 25 * n / r["seconds"] / 1e9))
