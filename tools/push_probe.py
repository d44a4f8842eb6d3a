#!/usr/bin/env python3
"""Where timing (ii) goes: bdx_push (copies enqueued) / copies complete / bdx_run, at configs[1] size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
import breakdancer_amd as bda
from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig, Options
from breakdancer_amd.synth import LIB_C2, make_chromosome
d = make_chromosome(length=50_000_000, seed=1)
n = len(d["tid"])
views = {}
keep = []
for k, dt in BATCH_FIELDS:
    arr = np.ascontiguousarray(d[k], dtype=dt)
    view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(arr.dtype)
    t = torch.from_numpy(arr.view(view) if view else arr).pin_memory()
    keep.append(t)
    views[k] = t.numpy().view(dt)
for it in range(4):
    bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=0)
    bd.lib.bdx_reserve(bd.h, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bd.push_reads(views)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    bd.run()
    t3 = time.perf_counter()
    print("push call %.3f ms, copies + K1 done after %.3f ms, run %.3f ms, total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
    bd.close()
