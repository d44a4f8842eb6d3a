#!/usr/bin/env python3
"""Throughput of the hot path with several contexts in flight on one GPU (one host thread and one HIP stream each,
all reading the same HBM-resident input): python tools/pipeline_probe.py [steps] [length]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import breakdancer_amd as bda
from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig, Options
from breakdancer_amd.synth import LIB_C2, make_chromosome

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
length = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
dev = torch.device("cuda", 0)
d = make_chromosome(length=length, seed=1, name_base=0)
n = len(d["tid"])
tens = {}
for k, dt in BATCH_FIELDS:
    arr = np.ascontiguousarray(d[k], dtype=dt)
    view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(arr.dtype)
    tens[k] = torch.from_numpy(arr.view(view) if view else arr).to(dev)
torch.cuda.synchronize()
for P in (1, 2, 3, 4):
    ctxs = []
    for _ in range(P):
        bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=0)
        bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
        for _ in range(3):
            bd.run()
        ctxs.append(bd)
    torch.cuda.synchronize()
    k1 = [[] for _ in range(P)]
    def work(i):
        for _ in range(steps // P):
            ctxs[i].run()
            k1[i].append(ctxs[i].timings()["classify"])
    th = [threading.Thread(target=work, args=(i,)) for i in range(P)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done = (steps // P) * P
    k1all = [x for l in k1 for x in l if x > 0]
    print("contexts %d: %.3f ms/step, %.2f G read-pairs/s, K1 avg %.1f us, svs %d" % (
        P, dt / done * 1e3, done * (n // 2) / dt / 1e9, 1e3 * float(np.mean(k1all)) if k1all else -1, ctxs[0].summary()["n_svs_printed"]), flush=True)
    for c in ctxs: c.close()
