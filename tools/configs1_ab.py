import sys, time, os
sys.path.insert(0, '/root/repo')
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0, ROOT)
import numpy as np, torch
import breakdancer_amd.api as bda
from breakdancer_amd.api import LibraryConfig, Options
from breakdancer_amd.synth import LIB_C2, make_chromosome
d = make_chromosome(length=50_000_000, seed=1)
n = len(d["pos"]); dev = torch.device("cuda", 0)
tens = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=0)
bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
bd.set_enqueue_ahead(1)
for _ in range(20): bd.run()
res = {0: [], 1: []}
for rnd in range(6):
    for m in (0, 1):
        bd.set_debug("asm_plain", m)
        for _ in range(5): bd.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): bd.run()
        torch.cuda.synchronize(); res[m].append((time.perf_counter() - t0) / 100 * 1e3)
print("configs[1], 100 steps per round, ms per step:")
for m in (0, 1): print("  asm_plain=%d  %s -> best %.4f" % (m, " ".join("%.4f" % x for x in res[m]), min(res[m])))
