#!/usr/bin/env python3
"""Raw host-to-device bandwidth of the box (pinned memory, hipMemcpyAsync through torch): the ceiling of timing (ii)."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (1, 4, 15, 60, 375):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(3, 600 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%4d MB per copy: %.3f ms, %.1f GB/s" % (mb, dt * 1e3, (mb << 20) / dt / 1e9))
# nine copies of the column sizes of configs[1] (what bdx_push issues for 15 M records: 5 x 60 MB, 30 MB, 3 x 15 MB)
sizes = [60, 60, 60, 60, 60, 30, 15, 15, 15]
hs = [torch.empty(s << 20, dtype=torch.uint8).pin_memory() for s in sizes]
ds = [torch.empty(s << 20, dtype=torch.uint8, device=dev) for s in sizes]
for _ in range(2):
    for h, d in zip(hs, ds):
        d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for h, d in zip(hs, ds):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("the nine column copies of 15 M records (375 MB): %.2f ms, %.1f GB/s" % (best * 1e3, sum(sizes) * (1 << 20) / best / 1e9))
