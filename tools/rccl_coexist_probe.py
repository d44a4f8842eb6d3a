#!/usr/bin/env python3
"""libbdx's RCCL communicator next to torch.distributed's own (backend nccl) in one process, world of one: what
bench.py --gpus N does on every rank, minus the other ranks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29777")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier()
import bench
out = {}
bench.whole_genome_exchange(0, 1, 0, dist, out, chroms_per_rank=3)
print(out)
dist.destroy_process_group()
