#!/usr/bin/env python3
"""bench.py -- read-pairs/s of the anomalous read-pair clustering hot path on N MI355X (one rank per GPU).

Workload (BASELINE.json configs[1]): synthetic single chromosome, 50 Mbp, 30x, 2x100 bp, 1 library, ~1 % discordant
pairs -> 15 M records = 7.5 M read pairs per GPU, resident in HBM before the timed region.  A "step" is one full pass
of the hot path (bdx_run: classify -> compact -> region cut -> mate join -> pair groups -> component walk on the device
(host walk for the few large components) -> Poisson scores -> final SV table in pinned host memory) over that batch.  With N > 1 every rank owns its own chromosome (the path shards by chromosome, no data-path collective), so
scaling is weak and `value` is the aggregate over all ranks.

One JSON line on rank 0; see the task contract for the fields.  `roofline` is for the dominant kernel (K1, the
streaming classifier): algorithmic bytes = 28 B/read x reads per launch, divided by the kernel's
average duration measured with HIP events on the context's stream during the timed steps (bdx_get_timings; the kernel
is bracketed by events on every 4th step, because an event pair idles the GPU for ~10 us).  `cpu_baseline` times the CPU
oracle (a single-threaded port of the reference's path; the reference itself needs Boost and cannot be built here)
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ = 28          # 27 B SoA record read + 1 B class byte written
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)
PATH_BYTES_PER_PAIR = 57.3        # whole path, per read pair, at 1 % discordant pairs
CHROM_LEN = 50_000_000
CPU_SAMPLE_LEN = 50_000_000       # CPU baseline sample: the full configs[1] chromosome, repeated until ~12 s of CPU work


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--length", type=int, default=CHROM_LEN, help="chromosome length per GPU (default: configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="after the timed region, also measure the same steps with three contexts in flight (reported under "
                         "config.overlapped_contexts_untimed; off by default so that a profile of the default command only "
                         "holds the plain sequence of steps)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="contexts in flight per GPU (each with its own host thread and HIP stream, all reading the same resident "
                         "input); the default 1 is the plain sequence of steps the roofline figures refer to")
    return ap.parse_args()


def cpu_baseline(seed):
    """Time the oracle (oracle/libbdoracle.so, single thread) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleRun, make_opts
    from breakdancer_amd.synth import make_chromosome
    d = make_chromosome(length=CPU_SAMPLE_LEN, seed=seed)
    n = len(d["tid"])
    cfg = "readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n"
    best, spent, reps = None, 0.0, 0
    while spent < 12.0 and reps < 40:
        run = OracleRun(cfg, make_opts())
        run.set_targets(["chrS"])
        st = dict(tid=d["tid"], pos=d["pos"], mtid=d["mtid"], mpos=d["mpos"], isize=d["isize"], flag=d["flag"],
                  qlen=d["qlen"].astype(np.int32), bdqual=d["mapq"], lib=np.zeros(n, np.int32), name_id=d["name_key"])
        run.set_stream(0, st)
        t0 = time.perf_counter()
        run.L.bdo_run(run.h)
        dt = time.perf_counter() - t0
        del run
        best = dt if best is None else min(best, dt)
        spent += dt
        reps += 1
    return {"value": (n / 2) / best, "unit": "read-pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle (single-thread C++ port of the reference path, records already decoded to SoA) on %d Mbp of "
                      "the same synthetic workload = %d read pairs; best of %d runs (%.1f s of CPU work), %.2f s per run"
                      % (CPU_SAMPLE_LEN // 1000000, n // 2, reps, spent, best)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import breakdancer_amd as bda
    from breakdancer_amd.api import BATCH_FIELDS, LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome

    d = make_chromosome(length=a.length, seed=1 + rank, name_base=rank << 40)
    n = len(d["tid"])
    # inputs resident in HBM before the timed region (torch owns the memory; libbdx adopts the pointers)
    tens = {}
    for k, dt in BATCH_FIELDS:
        arr = np.ascontiguousarray(d[k], dtype=dt)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(arr.dtype)
        tens[k] = torch.from_numpy(arr.view(view) if view else arr).to(dev)
    torch.cuda.synchronize()
    bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=local)
    bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --contexts > 1: further contexts on the same resident input; the timed steps are dealt out round-robin and run
    # concurrently (what a whole-genome caller does with one context per chromosome)
    ctxs = [bd]
    for _ in range(max(1, a.contexts) - 1):
        x = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=local)
        x.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
        ctxs.append(x)
    for _ in range(a.warmup):
        for x in ctxs:
            x.run()
    barrier()
    k1_ms, stage = [], {}
    t0 = time.perf_counter()
    if len(ctxs) == 1:
        for _ in range(a.steps):
            bd.run()
            k1_ms.append(bd.timings()["classify"])
    else:
        import threading

        def drive(i):
            for _ in range(i, a.steps, len(ctxs)):
                ctxs[i].run()
                k1_ms.append(ctxs[i].timings()["classify"])
        th = [threading.Thread(target=drive, args=(i,)) for i in range(len(ctxs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    barrier()
    dt = time.perf_counter() - t0
    # per-stage device timings need HIP events between the stages, which idle the GPU: three extra, untimed steps
    bd.set_stage_timing(True)
    stage = {}
    for _ in range(3):
        bd.run()
        for k, v in bd.timings().items():
            stage[k] = stage.get(k, 0.0) + v
    bd.set_stage_timing(False)
    # supplementary figure (untimed region, N=1 only): the same steps with three contexts in flight, which is how a
    # whole-genome caller (one context per chromosome) keeps the GPU busy across the latency-bound tail of each step
    overlapped = None
    if world == 1 and len(ctxs) == 1 and a.overlap:
        import threading
        more = [bd]
        for _ in range(2):
            x = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=local)
            x.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
            more.append(x)
        for x in more:
            for _ in range(3):
                x.run()
        torch.cuda.synchronize()
        per = max(4, a.steps // 2)
        th = [threading.Thread(target=lambda x=x: [x.run() for _ in range(per)]) for x in more]
        to = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        do = time.perf_counter() - to
        overlapped = {"contexts_in_flight": len(more), "steps": per * len(more), "ms_per_step": do / (per * len(more)) * 1e3,
                      "value": (n // 2) * per * len(more) / do, "unit": "read-pairs/s"}
        for x in more[1:]:
            x.close()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    summary = bd.summary()

    if rank == 0:
        pairs = n // 2
        value = world * pairs * a.steps / dt
        k1_avg_ms = float(np.mean(k1_ms))
        achieved = ALGO_BYTES_PER_READ * n / (k1_avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if tj.get("reads_per_launch") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "anomalous read-pairs/s (end-to-end SV call)", "value": value, "unit": "read-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic single chromosome %d Mbp, 30x, 2x100 bp, 1 library, ~1%% discordant "
                                   "pairs; %d read pairs (%d records) per GPU, HBM-resident SoA" % (a.length // 1000000, pairs, n),
                       "sharding": "one chromosome per GPU, no data-path collective", "contexts_in_flight": len(ctxs), "svs_per_gpu": summary["n_svs_printed"],
                       "stage_ms_profiled_steps": {k: v / 3 for k, v in stage.items()},
                       "sv_candidates": dict(zip(("assembled_on_device", "from_host_walk", "groups_to_host_walk", "device_placed_by_order_key"),
                                                 bd.walk_split() + (bd.cross_window_svs(),)))},
            "roofline": {"bound": "hbm", "kernel": "k1_classify_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_READ * n,
                         "avg_kernel_ms": k1_avg_ms,
                         # SURVEY.md 8d's whole-path figure: 57.3 algorithmic bytes per read pair (28 B per read + 64 B per
                         # anomalous read at 1 % discordant) over the wall time of the step, not only the dominant kernel
                         "whole_path": {"algorithmic_bytes_per_read_pair": PATH_BYTES_PER_PAIR,
                                        "achieved": value / world * PATH_BYTES_PER_PAIR / 1e9,
                                        "frac": value / world * PATH_BYTES_PER_PAIR / 1e9 / HBM_PEAK_GBS}},
        }
        if overlapped:
            out["config"]["overlapped_contexts_untimed"] = overlapped
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seed=1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
