#!/usr/bin/env python3
"""One GPU's share of a whole genome (configs[2] / configs[3] shape: hg38 lengths x fraction, 4 libraries, 30x) through
  (a) ONE context (bdx_run over all chromosomes at once: the single-context figure), first run and repeated runs with stage timings,
  (b) the sharded run (bdx_dist_*, one rank, RCCL backend or threads) with its phase timings,
and checks that both give the same SV table.  Usage: genome_probe.py [--fraction 0.125] [--ranks 1] [--t] [--repeat 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HG38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28, 114.36, 107.04,
            101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]
LIBS4 = ((400.0, 30.0), (350.0, 40.0), (500.0, 50.0), (300.0, 25.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fraction", type=float, default=0.125)
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--t", action="store_true")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--translocations", type=int, default=5000)
    ap.add_argument("--skip-single", action="store_true")
    a = ap.parse_args()
    import torch
    import breakdancer_amd as bda
    from breakdancer_amd import dist as D
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import make_genome
    lengths = [int(m * 1e6 * a.fraction) for m in HG38_MBP]
    ntids = len(lengths)
    libs = [LibraryConfig(mean_insertsize=m, std_insertsize=sd, uppercutoff=m + 3 * sd, lowercutoff=m - 3 * sd, readlens=100.0, name="lib%d" % i)
            for i, (m, sd) in enumerate(LIBS4)]
    t0 = time.perf_counter()
    d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=a.translocations)
    n = len(d["tid"])
    print("synthesised %d records in %.1f s" % (n, time.perf_counter() - t0), flush=True)
    opts = Options(transchr_rearrange=True) if a.t else Options()
    out = {"reads": n, "t_option": bool(a.t)}
    ref = None
    if not a.skip_single:
        bd = bda.BreakDancer(opts, libs, 1, ntids=ntids, max_read_window_size=200, device=0)
        bd.lib.bdx_reserve(bd.h, n)
        bd.push_reads(d)
        torch.cuda.synchronize()
        runs = []
        for i in range(1 + a.repeat):
            if i == 1:
                bd.set_enqueue_ahead(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bd.run()
            dt = time.perf_counter() - t0
            runs.append({"seconds": dt, "stage_ms": {k: round(v, 3) for k, v in bd.timings().items()}})
        sm = bd.summary()
        ref = bd.svs()
        out["single_context"] = {"first_run_seconds": runs[0]["seconds"], "best_seconds": min(r["seconds"] for r in runs[1:]) if a.repeat else None,
                                 "value_read_pairs_per_s": n / 2 / min(r["seconds"] for r in runs), "runs": runs,
                                 "regions": sm["n_regions"], "svs": sm["n_svs"], "svs_printed": sm["n_svs_printed"], "anomalous": sm["n_anomalous"],
                                 "walk_split": bd.walk_split()}
        print(json.dumps(out["single_context"]), flush=True)
        bd.close()
    # the sharded run
    world = a.ranks
    rank_of = D.plan(lengths, world)
    if world == 1:
        ranks = [D.DistRun.create(opts, libs, 1, ntids, 200, 0, 0, 1, D.unique_id())]
    else:
        ranks = D.DistRun.threads(opts, libs, 1, ntids, 200, [0] * world)
    tid = d["tid"]
    bounds = np.searchsorted(tid, np.arange(ntids + 1))
    for t in range(ntids):
        lo, hi = int(bounds[t]), int(bounds[t + 1])
        if hi > lo:
            ranks[rank_of[t]].chromosome(t).push_reads({k: v[lo:hi] for k, v in d.items()})
    if hasattr(ranks[0], "prepare"):
        for r in ranks:
            r.prepare()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world == 1:
        ranks[0].run(release=False)
        res = ranks[0].result()
    else:
        res = D.run_threads(ranks)
    dt = time.perf_counter() - t0
    sm = res.summary()
    first_phases = [r.phases() for r in ranks]
    t0 = time.perf_counter()   # the same handles again (every buffer is there, every kernel has run once)
    if world == 1:
        ranks[0].run(release=False)
        res = ranks[0].result()
    else:
        res = D.run_threads(ranks)
    dt2 = time.perf_counter() - t0
    out["sharded"] = {"ranks": world, "seconds": dt, "second_run_seconds": dt2, "first_run_phase_ms": first_phases, "value_read_pairs_per_s": n / 2 / dt, "regions": sm["n_regions"], "svs": sm["n_svs"],
                      "svs_printed": sm["n_svs_printed"], "exchange": [r.exchange() for r in ranks], "phase_ms": [r.phases() for r in ranks],
                      "walk_split": res.walk_split()}
    print(json.dumps(out["sharded"]), flush=True)
    if ref is not None:
        svs, (li, lp), (ck, cv) = res.svs()
        rs, (rli, rlp), (rck, rcv) = ref
        same = len(svs) == len(rs) and all(np.array_equal(svs[f], rs[f]) for f in ("chr", "pos", "flag", "size", "score", "num_reads", "printed")) \
            and np.array_equal(li, rli) and np.array_equal(lp, rlp) and np.array_equal(ck, rck) and np.array_equal(cv.view(np.uint32), rcv.view(np.uint32))
        out["sharded_equals_single_context"] = bool(same)
        print("sharded == single context:", same, flush=True)
    for r in ranks:
        r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
