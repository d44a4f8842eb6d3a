#!/usr/bin/env python3
"""A/B of run-time switches on ONE context holding a GPU's share of a genome (tools, not the product): the same records, bdx_run repeated
under each setting of bdx_set_debug in turn, several rounds interleaved.  usage: genome_ab.py [--fraction 0.125] [--rounds 3] name=value[,name=value] ..."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fraction", type=float, default=0.125)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--t", action="store_true")
    ap.add_argument("--lib", default=None, help="another build of libbdx.so (variants/...: tools/link_variant.sh), for A/Bs of builds on one box")
    ap.add_argument("settings", nargs="*", help="name=value[,name=value]; 'default' = no switch")
    a = ap.parse_args()
    if a.lib:
        import breakdancer_amd._lib as _l
        _l.LIB_PATH = os.path.abspath(a.lib)
    import torch
    import breakdancer_amd as bda
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.bamwrite import HG38_MBP, LIBS4
    from breakdancer_amd.synth import make_genome
    lengths = [int(m * 1e6 * a.fraction) for m in HG38_MBP]
    libs = [LibraryConfig(mean_insertsize=m, std_insertsize=sd, uppercutoff=m + 3 * sd, lowercutoff=m - 3 * sd, readlens=100.0, name="lib%d" % i) for i, (m, sd) in enumerate(LIBS4)]
    d = make_genome(lengths, coverage=30.0, seed=11, libs=LIBS4, lib_bam=(0, 0, 0, 0), n_translocations=5000)
    n = len(d["tid"])
    bd = bda.BreakDancer(Options(transchr_rearrange=True) if a.t else Options(), libs, 1, ntids=len(lengths), max_read_window_size=200, device=0)
    bd.lib.bdx_reserve(bd.h, n)
    bd.push_reads(d)
    torch.cuda.synchronize()
    bd.run()
    bd.set_enqueue_ahead(0)
    settings = a.settings or ["default"]
    names = sorted({kv.split("=")[0] for s in settings if s != "default" for kv in s.split(",")})
    res = {s: [] for s in settings}
    ref = None
    for rnd in range(a.rounds):
        for s in settings:
            for nm in names:
                if nm != "enqueue_ahead":
                    bd.set_debug(nm, 0)
            bd.set_enqueue_ahead(0)
            if s != "default":
                for kv in s.split(","):
                    k, v = kv.split("=")
                    if k == "enqueue_ahead":   # (not a debug switch: bdx_set_enqueue_ahead -- 1 = the later stages sized by the prior and launched while K1 runs)
                        bd.set_enqueue_ahead(int(v))
                    else:
                        bd.set_debug(k, int(v))
            bd.run()   # (settle)
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bd.run()
                ts.append(time.perf_counter() - t0)
            res[s].append(min(ts))
            svs = bd.svs()[0]
            sig = (len(svs), int(svs["pos"].astype(np.int64).sum()), int(svs["score"].astype(np.int64).sum()))
            if ref is None:
                ref = sig
            assert sig == ref, (s, sig, ref)
    print("walk split (device SVs, host SVs, groups to the host) %s; candidates placed by key (started in an earlier flush window) %d" % (bd.walk_split(), bd.cross_window_svs()))
    print("records %d; best of 6 runs per round, %d rounds interleaved; every setting gives the same table" % (n, a.rounds))
    for s in settings:
        print("  %-40s %s  -> best %.3f ms" % (s, " ".join("%.3f" % (x * 1e3) for x in res[s]), min(res[s]) * 1e3))
    bd.close()


if __name__ == "__main__":
    main()
