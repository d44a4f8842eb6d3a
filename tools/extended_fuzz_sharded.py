#!/usr/bin/env python3
"""One-off extended fuzz of the sharded run (not part of the default suite): many seeds through 1-4 ranks (threads sharing one GPU) with colliding
name keys + second hash, clashing names, and the supporting reads, against ONE oracle run each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, clash_names, make_case, make_graph_case
from helpers import make_opts
from runner import compare, compare_support, oracle_case, sharded_from_oracle

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = n = replayed = 0
for seed in range(a, b):
    gen, sets = (make_case, OPTION_SETS) if seed % 2 == 0 else (make_graph_case, GRAPH_OPTION_SETS)
    cfg, streams, targets = gen(seed)
    if seed % 3 == 0:
        streams = clash_names(streams, seed, frac=0.01 + 0.01 * (seed % 4))
    for i, o in enumerate((sets[(seed * 5 + 1) % len(sets)], dict(transchr_rearrange=1, min_read_pair=1))):
        if o.get("min_len", 0) < 0:
            continue
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        support = (seed + i) % 4 == 0
        try:
            util = sharded_from_oracle(run, world=1 + (seed + i) % 4, collide=(0, 2, 5)[(seed + i) % 3], support=support)
            compare(run, util, check_cls=False)
            if support:
                compare_support(run, util)
            replayed += util.was_replayed()
            n += 1
        except Exception as e:  # noqa
            bad += 1
            print("MISMATCH", seed, o, str(e)[:300], flush=True)
print("seeds", a, b, "runs", n, "mismatches", bad, "replayed", replayed)
