#!/usr/bin/env python3
"""One-off extended differential fuzz (not part of the default suite): many more seeds of the random small-component
graphs and of the dense multi-BAM cases through the product (device walk + host walk) and the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzzgen import GRAPH_OPTION_SETS, OPTION_SETS, clash_names, make_case, make_graph_case
from helpers import make_opts
from runner import compare, compare_support, oracle_case, product_from_oracle

a, b = int(sys.argv[1]), int(sys.argv[2])
clash = len(sys.argv) > 3 and sys.argv[3] == "clash"  # names seen three, four ... times (read-level host replay)
bad = 0
replayed = 0
for seed in range(a, b):
    for gen, sets, tag in ((make_graph_case, GRAPH_OPTION_SETS, "graph"), (make_case, OPTION_SETS, "dense"),
                           (lambda sd: make_graph_case(sd, n_slots=400, sizes=(1, 3, 5, 6, 8, 10, 14, 20, 30, 45, 66)), GRAPH_OPTION_SETS, "medium")):
        cfg, streams, targets = gen(seed)
        if clash:
            streams = clash_names(streams, seed, frac=0.005 + 0.01 * (seed % 5))
        o = sets[(seed * 5 + 1) % len(sets)]
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **o))
        try:
            bd = product_from_oracle(run, support=True)
            compare(run, bd)
            compare_support(run, bd)
            replayed += bd.was_replayed()
            bd.close()
        except Exception as e:  # noqa
            bad += 1
            print("MISMATCH", tag, seed, o, str(e)[:300], flush=True)
print("seeds", a, b, "mismatches", bad, "replayed", replayed)
