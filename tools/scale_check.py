#!/usr/bin/env python3
"""One-off scale validation: a chromosome several times larger than configs[1] through the product and the oracle
(exercises multi-iteration tile scans, multi-block folds, larger join partitions)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import OracleRun, make_opts
from runner import compare, product_from_oracle
from breakdancer_amd.synth import make_chromosome

L = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
DISC = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02   # 0.08 at 200 Mbp puts > 4 M reads into the join: the partitioned path
t0 = time.time()
d = make_chromosome(length=L, seed=21, discordant=DISC)
n = len(d["tid"])
print("generated", n, "reads in %.1fs" % (time.time() - t0), flush=True)
cfg = "readgroup:rg1\tplatform:illumina\tmap:syn.bam\treadlen:100.00\tlib:lib1\tlower:310.00\tupper:490.00\tmean:400.00\tstd:30.00\n"
run = OracleRun(cfg, make_opts())
run.set_targets(["chrS"])
st = dict(tid=d["tid"], pos=d["pos"], mtid=d["mtid"], mpos=d["mpos"], isize=d["isize"], flag=d["flag"], qlen=d["qlen"].astype(np.int32),
          bdqual=d["mapq"], lib=np.zeros(n, np.int32), name_id=d["name_key"])
run.set_stream(0, st)
t0 = time.time()
run.run()
print("oracle %.1fs: regions %d svs %d W %d" % (time.time() - t0, run.n_regions, run.n_svs, run.W), flush=True)
bd = product_from_oracle(run)
s = compare(run, bd)
print("product == oracle:", s, bd.timings(), "device / host SVs, host groups:", bd.walk_split())
