#!/usr/bin/env python3
"""One-off extended fuzz of the CLI's readers (not part of the default suite): two-BAM fuzz cases written as BAM files, the device-side
decode + gather merge (default and with tiny pieces / batches) against the host reader, text for text."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzzgen import make_case
from helpers import filter_cmd_lines
from test_gpu_cli_fuzz import EXE, FLAGSETS, write_case

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = n = 0
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    cfg, streams, targets = make_case(seed, n_pairs=int(rng.integers(300, 6000)))
    with tempfile.TemporaryDirectory() as td:
        write_case(td, streams, targets, rng)
        open(os.path.join(td, "cfg"), "w").write(cfg)
        args = FLAGSETS[seed % len(FLAGSETS)][0]
        texts = {}
        for label, env in (("device", {}), ("small", dict(BDX_BAM_PIECE_BYTES=str(int(rng.integers(70000, 200000))), BDX_BAM_BATCH_BLOCKS=str(int(rng.integers(1, 6))),
                                                          BDX_BAM_RING_BYTES=str(1 << 21))), ("host", dict(BDX_DECODE="host"))):
            p = subprocess.run([EXE, "-y", "-1"] + args + ["cfg"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            texts[label] = (p.returncode, filter_cmd_lines(p.stdout.decode()), p.stderr.decode()[-300:])
        n += 1
        if not (texts["device"][:2] == texts["host"][:2] == texts["small"][:2]) or texts["host"][0] != 0:
            bad += 1
            print("MISMATCH", seed, args, {k: (v[0], v[2]) for k, v in texts.items()}, flush=True)
print("seeds", a, b, "cases", n, "mismatches", bad)
