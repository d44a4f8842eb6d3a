"""Multi-process tests of the sharded path.  CPU: world_size 2 over gloo (collectives + routing).  GPU box: the real
staged path with two processes sharing cuda:0 (control plane over gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(mode, nproc, out, port):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER, mode, out]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("nproc", [2, 3])
def test_collectives_and_routing_over_gloo(tmp_path, nproc):
    r = launch("comm", nproc, str(tmp_path / "out.json"), 29600 + nproc)
    assert r["ok"] and r["world"] == nproc and r["groups"] > 0


def test_plan_and_helpers():
    import numpy as np
    from breakdancer_amd import shard
    hg38 = dict(enumerate([248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]))
    plan = shard.plan_chromosomes(hg38, 8)
    loads = [sum(hg38[t] for t in b) for b in plan]
    assert sorted(t for b in plan for t in b) == list(range(24))
    assert max(loads) <= 1.06 * sum(hg38.values()) / 8           # LPT: a few percent imbalance on the primary contigs
    own = shard.owner_of(np.arange(1, 100000, dtype=np.uint64), 8)
    assert np.bincount(own, minlength=8).min() > 11000           # balanced owners
    # uint32 covered vs size_t sums (BamSummary.cpp:123-126): 5 -> (2^32+7 truncated) 7 -> 9
    assert shard.covered_from(np.array([5, 2**32 + 7, 9], dtype=np.uint64)) == 9


@pytest.mark.gpu
def test_two_processes_staged_path_on_one_gpu(tmp_path):
    r = launch("gpu", 2, str(tmp_path / "out.json"), 29650)
    assert r["ok"] and r["world"] == 2 and len(r["cases"]) == 3
