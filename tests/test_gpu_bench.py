"""GPU test of bench.py's N > 1 control flow on a one-GPU box: two ranks share device 0 behind the test hook
BDX_BENCH_TEST_SHARED_GPU (gloo process group).  Checks the contract of the JSON line -- n_gpus, aggregate value, the
sharded whole-genome leg present with both its runs (here the two ranks are threads of rank 0's process on the one device:
RCCL refuses two ranks on one device) -- not any number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_print_one_aggregate_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BDX_BENCH_TEST_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--length", "6000000",
                        "--genome-fraction", "0.004"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["scaling"] == "weak"
    pairs = 6_000_000 * 30 // 200
    assert abs(out["value"] - 2 * pairs / (out["ms_per_step"] * 1e-3)) < 1e-3 * out["value"]
    assert "cpu_baseline" not in out and "test_hook" in out["config"]
    g = out["config"]["genome"]
    assert "error" not in g, g
    assert g["ranks"] == 2 and g["scaling"] == "strong" and len(g["reads_per_rank"]) == 2 and sum(g["reads_per_rank"]) == g["reads"]
    assert 1.0 <= g["lpt_imbalance_max_over_mean"] < 1.2
    for leg in ("default_options", "t_option"):
        assert g[leg]["seconds"] > 0 and g[leg]["svs_printed"] > 0 and len(g[leg]["bdx_dist_run_ms_per_rank"]) == 2
    assert g["t_option"]["ctx_records_exchanged"] > 1000
