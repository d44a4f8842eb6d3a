"""GPU tests of bench.py's contract.  N > 1 control flow on a one-GPU box: two ranks share device 0 behind the test hook
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
                        "--genome-fraction", "0.004", "--sharded-cli-fraction", "0.002"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    # N > 1: rank 0 runs the one command whose ranks decode their chromosomes of ONE indexed BAM on their GPUs (here: both on device 0)
    sh = out["config"]["timings"]["bam_to_table_sharded"]
    assert "error" not in sh, sh
    assert sh["every_rank_decoded_on_its_gpu"] and sh["same_table_as_one_gpu"] and sh["sv_rows"] > 0 and sh["BDX_GPUS"] == "0,0"
    assert "error" not in sh["two_files"], sh["two_files"]
    assert sh["two_files"]["every_rank_decoded_on_its_gpu"] and sh["two_files"]["same_table_as_one_gpu"] and sh["two_files"]["sv_rows"] > 0
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["scaling"] == "weak"
    assert "cpu_baseline" not in out and "test_hook" in out["config"]
    g = out["config"]["genome"]
    assert "error" not in g, g
    # N > 1: the line's value is the sharded whole-genome run's -- K timed bdx_dist_run steps over all ranks -- and the N replicas of configs[1] are a note
    ts = g["default_options"]["timed_steps"]
    assert ts["steps"] == 8 and out["value"] == ts["value"] and abs(out["ms_per_step"] - ts["ms_per_step"]) < 1e-9
    assert abs(out["value"] - g["reads"] / 2 * 8 / ts["seconds"]) < 1e-6 * out["value"] and out["config"]["workload"].startswith("configs[2]")
    pairs = 6_000_000 * 30 // 200
    rep = out["config"]["per_rank_replicas"]
    assert abs(rep["value"] - 2 * pairs / (rep["ms_per_step"] * 1e-3)) < 1e-3 * rep["value"]
    # the weak-scaling size (hg38 x N/8 unless --genome-fraction says otherwise) ...
    assert g["ranks"] == 2 and g["scaling"] == "weak" and len(g["reads_per_rank"]) == 2 and sum(g["reads_per_rank"]) == g["reads"]
    assert 1.0 <= g["lpt_imbalance_max_over_mean"] < 1.2
    for leg in ("default_options", "t_option"):
        x = g[leg]
        assert x["seconds"] > 0 and x["second_run_seconds"] > 0 and x["svs_printed"] > 0 and len(x["bdx_dist_run_ms_per_rank"]) == 2
        assert "rank0_only" in x and 0 <= x["rank0_only"]["share_of_run"] <= 1 and "rank0_only_merge" in x["rank0_phase_ms"]
        assert sum(x["sv_candidates_device_host"]) > 0
    assert g["t_option"]["ctx_records_exchanged"] > 100   # (inter-chromosomal reads whose mates lie on a later chromosome of the other rank)


def test_one_rank_line_carries_the_end_to_end_ratio_and_the_single_context_figure():
    """N = 1: vs_baseline = BAM -> table over the CPU baseline (like for like), the bam_to_table block's own figure is the one-process
    run, and the genome leg holds the single-context figure beside the sharded run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--length", "4000000", "--genome-fraction", "0.004",
                        "--no-pmc", "--no-overlap", "--cpu-parallel", "0", "--sharded-cli-fraction", "0.002", "--genome-bam-fraction", "0.004"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([x for x in p.stdout.decode().splitlines() if x.startswith("{")][0])
    e2e = out["config"]["timings"]["bam_to_table"]
    assert "error" not in e2e, e2e
    assert e2e["seconds"] >= e2e["command_return"]["seconds"] * 0.5 and "exit included" in e2e["reader"]
    assert abs(out["vs_baseline"] - e2e["value"] / out["cpu_baseline"]["value"]) < 1e-9 * out["vs_baseline"] and "vs_baseline_is" in out
    assert out["config"]["timings"]["hbm_resident"]["over_cpu_compute_only"] > 1
    sh = out["config"]["timings"]["bam_to_table_sharded"]
    assert "error" not in sh, sh
    assert sh["every_rank_decoded_on_its_gpu"] and sh["same_table_as_one_gpu"] and sh["sv_rows"] > 0 and sh["seconds"] > 0
    # BAM -> table at (here: a sliver of) a GPU's share of a genome, with the ceilings measured beside it and the CPU path on a slice of the same genome
    gb = out["config"]["timings"]["bam_to_table_genome"]
    assert "error" not in gb, gb
    assert gb["sv_rows"] > 0 and gb["records"] > 3_000_000 and gb["steady_state_gb_s"] > 0 and gb["file_gb_per_s"] > 0
    assert gb["ceilings"]["feed"]["both_pipelined_gb_s"] > 1 and gb["ceilings"]["inflate_kernel_alone"]["ok"] and gb["ceilings"]["inflate_kernel_alone"]["file_gb_s"] > 1
    assert gb["cpu_port_on_a_slice"]["value"] > 0 and gb["over_cpu_port"] > 1
    g = out["config"]["genome"]
    assert "error" not in g, g
    assert g["ranks"] == 1 and g["single_context"]["seconds"] > 0 and g["default_options"]["svs_printed"] == g["single_context"]["svs_printed"]
    assert out["roofline"]["kernel"] == "k1_classify_kernel" and 0 < out["roofline"]["frac"] < 1
