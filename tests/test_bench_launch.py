"""CPU tests of bench.py's launch contract: --gpus N means N ranks, however the script is started."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks_by_itself():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [x for x in p.stdout.decode().splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry"] is True and out["n_gpus"] == 2 and out["ranks"] == [0, 1]


def test_world_size_must_match_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0
    assert b"WORLD_SIZE=2" in p.stderr


def test_missing_gpus_fail_loudly():
    """without N visible devices a real (non --dry) run must refuse, not print a 1-GPU number"""
    try:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except ImportError:
        have = 0
    want = have + 1 if have else 2
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(want), "--steps", "1", "--warmup", "0"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b"visible" in p.stderr
    assert not [x for x in p.stdout.decode().splitlines() if x.startswith("{")]
