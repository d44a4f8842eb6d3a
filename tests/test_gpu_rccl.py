"""The sharded run over REAL RCCL ranks: one process per GPU (torch.distributed.run), communicator from ncclCommInitRank,
ncclAllReduce / ncclAllToAllv / grouped send-receive between devices.  Needs at least two GPUs: on the one-GPU test boxes
it is skipped (RCCL refuses two ranks on one device; tests/test_gpu_sharded.py runs the same orchestration with the ranks as
threads there), on any multi-GPU lease it runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    # (asked of the HIP runtime this process already has -- libbdx's; importing torch here would bring its bundled copy of the
    # runtime and of RCCL into a process that may have created a communicator with the system's: two runtimes, one exit)
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        n = C.c_int(0)
        return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("nproc", [1, 2, 4])   # (1: the same worker with a communicator of one rank, on any GPU box)
def test_sharded_run_over_rccl_ranks_equals_oracle(nproc, tmp_path):
    if _devices() < nproc:
        pytest.skip("needs %d GPUs, %d visible" % (nproc, _devices()))
    out = str(tmp_path / "out.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + nproc), os.path.join(ROOT, "tests", "rccl_worker.py"), out]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    r = json.load(open(out))
    assert r["ok"] and r["world"] == nproc and len(r["cases"]) == 4
    assert r["cases"][-1]["replayed"] and not r["cases"][0]["replayed"]
