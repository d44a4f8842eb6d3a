"""The command's process structure (host/main.cpp): the GPU work runs in a child, the process that was started returns with the
child's status as soon as the child reports its output complete -- or, if the child ends without a report (usage errors leave
through exit() inside the option parser), with the child's own exit status.  BDX_FOREGROUND=1 keeps everything in one process.
These paths need no GPU: every one of them fails before or at bdx_create."""
import os
import subprocess

import pytest

from helpers import GOLDEN, ROOT

EXE = os.path.join(ROOT, "bin", "breakdancer-max")


def run(args, env=None, cwd=None):
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([EXE] + args, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


@pytest.mark.parametrize("env", [{}, {"BDX_FOREGROUND": "1"}])
def test_usage_and_option_errors_keep_their_exit_status(env):
    p = run([], env)
    assert p.returncode == 1 and b"Usage: breakdancer-max <analysis.config>" in p.stderr and p.stdout == b""
    p = run(["-Z", "x"], env)
    assert p.returncode == 1


@pytest.mark.parametrize("env", [{}, {"BDX_FOREGROUND": "1"}])
def test_missing_configuration_is_reported_by_the_started_process(env, tmp_path):
    p = run([str(tmp_path / "nothing.cfg")], env)
    assert p.returncode == 1
    assert b"unable to open config file" in p.stderr


def test_no_gpu_means_exit_status_one_and_a_message():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    p = run(["inv_del_bam_config"], cwd=os.path.join(GOLDEN, "chr21"))
    assert p.returncode == 1 and b"ERROR: bdx_create" in p.stderr and not [l for l in p.stdout.splitlines() if not l.startswith(b"#")]


def test_the_command_as_pid_1_of_a_container(tmp_path):
    """the started process can legitimately be PID 1 (a container's entry point): the child must not take its parent for gone"""
    import shutil
    if not shutil.which("unshare"):
        pytest.skip("no unshare")
    probe = subprocess.run(["unshare", "-pf", "true"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if probe.returncode != 0:
        pytest.skip("unshare -pf not permitted here")
    p = subprocess.run(["unshare", "-pf", EXE, str(tmp_path / "nothing.cfg")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 1 and b"unable to open config file" in p.stderr
