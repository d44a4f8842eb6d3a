"""CPU test: the CMake build (north_star: "host code stays C++ (CMake)") configures and builds out of tree, and the library it
produces exports exactly what include/bdx.h declares -- the same list the in-tree Makefile build is checked against."""
import os
import shutil
import subprocess

import pytest

import breakdancer_amd._lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("cmake"), reason="cmake not installed")
def test_cmake_builds_library_and_cli_out_of_tree(tmp_path):
    build = str(tmp_path / "build")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    cfg = subprocess.run(["cmake", "-S", ROOT, "-B", build] + gen, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert cfg.returncode == 0, cfg.stdout.decode()[-3000:]
    b = subprocess.run(["cmake", "--build", build, "-j", str(min(os.cpu_count() or 2, 8)), "--target", "bdx", "breakdancer-max", "bdx-inflate-check"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode()[-3000:]
    lib = os.path.join(build, "lib", "libbdx.so")
    cli = os.path.join(build, "bin", "breakdancer-max")
    assert os.path.exists(lib) and os.path.exists(cli) and os.path.exists(os.path.join(build, "bin", "bdx-inflate-check"))
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, stdout=subprocess.PIPE).stdout.decode()
    exported = set(line.split()[-1] for line in nm.splitlines() if " T bdx_" in line)
    assert exported == set(L.EXPORTS), (sorted(exported - set(L.EXPORTS)), sorted(set(L.EXPORTS) - exported))
    # nothing is written into the source tree (the Makefile's in-tree artefacts are another build)
    assert not os.path.exists(os.path.join(ROOT, "lib", "libbdx.so"))
    # the CLI links against the library of its own build and prints the reference's usage text without a GPU
    p = subprocess.run([cli], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b"breakdancer-max" in p.stdout
