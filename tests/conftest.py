import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # no test may sit on the GPU box for ever (a kernel that never ends, ranks that wait for each other): pytest-timeout where it is installed
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


@pytest.fixture(scope="session", autouse=True)
def _built_artefacts():
    """libbdx.so, the CLI tools and the oracle are build products (git-ignored): build them when a fresh checkout runs the tests"""
    need = [os.path.join(ROOT, "breakdancer_amd", "libbdx.so"), os.path.join(ROOT, "bin", "breakdancer-max"),
            os.path.join(ROOT, "bin", "bdx-dump-reads"), os.path.join(ROOT, "oracle", "libbdoracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()
    yield
