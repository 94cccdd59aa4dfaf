import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu(pkg):
    """Select cuda:0 once per session; single rank."""
    pkg.setup_gpu(0, 0)
    pkg.init_parallel(0, 1, None)
    return pkg
