import os
import sys

import pytest
import torch  # noqa: F401  (before librayz_hip.so: torch carries its own HIP runtime, and the process must load only one —
#                the library's libamdhip64 dependency then resolves to the copy torch has already mapped)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the in-tree HIP library and the oracle."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import binding

    binding.load()
    return binding


@pytest.fixture(scope="session")
def gpu(built):
    """Initialise device 0 through the C ABI; a missing GPU or extension is a hard failure, not a skip."""
    from rayz_amd import render

    render.init(0)
    return render
