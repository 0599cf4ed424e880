import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no GPU is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _native_library_present():
    """The HIP library is built in-tree by __graft_entry__.build(); build it if a fresh checkout lacks it."""
    from evfly_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
