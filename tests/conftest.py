import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on 8 CPU cores")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a GPU; none is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle (torch on the host cores) is what the GPU tests compare with; on a 256-thread host torch's default
    (one thread per core) runs it several times slower than 32 threads do."""
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    yield
