import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")
    # torch's OpenMP team follows the machine (128 threads), the container's CPU quota is 16: every parallel CPU op of a test (the float64
    # torch references, scene construction) would get the whole process suspended for the rest of a scheduler period (moss_amd/host.py)
    from moss_amd.host import limit_cpu_threads
    limit_cpu_threads()


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import oracle
    oracle.build()


@pytest.fixture(scope="session")
def hip_lib():
    """The HIP library must exist (built by __graft_entry__.build() / python -m moss_amd.build). No fallback."""
    from moss_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from moss_amd import build
        build.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
