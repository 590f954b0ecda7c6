import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load libivlm_hip.so; GPU tests must go through it."""
    from interactvlm_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (no CPU fallback exists)")
    return torch.device("cuda:0")
