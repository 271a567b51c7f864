import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle import oracle
    return oracle.backend()


@pytest.fixture(scope="session")
def hip_backend():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from i2pnet_amd import ops
    return ops.hip_backend()
