import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "woq_golden.npz"))


@pytest.fixture(scope="session")
def rtn_model_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "rtn_opt125m_like.npz"))


@pytest.fixture(scope="session")
def hip():
    """The HIP device; GPU tests must run the native library, never a fallback."""
    import torch

    assert torch.cuda.is_available(), "GPU test started without a visible HIP device"
    import neural_compressor_amd  # noqa: F401  (loads libinc_mi355x.so or raises)

    return torch.device("cuda:0")
