import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libinc_mi355x.so (built artefacts are not in git): build it once, exactly as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU), so that the symbol / host-logic tests can
    load the real library.  There is no fallback to fall back to."""
    lib = os.path.join(ROOT, "neural_compressor_amd", "libinc_mi355x.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "neural_compressor_amd", "csrc"), "-j", str(os.cpu_count() or 4)], check=True)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "woq_golden.npz"))


@pytest.fixture(scope="session")
def golden_bits():
    """INCWeightOnlyLinear at the widths other than 2 / 4 / 8, from the unmodified reference (tests/golden/make_golden_bits.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "woq_bits_golden.npz"))


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
