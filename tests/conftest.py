import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def emu():
    """TEST-ONLY host emulation of the HIP kernels (tests/emu); never used by the product package."""
    from tests.emu.emu_backend import emu_backend
    return emu_backend()


@pytest.fixture(scope="session")
def hip():
    import torch
    from openp5_amd._lib import hip_backend
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return hip_backend()
