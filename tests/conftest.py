import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _emulate_gpu_tests(request):
    """FCN_EMULATE=1: run the -m gpu tests on the CPU against the host emulation of the kernels (tests/emu_shim.py) -- a
    development aid (`FCN_EMULATE=1 python -m pytest tests -m gpu -k ...` before spending GPU minutes); the driver's GPU run
    never sets it."""
    if os.environ.get("FCN_EMULATE", "0") == "1" and request.node.get_closest_marker("gpu") is not None:
        from emu_shim import emulated_gpu
        with emulated_gpu():
            yield
    else:
        yield
