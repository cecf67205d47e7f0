import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_bytes(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def golden():
    return golden_bytes


def enable_hooks(lib=None):
    """The library reads its NAFGPU_* experiment variables only after nafgpu_test_hooks(1) (include/nafgpu.h)."""
    from nafcodec_amd import _ffi
    (lib or _ffi.default()).c.nafgpu_test_hooks(1)
