import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle, build

    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def native_lib():
    """The product library; on the GPU box it must exist (no CPU fallback)."""
    from debigulator_amd import _native as N

    if not os.path.exists(N.LIB_PATH):
        from debigulator_amd.build import build

        build()
    return N.lib()


@pytest.fixture(scope="session")
def gpu_device(native_lib):
    import torch

    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    assert native_lib.debig_hip_device_count() >= 1
    return "cuda:0"
