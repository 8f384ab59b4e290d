"""pytest configuration.

Markers
  gpu : needs a real MI355X (run by the driver with `-m gpu` on the GPU box). These tests call the product
        library libsgx_hip.so through the C ABI and compare against the CPU oracle (oracle/ + ATen CPU ops).
Everything else runs on CPU: the oracle against golden vectors / the reference, host logic, and the SAME
kernel sources compiled against the host emulation in tests/emu (kernel-logic checks without a GPU).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP GPU (MI355X); selected with -m gpu")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """Module constructors draw their initial weights from torch's global generator: seed it per test so that every test sees
    the same weights whatever ran before it (parity checks with ReLU masks are deterministic then, not order-dependent)."""
    import torch

    torch.manual_seed(1234)
    yield


def _has_gpu():
    import torch

    return torch.cuda.is_available()


@pytest.fixture(params=[pytest.param("gpu", marks=pytest.mark.gpu), "emu"])
def backend(request):
    """'gpu': tensors on cuda:0, product library.  'emu': CPU tensors, tests/emu host build of the same kernels."""
    import torch

    if request.param == "gpu":
        if not _has_gpu():
            pytest.skip("no HIP GPU")
        from super_gradients_amd import _lib

        _lib._LIB = None
        _lib._TEST_HOST_MODE = False
        yield torch.device("cuda:0")
    else:
        import emu_env

        emu_env.activate()
        try:
            yield torch.device("cpu")
        finally:
            emu_env.deactivate()


@pytest.fixture
def gpu_device():
    import torch

    if not _has_gpu():
        pytest.skip("no HIP GPU")
    from super_gradients_amd import _lib

    _lib._TEST_HOST_MODE = False
    return torch.device("cuda:0")
