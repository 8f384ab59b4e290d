"""TEST INFRASTRUCTURE ONLY: routes super_gradients_amd._lib to the host emulation build so kernel logic
and the Python glue can be exercised on CPU tensors.  Imported only by `-m "not gpu"` tests."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def activate():
    import build_emu
    from super_gradients_amd import _lib

    so = build_emu.build()
    _lib._LIB = _lib.bind(ctypes.CDLL(so))
    _lib._TEST_HOST_MODE = True
    _clear_caches()
    return _lib


def _clear_caches():
    import sys as _sys

    k = _sys.modules.get("super_gradients_amd.kernels")
    if k is not None:
        k.clear_caches()


def deactivate():
    from super_gradients_amd import _lib

    _lib._LIB = None
    _lib._TEST_HOST_MODE = False
    _clear_caches()
