"""Route hawkeye_amd.functional through the CPU-emulated build of the kernel sources - for tests only.

`emulated()` is a context manager that (1) builds/loads tests/emu/_build/libhawkeye_emu.so, (2) swaps it in for the
gfx950 library inside hawkeye_amd._lib and (3) lets CPU tensors through the pointer/stream helpers of
hawkeye_amd.functional.  Outside the context the product behaviour (HIP tensors only, no CPU fallback) is untouched.
"""
import contextlib
import ctypes
import os


from hawkeye_amd import _lib
import hawkeye_amd.functional as F

from . import build_emu

_emu = None


def load_emu():
    global _emu
    if _emu is None:
        lib = ctypes.CDLL(build_emu.build())
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _emu = lib
    return _emu


def _cpu_ptr(t):
    if t is None:
        return None
    assert not t.is_cuda and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


@contextlib.contextmanager
def emulated():
    lib = load_emu()
    saved = (_lib._lib, F.ptr, F.stream, F._on)
    _lib._lib, F.ptr, F.stream, F._on = lib, _cpu_ptr, (lambda: None), (lambda device: contextlib.nullcontext())
    # the plugins route their wide classifier through hk_linear_fwd by default; emulating a 262144-feature split-K
    # GEMM takes minutes, so whole-model cases keep nn.Linear here unless a test asks for the kernel explicitly
    had = os.environ.get('HAWKEYE_HIP_LINEAR')
    if had is None:
        os.environ['HAWKEYE_HIP_LINEAR'] = '0'
    # the CIN forward takes the library GEMMs for the shapes its one-kernel form does not cover; on the emulated device
    # every shape goes through the kernels (there is no library to take, and the chain stays covered)
    cin_saved = F._CIN_SCI_FWD_HIP
    F._CIN_SCI_FWD_HIP = True
    try:
        yield F
    finally:
        _lib._lib, F.ptr, F.stream, F._on = saved
        F._CIN_SCI_FWD_HIP = cin_saved
        if had is None:
            os.environ.pop('HAWKEYE_HIP_LINEAR', None)
