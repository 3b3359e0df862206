"""Route hawkeye_amd.functional through the CPU-emulated build of the kernel sources - for tests only.

`emulated()` is a context manager that (1) builds/loads tests/emu/_build/libhawkeye_emu.so, (2) swaps it in for the
gfx950 library inside hawkeye_amd._lib and (3) lets CPU tensors through the pointer/stream helpers of
hawkeye_amd.functional.  Outside the context the product behaviour (HIP tensors only, no CPU fallback) is untouched.
"""
import contextlib
import ctypes

import torch
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
    assert not t.is_cuda and (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)))   # dense either way
    return ctypes.c_void_p(t.data_ptr())


def _plugin_modules():
    import importlib
    return [importlib.import_module('hawkeye_amd.model.methods.' + n) for n in ('BCNN', 'CBCNN', 'MPNCOV', 'OSME')]


def _torch_wide_linear(layer, x):
    return layer(x)


def _torch_pooled_classifier(model, feats):
    return model.classifier(model.bilinear_pooling(feats))


def _unfused_pooled_classifier(model, feats):
    from hawkeye_amd.model.utils import wide_linear
    return wide_linear(model.classifier, model.bilinear_pooling(feats))


def set_wide_linear(kernel, fused=True):
    """Test lever: the plugins' classifier on the kernels (True: hawkeye_amd.model.utils.wide_linear / pooled_classifier,
    the product's only path) or on torch's nn.Linear (False: the yardstick some tests compare against).  fused=False (with
    kernel=True): BCNN's pooling and classifier as two autograd nodes (the round-4 composition) instead of the fused
    node - the pooled vector is then visible to module hooks.  Returns the previous bindings for restore_wide_linear()."""
    from hawkeye_amd.model.utils import wide_linear
    from hawkeye_amd.model.utils import pooled_classifier
    plugs = _plugin_modules()
    saved = [(getattr(m, 'wide_linear', None), getattr(m, 'pooled_classifier', None)) for m in plugs]
    for m in plugs:
        if hasattr(m, 'wide_linear'):
            m.wide_linear = wide_linear if kernel else _torch_wide_linear
        if hasattr(m, 'pooled_classifier'):
            m.pooled_classifier = (pooled_classifier if fused else _unfused_pooled_classifier) if kernel else _torch_pooled_classifier
    return saved


def restore_wide_linear(saved):
    for m, (f, g) in zip(_plugin_modules(), saved):
        if f is not None:
            m.wide_linear = f
        if g is not None:
            m.pooled_classifier = g


@contextlib.contextmanager
def emulated():
    lib = load_emu()
    saved = (_lib._lib, F.ptr, F.stream, F._on)
    _lib._lib, F.ptr, F.stream, F._on = lib, _cpu_ptr, (lambda: None), (lambda device: contextlib.nullcontext())
    # the plugins route their wide classifier through hk_linear_fwd / bwd; emulating a 262144-feature GEMM takes
    # minutes, so whole-model cases run `layer(x)` (torch CPU) here unless a test puts the real wide_linear back
    # (torch_classifier() below does the same for a GPU test that wants the library as its yardstick)
    wl_saved = set_wide_linear(False)
    try:
        yield F
    finally:
        _lib._lib, F.ptr, F.stream, F._on = saved
        restore_wide_linear(wl_saved)
