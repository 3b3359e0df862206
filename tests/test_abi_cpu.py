"""CPU-side checks of the drop-in boundary: the in-tree library loads, exports
every function include/hawkeye_hip.h declares (and nothing the ctypes table does
not know), and refuses CPU tensors loudly (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'hawkeye_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', src)))


def test_header_and_ctypes_table_agree():
    from hawkeye_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from hawkeye_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.hk_version()
    # pure host-side queries are callable without a GPU
    assert lib.hk_bcnn_pool_ws_bytes(64, 512, 196) >= 64 * 8 * 196 * 4
    assert lib.hk_cbp_plan_bytes(512, 6000) > 512 * 512 * 4
    assert lib.hk_ns_sqrtm_ws_bytes(64, 256, 5, 1) >= 9 * 64 * 256 * 256 * 4


def test_cpu_tensors_are_refused():
    import hawkeye_amd.functional as F
    from hawkeye_amd._lib import HawkeyeHipError
    with pytest.raises(HawkeyeHipError):
        F.bilinear_pool(torch.rand(1, 8, 2, 2))
    with pytest.raises(HawkeyeHipError):
        F.covpool(torch.rand(1, 8, 2, 2))


def test_a_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    """No CPU path behind the ops: without the built extension every op raises (and says how to build it)."""
    from hawkeye_amd import _lib
    import hawkeye_amd.functional as F
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libhawkeye_hip.so'))
    with pytest.raises(_lib.HawkeyeHipError, match='no CPU fallback'):
        _lib.load()
    with pytest.raises(_lib.HawkeyeHipError):
        F.bilinear_pool(torch.rand(1, 8, 2, 2))
