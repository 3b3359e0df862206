"""Error behaviour of the C ABI (include/hawkeye_hip.h): argument validation happens before anything touches the GPU, so
it can be exercised on a box without one.  Null pointers / zero sizes -> HK_ERR_BAD_ARG from every compute entry point; a
workspace smaller than the matching *_ws_bytes() -> HK_ERR_WORKSPACE (nothing launched, nothing written past a short
buffer).  Runs in a child process: a missing check would be a segmentation fault, which must fail one test, not the run."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import ctypes, sys
sys.path.insert(0, sys.argv[1])
from hawkeye_amd import _lib
lib = _lib.load()
mode = sys.argv[2]
# the forward's workspace is optional by contract (a short one selects the one-launch column sums): not in the ws table
WS_OPTIONAL = {'hk_bcnn_pool_fwd', 'hk_bcnn_colsum_norm'}
for name, (res, args) in _lib.SIGNATURES.items():
    if res is not _lib.c_i or name in ('hk_tuning_set', 'hk_tuning_get'):
        continue
    if mode == 'ws' and (_lib.c_sz not in args or name in WS_OPTIONAL):
        continue
    vals = []
    for a in args:
        if a is _lib.c_f:
            vals.append(None if mode == 'null' else ctypes.c_void_p(0x100000))     # never dereferenced on the host
        elif a is _lib.c_fl:
            vals.append(0.0 if mode == 'null' else 1.0)
        elif a is _lib.c_sz:
            vals.append(0)
        else:
            vals.append(0 if mode == 'null' else 64)
    if mode == 'ws':
        vals[-1] = None                                   # stream: the default queue
    print('CALL', name, flush=True)
    print('RC', name, getattr(lib, name)(*vals), flush=True)
print('DONE', flush=True)
'''


def _run(mode):
    p = subprocess.run([sys.executable, '-c', _CHILD, ROOT, mode], capture_output=True, text=True, timeout=300)
    lines = p.stdout.splitlines()
    assert lines and lines[-1] == 'DONE', f'child died after: {lines[-1] if lines else "<nothing>"}\n{p.stderr[-1500:]}'
    return {ln.split()[1]: int(ln.split()[2]) for ln in lines if ln.startswith('RC ')}


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip.so')), reason='library not built')
def test_null_and_zero_arguments_are_rejected_before_any_launch():
    from hawkeye_amd._lib import HK_ERR_BAD_ARG
    rc = _run('null')
    assert len(rc) >= 46
    assert {n: c for n, c in rc.items() if c != HK_ERR_BAD_ARG} == {}


@pytest.mark.skipif(torch.cuda.is_available(), reason='placeholder pointers must never reach a real GPU')
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip.so')), reason='library not built')
def test_short_workspace_is_rejected_before_any_launch():
    from hawkeye_amd._lib import HK_ERR_WORKSPACE
    rc = _run('ws')
    assert len(rc) >= 16
    assert {n: c for n, c in rc.items() if c != HK_ERR_WORKSPACE} == {}
