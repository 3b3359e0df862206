import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


from hawkeye_amd.miopen_cache import use_in_tree_cache  # noqa: E402

use_in_tree_cache()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import torch
    if torch.get_num_threads() > 16:       # a 256-core host: the CPU tier's small oracle ops are slower on all cores
        torch.set_num_threads(16)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def tune():
    """`tune(name, value)` flips an A/B knob of the loaded library (hk_tuning_set: the gfx950 build on the GPU tier, the
    emulated build inside tests/emu's context) and restores every touched knob when the test ends."""
    import ctypes

    from hawkeye_amd import _lib
    saved = {}

    def set_(name, value):
        lib = _lib.load()
        if name not in saved:
            old = ctypes.c_int(0)
            assert lib.hk_tuning_get(name.encode(), ctypes.byref(old)) == 0, name
            saved[name] = (lib, old.value)
        assert lib.hk_tuning_set(name.encode(), int(value)) == 0, name

    yield set_
    for name, (lib, value) in saved.items():
        lib.hk_tuning_set(name.encode(), value)
