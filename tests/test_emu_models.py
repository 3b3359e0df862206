"""CPU tier: the whole-model cases of tests/test_gpu_models.py (reference logits goldens, train steps) with the backbone
on torch-CPU and every head running the emulated kernel sources (tests/emu).  Test infrastructure only."""
import importlib.util
import os

import pytest

from emu.harness import emulated

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location('_emu_cases_test_gpu_models', os.path.join(_here, 'test_gpu_models.py'))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = 'cpu'

# device-runtime specific (RCCL reducer on a GPU stream, Trainer device placement)
_SKIP = {'test_reducer_on_gpu_single_rank_matches_plain_sgd', 'test_trainer_runs_one_synthetic_epoch'}
for _name, _obj in list(vars(_mod).items()):
    if _name.startswith('test_') and callable(_obj) and _name not in _SKIP:
        globals()[_name] = _obj


# minutes when emulated (HK_EMU_FULL=1 runs them)
_HEAVY = {'test_train_step_runs_and_updates[MPN-128]', 'test_train_step_runs_and_updates[OSMENet-224]',
          'test_osmenet_eval_matches_reference', 'test_train_step_runs_and_updates[CBCNN-128]',
          'test_logits_match_reference[MPN]', 'test_apcnn_exact_random_stream_mode'}


@pytest.fixture(autouse=True)
def _skip_heavy(request):
    if request.node.name in _HEAVY and os.environ.get('HK_EMU_FULL') != '1':
        pytest.skip('slow under emulation (HK_EMU_FULL=1 runs it)')


@pytest.fixture(autouse=True, scope='module')
def _emulated_heads():
    from emu import build_emu
    if build_emu._compiler() is None:
        pytest.skip('no clang++ to build the emulated kernels')
    with emulated():
        yield
