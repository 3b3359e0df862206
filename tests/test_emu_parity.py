"""CPU tier: the parity cases of tests/test_gpu_parity.py run against the kernel SOURCES compiled for x86 on the HIP
emulation of tests/emu (fibers for work-items, emulated wave collectives and MFMA).  Same inputs, same oracle, same
tolerances as on the GPU - so indexing, LDS staging, barrier placement, reduction orders and the host-side dispatch of
every entry point are checked here, without a GPU.  It does not replace the `-m gpu` tier (which runs the gfx950
binary) and nothing in the product imports it.
"""
import importlib.util
import os

import pytest

from emu.harness import emulated

_here = os.path.dirname(os.path.abspath(__file__))
_modules = []
for _file in ('test_gpu_parity.py', 'test_gpu_kernels.py'):
    _spec = importlib.util.spec_from_file_location('_emu_cases_' + _file[:-3], os.path.join(_here, _file))
    _mod = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_mod)
    _mod.DEV = 'cpu'
    _modules.append(_mod)

# cases that are only meaningful on the device (timing / determinism of the real scheduler) or too slow emulated
_SKIP = {'test_bcnn_full_size_properties', 'test_bcnn_full_size_backward_vs_generic'}   # B=64 full-size: minutes when emulated

for _mod in _modules:
    for _name, _obj in list(vars(_mod).items()):
        if _name.startswith('test_') and callable(_obj) and _name not in _SKIP:
            globals()[_name] = _obj


# parameter sets that take 20 s .. 80 s each when emulated: run them with HK_EMU_FULL=1
_HEAVY = {'test_cbp_rowsketch_equals_csr[512-6000-40]', 'test_models_with_hip_classifier[BCNN]',
          'test_models_with_hip_classifier[MPN]', 'test_cin_model_matches_reference', 'test_cin_model_at_448_input_matches_reference',
          'test_bcnn_backward_in_one_launch[2-128-14-7-True]', 'test_signed_sqrt_norm_formed_in_the_classifier_reduce[5-512-14-7]',
          'test_linear_bwd_single_products[33-16384-200]',
          'test_ns_grouped_products[2-256-2]', 'test_ns_grouped_products[3-200-3]',
          'test_cov_and_cbp_panel_kernels_vs_generic[70-256-8]', 'test_mpn_256_vs_golden',
          'test_linear_bwd_direct_at_classifier_shapes[3-262144-200]', 'test_bcnn_signed_sqrt_512_vs_golden',
          'test_linear_bwd_direct_at_classifier_shapes[16-6000-8142]', 'test_ns_symmetric_forward[17-256-5]',
          'test_sqrtm_triuvec_in_one_chain[17-256-5]',
          'test_backward_bwd3_kernel[2-512-14]', 'test_bcnn_backward_in_one_launch[2-512-14-200-True]',
          'test_bcnn_backward_in_one_launch[3-256-10-20-True]', 'test_bcnn_backward_in_one_launch[2-256-8-13-True]',
          'test_bcnn_backward_in_one_launch[3-192-12-5-True]',
          'test_linear_bwd_direct_at_classifier_shapes[64-65536-200]', 'test_linear_bwd_direct_at_classifier_shapes[37-65728-130]',
          'test_linear_bwd_single_products[9-16384-260]', 'test_linear_bwd_single_products[64-16384-193]',
          'test_signed_sqrt_pool_with_the_scale_folded_into_the_classifier[5-192-8-208]', 'test_ns_symmetric_forward[1-384-2]', 'test_linear_bwd_direct_at_classifier_shapes[10-100352-1024]',
          'test_linear_bwd_single_products[5-16448-208]', 'test_ns_two_queue_dispatch_bit_identical[16-64]',
          'test_linear_bwd_direct_at_classifier_shapes[16-20032-1000]', 'test_ns_symmetric_forward[2-200-3]',
          'test_signed_sqrt_pool_with_the_scale_folded_into_the_classifier[3-128-14-200]',
          'test_linear_bwd_direct_at_classifier_shapes[2-32896-200]', 'test_linear_bwd_direct_at_classifier_shapes[4-32768-200]',
          'test_cin_channel_interaction_ops[2-2048-49]', 'test_cin_channel_interaction_ops[2-512-49]',
          'test_cin_channel_interaction_ops[4-1024-64]', 'test_cin_channel_interaction_ops[2-2048-196]',
          'test_cin_channel_interaction_ops[2-1024-144]', 'test_cin_module_at_plugin_width_matches_reference',
          'test_linear_bwd_direct_at_classifier_shapes[8-32896-200]'}


@pytest.fixture(autouse=True)
def _skip_heavy(request):
    if request.node.name in _HEAVY and os.environ.get('HK_EMU_FULL') != '1':
        pytest.skip('slow under emulation (HK_EMU_FULL=1 runs it)')


@pytest.fixture(scope='module')
def F():
    from emu import build_emu
    if build_emu._compiler() is None:
        pytest.skip('no clang++ to build the emulated kernels (they use ext_vector_type)')
    with emulated() as f:
        yield f
