"""Static checks on the gfx950 ISA of one kernel (hipcc cross-compiles without a GPU).

The streamed CIN products (hk_bgemm.h, DEEP = 2) keep two chunks of the big operand in flight only if the compiler can
COUNT the outstanding requests: a request behind a bounds branch makes every wait `s_waitcnt vmcnt(0)` and the kernel
runs at a third of its speed.  The loaders of that path are branch-free; this test keeps them so.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _loops(lines):
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield lines[labels[m.group(1)]:i]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_streamed_cin_product_counts_its_requests(tmp_path):
    src = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'cin.hip')
    out = str(tmp_path / 'cin.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=600)
    txt = open(out).read()
    # bgemm_kernel<true, false, LdPlainV, LdPlainN, EpAffine, 2, false>: W X with W streamed (hk_cin_sci_bwd, hk_cin_sci_fwd's chain)
    m = re.search(r'^(_ZN2hk12bgemm_kernelILb1ELb0ENS_8LdPlainVENS_8LdPlainNENS_8EpAffineELi2ELb0E\w*):\s.*?\n(.*?)s_endpgm', txt,
                  re.S | re.M)
    assert m, 'kernel not found in the ISA'
    main = [seg for seg in _loops(m.group(2).split('\n'))
            if sum('v_mfma' in l for l in seg) >= 16 and sum('global_load' in l for l in seg) >= 8]
    assert main, 'main loop not found'
    loop = max(main, key=len)
    waits = [int(k) for l in loop for k in re.findall(r's_waitcnt vmcnt\((\d+)\)', l)]
    n_req = sum('global_load' in l for l in loop) // 2           # requests of one chunk (the loop body holds a chunk pair)
    assert waits, 'no vmcnt wait in the loop'
    # every wait leaves the newest chunk's requests in flight: none is vmcnt(0), none waits into the newest set
    assert min(waits) >= n_req - 2, (waits, n_req)
    assert not any('s_cbranch' in l for l in loop[1:-1]), 'a branch inside the chunk loop'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_gram_forward_leaves_in_16_byte_stores_and_never_spills(tmp_path):
    """bcnn_gram_panel_kernel (bcnn_fast.hip, hk_gram_tile.h): every element of y - direct and mirrored - leaves in a
    16-byte store (the epilogue turns each sub-tile through wave-private LDS for that), nothing goes to scratch, and the
    150 KB of panels + 8 KB of turn-tables fit the 160 KB of a CU."""
    src = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'bcnn_fast.hip')
    out = str(tmp_path / 'bcnn_fast.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=900)
    txt = open(out).read()
    seen = 0
    for m in re.finditer(r'^(_ZN2hk22bcnn_gram_panel_kernelILi(\d+)ELi(\d)ELb(\d)\w*):.*?\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        body = m.group(5)
        seen += 1
        assert 'scratch_' not in body, m.group(1)
        assert len(re.findall(r'global_store_dwordx4', body)) >= 16, m.group(1)
        # the only narrow stores: the norm / colsum / partial-norm words of the prologue and tail (not y); the centred
        # (covariance) instance also writes its channel means, four rows per wave and call site of center_panel
        assert len(re.findall(r'global_store_dword ', body)) <= (3 if m.group(4) == '0' else 16), m.group(1)
        assert int(re.search(r'group_segment_fixed_size (\d+)', body).group(1)) <= 160 * 1024, m.group(1)
    assert seen == 16                                              # 4 map sizes x (BCNN, signed sqrt, covariance centred / raw)
