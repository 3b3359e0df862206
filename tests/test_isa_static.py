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
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'hawkeye_amd', 'csrc')   # hk_common.h includes <hk_isa.h>


def _loops(lines):
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield lines[labels[m.group(1)]:i]


def _serialised_staging_loops(txt):
    """Kernels whose ISA holds a small loop of  global_load - s_waitcnt vmcnt(0) - ds_write : a global -> LDS copy written as
    `for (f = tid; f < n; f += 256) dst[f] = src[f]` that the compiler kept as a loop, one memory round trip per iteration.
    (Rounds 3-4 shipped three CIN kernels with that at the top of every column block; found by reading the ISA in round 5.)"""
    bad = []
    for m in re.finditer(r'^(_ZN2hk\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        for seg in _loops(m.group(2).split('\n')):
            nested = re.search(r'Depth=([2-9])', ' '.join(seg[:3])) is not None   # inside another loop (a once-per-workgroup prologue copy is fine)
            if nested and len(seg) <= 24 and any('global_load' in l for l in seg) and any('vmcnt(0)' in l for l in seg) and any('ds_write' in l for l in seg):
                bad.append(m.group(1))
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_streamed_cin_product_counts_its_requests(tmp_path):
    src = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'cin.hip')
    out = str(tmp_path / 'cin.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + CSRC, '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=600)
    txt = open(out).read()
    assert not _serialised_staging_loops(txt), _serialised_staging_loops(txt)
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


    # cin_ax_kernel<HW, MODE> (the 14 x 14 / 12 x 12 / 10 x 10 SCI forward and the backward's two big products): the pieces of
    # the C x C matrix travel in registers the compiler keeps no books on (HK_LOAD16_ASYNC / HK_LOAD4_ASYNC) and X_{J+2} by
    # LDS-DMA.  What makes that correct is (a) the counted wait at the end of every step - vmcnt(NPW) leaves only that step's
    # NPW LDS-DMA requests in flight, vmcnt(NPW + 2 NL) behind the body's first step also the 2 NL piece loads it issued -
    # (b) no compiler-inserted vmcnt wait inside the loop (a vmcnt(0) would drain the pipeline), and (c) NO instruction
    # touching a register between its request and the wait that covers it: walked here, operand by operand.
    for hw, npw in ((196, 13), (144, 9), (100, 7)):
        for mode, nl in ((0, 2), (1, 8), (2, 10)):
            m = re.search(r'^(_ZN2hk13cin_ax_kernelILi%dELi%dE\w*):\s.*?\n(.*?)s_endpgm(.*?)\.end_amdhsa_kernel' % (hw, mode), txt, re.S | re.M)
            assert m, f'cin_ax_kernel<{hw}, {mode}> not found in the ISA'
            assert re.search(r'\.amdhsa_private_segment_fixed_size 0\b', m.group(3)), f'<{hw}, {mode}> uses scratch'
            body = m.group(2).split('\n')
            loop_start = next(i for i, l in enumerate(body) if re.search(r's_waitcnt vmcnt\(0\)', l))
            loop_end = [i for i, l in enumerate(body) if i > loop_start and re.search(r's_waitcnt.*?vmcnt', l)][1]   # the body's second step ends here
            waits = [int(k) for l in body[loop_start + 1:loop_end + 1] for k in re.findall(r's_waitcnt.*?vmcnt\((\d+)\)', l)]
            assert waits == [npw + 2 * nl, npw], (hw, mode, waits)
            first_wait = next(i for i, l in enumerate(body) if i > loop_start and re.search(r's_waitcnt.*?vmcnt', l))
            loop_label = max(i for i, l in enumerate(body[:first_wait]) if re.match(r'^\.LBB', l))
            assert not any(re.search(r's_cbranch|s_branch', l) for l in body[loop_label:loop_end]), (hw, mode, 'a branch inside the loop body')
            nmfma = 8 * (hw // 32 if hw % 32 == 4 else (hw + 31) // 32)
            segs, prev = [], loop_start + 1
            for i in range(loop_start + 1, loop_end + 1):
                w = re.search(r's_waitcnt.*?vmcnt\((\d+)\)', body[i])
                if w:
                    segs.append((int(w.group(1)), body[prev:i])); prev = i + 1
            for cnt, seg in segs:
                first = cnt != npw                                                  # the body's first step: it also requests the next body's pieces
                ops = ''.join('S' if 'global_store_dwordx4' in l else 'D' if 'global_load_lds_dword' in l else 'L'
                              for l in seg if re.search(r'global_(store|load)', l))
                assert ops.count('D') == npw and ops.count('L') == (2 * nl if first else 0), (hw, mode, cnt, ops)
                assert ops.count('S') == (2 if mode == 0 else 0) and ops.startswith('SS' if mode == 0 else ''), (hw, mode, cnt, ops)
                if not first: assert ops[-npw:] == 'D' * npw, (hw, mode, cnt, ops)  # vmcnt(NPW) leaves exactly the LDS-DMA requests
                assert sum('v_mfma' in l for l in seg) == nmfma, (hw, mode, cnt)
            # (c): registers in flight are never operands
            flying, in_asm = set(), False

            def regs(text):
                out = set()
                for lo, hi in re.findall(r'\bv\[(\d+):(\d+)\]', text): out.update(range(int(lo), int(hi) + 1))
                out.update(int(v) for v in re.findall(r'\bv(\d+)\b', text))
                return out
            # ... up to the wait for everything behind the loop: nothing may be in flight on ANY way out of it (behind the loop
            # the accumulators are moved into registers the compiler considers free)
            after = next(i for i, l in enumerate(body) if i > loop_end and re.search(r's_waitcnt vmcnt\(0\)', l))
            for l in body[:after + 1]:
                code = l.split(';')[0].strip()
                if '#ASMSTART' in l: in_asm = True; continue
                if '#ASMEND' in l: in_asm = False; continue
                if not code or code.endswith(':'): continue
                w = re.search(r's_waitcnt.*?vmcnt\((\d+)\)', code)
                if w:
                    if int(w.group(1)) <= npw: flying.clear()
                    continue
                if in_asm and code.startswith('global_load_dword'):
                    dst, addr = code.split(',')[0], ','.join(code.split(',')[1:])
                    assert not (regs(addr) & flying), (hw, mode, code)
                    flying |= regs(dst)
                    continue
                assert not (regs(code) & flying), (hw, mode, 'reads or writes a register in flight', code)
            assert not flying


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_gram_forward_leaves_in_16_byte_stores_and_never_spills(tmp_path):
    """bcnn_gram_panel_kernel (bcnn_fast.hip, hk_gram_tile.h): every element of y - direct and mirrored - leaves in a
    16-byte store (the epilogue turns each sub-tile through wave-private LDS for that), nothing goes to scratch, and the
    150 KB of panels + 8 KB of turn-tables fit the 160 KB of a CU."""
    src = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'bcnn_fast.hip')
    out = str(tmp_path / 'bcnn_fast.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + CSRC, '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=900)
    txt = open(out).read()
    assert not _serialised_staging_loops(txt), _serialised_staging_loops(txt)
    seen = 0
    for m in re.finditer(r'^(_ZN2hk22bcnn_gram_panel_kernelILi(\d+)ELi(\d)ELb(\d)\w*):.*?\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        body = m.group(5)
        seen += 1
        assert 'scratch_' not in body, m.group(1)
        assert len(re.findall(r'global_store_dwordx4', body)) >= 16, m.group(1)
        # the only narrow stores: the norm / colsum / partial-norm words of the prologue and tail (not y); the centred
        # (covariance) instance also writes its channel means, four rows per wave and call site of center_panel
        # (BCNN instance: colsum / inv_norm are written on two exclusive paths - partial sums from another launch, or the
        #  column sums formed here)
        assert len(re.findall(r'global_store_dword ', body)) <= (6 if m.group(4) == '0' else 16), m.group(1)
        assert int(re.search(r'group_segment_fixed_size (\d+)', body).group(1)) <= 160 * 1024, m.group(1)
    assert seen == 16                                              # 4 map sizes x (BCNN, signed sqrt, covariance centred / raw)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_classifier_kernels_keep_their_counted_waits(tmp_path):
    """linear_bwd64_kernel / linear_skinny_kernel (hk_linear_bwd.h, hk_linear_fwd.h) end every pipeline unit with
    `s_waitcnt vmcnt(n)` + `s_barrier`, n = the vector-memory operations the wave has issued SINCE the pieces it waits for.
    The count is exact only while every LDS-DMA piece and every buffer store is one instruction that always issues.  The
    CPU emulation executes LDS-DMA synchronously, so a compiler (or an edit) that merged, dropped or predicated one of them
    would pass every emulated test and race on the GPU only.  Pinned here, per instance:
      * the number of global_load_lds / buffer_store instructions in the ISA (each source-level request is one instruction);
      * that the waits the source asks for are there: dy role vmcnt(PW), (PW - 2), (PW + SW), (PW - 2 + SW), dW role 4 / 22,
        with NPW = ceil(NKS / 8), PW = 2 NPW, SW = 3 x (2 dy + 2 tile-12 stores) per unit triple;
      * spill code: at most four dwords of scratch (the both-products instances sit at the 256-register budget) and none of
        it inside the steady-state loops - spill traffic is vector memory too; it can only make a counted wait stricter,
        but it does not belong between two MFMA units."""
    src = os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'linear.hip')
    out = str(tmp_path / 'linear.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + CSRC, '-S', '--cuda-device-only', '-o', out, src],
                   check=True, capture_output=True, timeout=900)
    txt = open(out).read()
    assert not _serialised_staging_loops(txt), _serialised_staging_loops(txt)
    seen = 0
    for m in re.finditer(r'^_ZN2hk19linear_bwd64_kernelILi(\d+)ELi(\d)ELi0E\w*:.*?\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        nks, mode, body = int(m.group(1)), int(m.group(2)), m.group(3)
        seen += 1
        lines = body.split('\n')
        waits = {int(k) for k in re.findall(r's_waitcnt vmcnt\((\d+)\)', body)}
        npw = (nks // 2 + 3) // 4
        pw = 2 * npw if mode != 2 else 0
        sw = 3 * ((2 if mode != 2 else 0) + (2 if mode != 1 else 0))
        if mode != 2:
            assert {pw, pw - 2, pw + sw, pw - 2 + sw} <= waits, (nks, mode, sorted(waits))
        if mode != 1:
            assert {4, 22} <= waits, (nks, mode, sorted(waits))
        assert int(re.search(r'private_segment_fixed_size (\d+)', body).group(1)) <= 16, (nks, mode)
        # every unrolled unit issues its pieces / stores as single instructions: the totals of the instance
        n_glds = len(re.findall(r'global_load_lds_dwordx4', body))
        n_st16 = len(re.findall(r'buffer_store_dwordx4', body))
        n_st4 = len(re.findall(r'buffer_store_dword ', body))
        want = {0: (117, 112, 28), 1: (91, 28, 0), 2: (26, 84, 28)}[mode]
        assert (n_glds, n_st16, n_st4) == want, (nks, mode, n_glds, n_st16, n_st4)
        # no spill instruction inside a STEADY-STATE loop: an innermost loop (no other backward branch inside it) with the
        # MFMAs of at least two pipeline units and LDS-DMA.  (The peeled head / tail units - every one of them ends on
        # vmcnt(0) or runs at most four times - may carry the role's few spill reloads.)
        labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loops.append((labels[mm.group(1)], i))
        steady = 0
        for a, b in loops:
            if any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in loops):
                continue
            seg = lines[a:b]
            if sum('v_mfma' in x for x in seg) >= 200 and any('global_load_lds' in x for x in seg):
                steady += 1
                assert not any(re.match(r'\s+scratch_', x) for x in seg), (nks, mode, a, b)
        assert steady >= 1 or mode != 0, (nks, mode, 'no steady-state loop found')
    assert seen == 6
    for m in re.finditer(r'^_ZN2hk20linear_skinny_kernelILi(\d+)ELi(\d+)ELi0E\w*:.*?\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        body = m.group(3)
        seen += 1
        assert 'scratch_' not in body, m.group(1)
        waits = [int(k) for k in re.findall(r's_waitcnt vmcnt\((\d+)\)', body)]
        assert any(w_ > 0 for w_ in waits), 'the chunk loop waits for all but the newest pieces'
        assert len(re.findall(r'global_load_lds_dwordx4', body)) >= 10
    assert seen >= 8
