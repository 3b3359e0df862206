"""bench.py's final stdout line: the driver parses exactly that line, and round 3's (24.5 KB: 84 A/B rows on it) was past
its capture window (BENCH_r03.json `parsed: null`).  These tests hold the line builder to < 4 KB, to the contract's
keys, and to honest roofline fractions (never > 1; priced on executed work where a kernel skips part of the
algorithmic FLOPs).  No GPU, no kernels: canned numbers."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned():
    ks = [
        bench.roof_row('bcnn_gram_panel_kernel<196>', 44.65, 6.576e9, 92.8e6, 6.576e9 * 36 / 64, 49.0),
        bench.roof_row('gram_bwd3_kernel<196,0,2>', 62.47, 6.576e9, 185.6e6, None, 70.0),
        bench.roof_row('hk_linear_fwd: linear_skinny_kernel + linear_reduce_kernel (classifier 262144->200)', 72.4, 6.711e9,
                       276.9e6, 6.711e9 * 208 / 200, 80.0),
        bench.roof_row('bcnn_rank1_fix_kernel', 10.7, 0.0, 51.4e6),
    ]
    dom = ks[2]
    res = {
        'metric': 'images/sec (train fwd+bwd) BCNN VGG-16 448^2 bs64', 'value': 301.12, 'unit': 'images/sec', 'n_gpus': 1,
        'steps': 20, 'warmup': 5, 'ms_per_step': 212.541, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BCNN_S2: VGG-16 448x448, per-GPU batch 64, 200 classes, SGD momentum 0.9, CE label_smoothing '
                               '0.1, full train step', 'global_batch': 64, 'parallelism': 'dp1', 'memory_format': 'channels_last'},
        'roofline': {'bound': dom['bound'], 'achieved': dom['achieved'], 'peak': dom['peak'], 'unit': dom['unit'],
                     'frac': dom['frac'], 'traffic': 303269365, 'kernel': dom['kernel'][:96], 'us': dom['us'],
                     'flops_algorithmic': dom['flops_algorithmic'], 'flops_executed': dom['flops_executed'],
                     'bytes_algorithmic': dom['bytes_algorithmic'], 'traffic_source': 'profiles/r5_pool_kernels_pmc.csv',
                     # the SURVEY 8(a) entry points beside the dominant kernel: kernel, us, frac (round 6)
                     'also': [{'kernel': k['kernel'][:48], 'us': k['us'], 'frac': k['frac'], 'bound': k['bound']} for k in ks[:3]]},
        'ddp': {'buckets_mib': [200.0] + [16.0] * 5, 'issued_ms': [0.3, 20.0, 40.0, 60.0, 80.0, 139.0], 'backward_end_ms': 140.0,
                'joined_ms': 141.0, 'ranks_seen': list(range(8)), 'distinct_devices': 8, 'devices': ['0:5:0'] * 8, 'backend': 'nccl',
                'comm_size': 8, 'host_threads_per_rank': 32,
                'rccl': {'version': 'RCCL 2.22.3', 'channels': 32, 'rings': 32, 'trees': 32, 'transports': {'P2P/IPC': 448},
                         'tuning': ['AllReduce: 209715200 Bytes -> Algo 1 proto 2 time 1234.5'[:80]] * 2}},
        'cpu_baseline': {'value': 2.35, 'unit': 'images/sec', 'cores': 32, 'kind': 'port',
                         'sample': 'BCNN stage-2 train step, batch 4, 448x448, best of 3 after 1 warm-up, torch CPU fp32, 32 '
                                   'threads of 256 host cores'},
        'kernels': ks, 'candidates': [{'x': 'y' * 100}] * 300,          # must never reach the line
    }
    detail = {'kernels': ks, 'other_models': [dict(k, model='MPN') for k in ks] * 20}
    return res, detail


def test_final_line_is_small_and_has_the_contract_keys():
    res, _ = canned()
    line = bench.final_line(res)
    assert len(line) < 4096 and '\n' not in line
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert set(d) <= set(bench.LINE_KEYS)                      # nothing else rides on it
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in d['roofline'], k
    assert all(set(r) == {'kernel', 'us', 'frac', 'bound'} and r['frac'] <= 1.0 for r in d['roofline']['also'])
    assert d['ddp']['rccl']['channels'] == 32
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in d['cpu_baseline'], k
    assert 'workload' in d['config'] and 'model' not in d['config']


def test_final_line_survives_oversized_strings():
    res, _ = canned()
    res['cpu_baseline']['sample'] = 'x' * 6000
    line = bench.final_line(res)
    assert len(line) < 4096
    assert json.loads(line)['value'] == res['value']


def test_emit_prints_the_line_last_on_stdout_and_detail_elsewhere(tmp_path, monkeypatch):
    res, detail = canned()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.delenv('HAWKEYE_BENCH_DETAIL', raising=False)
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(res, detail)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096             # the ONLY stdout line
    assert json.loads(lines[-1])['roofline']['frac'] <= 1.0
    side = json.load(open(tmp_path / 'gpurun_out' / 'bench_detail.json'))
    assert len(side['kernels']) == 4 and len(side['other_models']) == 80
    assert 'linear_skinny_kernel' in err.getvalue()


def test_fractions_are_priced_on_executed_work_and_never_exceed_one():
    g = bench.roof_row('bcnn_gram_panel_kernel<196>', 44.65, 6.576e9, 92.8e6, 6.576e9 * 36 / 64)
    assert g['bound'] == 'mfma'
    assert g['frac'] == pytest.approx(3.699e9 / 44.65e-6 / 157.3e12, rel=1e-3)         # 0.53, not the 1.01 of round 3
    fast = bench.roof_row('x', 20.0, 6.576e9, 92.8e6)                                   # faster than the algorithmic bound
    assert fast['frac'] <= 1.0
    pad = bench.roof_row('linear', 72.4, 6.711e9, 276.9e6, 6.711e9 * 1.04)             # padding is not credited
    assert pad['frac'] == pytest.approx(6.711e9 / 72.4e-6 / 157.3e12, rel=1e-3)
    hbm = bench.roof_row('rank1', 10.7, 0.0, 51.4e6)
    assert hbm['bound'] == 'hbm' and hbm['unit'] == 'GB/s' and 0 < hbm['frac'] < 1
