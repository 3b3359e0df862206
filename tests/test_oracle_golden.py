"""Pins the CPU oracle (oracle/hawkeye_oracle.py) to outputs of the REFERENCE
ITSELF (tests/golden/*.npz, written by oracle/gen_golden.py from /root/reference).
CPU only.  Tolerances: the oracle re-runs the same torch-CPU ops, so results are
expected bit-identical or within a few ulp (different bmm blocking is allowed)."""
import os
import random

import numpy as np
import pytest
import torch

import hawkeye_oracle as O
from inputs import rs_randn, rs_relu_randn, sub

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name + '.npz'))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=2e-6, atol=1e-7):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_bcnn_small():
    g = load('bcnn_small')
    x = t(rs_relu_randn(11, (3, 32, 5, 7))).requires_grad_(True)
    y = O.bilinear_pool(x)
    (y * t(rs_randn(12, tuple(y.shape)))).sum().backward()
    close(y, g['y'])
    close(x.grad, g['dx'], rtol=1e-5, atol=1e-7)


def test_bcnn_512_known_answers():
    g = load('bcnn_512')
    x = t(rs_relu_randn(1234, (2, 512, 14, 14))).requires_grad_(True)
    y = O.bilinear_pool(x)
    (y * t(rs_randn(1235, tuple(y.shape)))).sum().backward()
    close(sub(y), g['y_sub'])
    assert y.argmax(dim=1).tolist() == g['y_argmax'].tolist()
    close(y.norm(dim=1), g['y_rownorm'])
    close(sub(x.grad), g['dx_sub'], rtol=1e-5, atol=1e-8)
    assert abs(float(x.grad.double().abs().sum()) / float(g['dx_abs']) - 1) < 1e-6


def test_sketch_hashes_match_reference_matrices():
    g = load('cbp_512')
    h1, s1, h2, s2 = O.sketch_hashes(512, 512, 6000)
    assert (h1 == g['h1']).all() and (h2 == g['h2']).all()
    assert (s1 == g['sgn1']).all() and (s2 == g['sgn2']).all()
    # SURVEY.md section 8 A2' golden prefix
    assert h1[:8].tolist() == [5157, 235, 3980, 5192, 905, 2763, 2895, 5056]
    assert s2[:8].tolist() == [1, -1, 1, -1, 1, 1, 1, 1]


def test_cbp():
    g = load('cbp_small')
    x = t(rs_relu_randn(21, (2, 16, 3, 5))).requires_grad_(True)
    y = O.compact_bilinear_pool(x, 64)
    (y * t(rs_randn(22, tuple(y.shape)))).sum().backward()
    close(y, g['y'], rtol=1e-5, atol=1e-7)
    close(x.grad, g['dx'], rtol=1e-4, atol=1e-6)
    g = load('cbp_512')
    x = t(rs_relu_randn(1234, (2, 512, 14, 14))).requires_grad_(True)
    y = O.compact_bilinear_pool(x, 6000)
    (y * t(rs_randn(1236, tuple(y.shape)))).sum().backward()
    close(y, g['y'], rtol=1e-5, atol=1e-7)
    assert abs(float(x.grad.double().abs().sum()) / float(g['dx_abs']) - 1) < 1e-5


def test_cbp_gram_identity_equals_fft_route():
    """The count-sketch identity the HIP path relies on, checked on the CPU against the reference's FFT route
    (goldens from /root/reference) for inputs without exactly-zero bins: forward and input gradient."""
    g = load('cbp_small_dense')
    xn = (np.abs(rs_randn(23, (2, 16, 3, 5))) + 0.1).astype(np.float32)
    x = t(xn).requires_grad_(True)
    y = O.compact_bilinear_pool_gram(x, 64)
    (y * t(rs_randn(24, (2, 64)))).sum().backward()
    close(y, g['y'], rtol=1e-4, atol=1e-6)
    close(x.grad, g['dx'], rtol=1e-3, atol=1e-5)
    x2 = t(xn).requires_grad_(True)
    y2 = O.compact_bilinear_pool(x2, 64)
    (y2 * t(rs_randn(24, (2, 64)))).sum().backward()
    close(y2, g['y'], rtol=1e-5, atol=1e-7)
    g = load('cbp_512')
    x = t(rs_relu_randn(1234, (2, 512, 14, 14))).requires_grad_(True)
    y = O.compact_bilinear_pool_gram(x, 6000)
    (y * t(rs_randn(1236, (2, 6000)))).sum().backward()
    assert float((y.detach() - t(g['y'])).norm() / t(g['y']).norm()) < 2e-6
    assert abs(float(x.grad.double().abs().sum()) / float(g['dx_abs']) - 1) < 1e-4


@pytest.mark.parametrize('it', [5, 3, 2, 1])
def test_mpn_small(it):
    g = load('mpn_small')
    x = t(rs_relu_randn(31, (2, 16, 4, 5))).requires_grad_(True)
    cov = O.covpool(x)
    cov.retain_grad()
    sq = O.sqrtm(cov, it)
    sq.retain_grad()
    tv = O.triuvec(sq)
    assert list(tv.shape) == g['triu_shape'].tolist()
    (tv * t(rs_randn(32, tuple(tv.shape)))).sum().backward()
    close(cov, g[f'cov_it{it}'], atol=1e-7)
    close(sq, g[f'sqrtm_it{it}'], rtol=1e-5, atol=1e-7)
    close(tv, g[f'triu_it{it}'], rtol=1e-5, atol=1e-7)
    close(sq.grad, g[f'dsq_it{it}'])
    close(cov.grad, g[f'dcov_it{it}'], rtol=1e-4, atol=1e-5)
    close(x.grad, g[f'dx_it{it}'], rtol=1e-4, atol=1e-5)


def test_mpn_256():
    g = load('mpn_256')
    x = t(rs_relu_randn(31, (2, 256, 14, 14))).requires_grad_(True)
    cov = O.covpool(x)
    sq = O.sqrtm(cov, 5)
    tv = O.triuvec(sq)
    (tv * t(rs_randn(32, tuple(tv.shape)))).sum().backward()
    close(sub(cov), g['cov_it5'], rtol=1e-5, atol=1e-7)
    close(sub(sq), g['sqrtm_it5'], rtol=1e-4, atol=1e-6)
    close(sub(x.grad), g['dx_it5'], rtol=1e-3, atol=1e-6)
    assert abs(float(tv.double().sum()) / float(g['triu_sum_it5']) - 1) < 1e-5


def _apn_weights(g):
    return {k[2:].replace('__', '.'): t(g[k]) for k in g.files if k.startswith('w_')}


def test_apcnn_attention_pool():
    g = load('apcnn_apn')
    w = _apn_weights(g)
    feats = [t(rs_randn(41 + i, (2, 32, s, s))).requires_grad_(True) for i, s in enumerate((28, 14, 7))]
    spatial, channel = [], []
    for lvl, f in zip((3, 4, 5), feats):
        spatial.append(torch.sigmoid(torch.nn.functional.conv_transpose2d(
            f, w[f'A{lvl}_1.conv.weight'], w[f'A{lvl}_1.conv.bias'], stride=1, padding=1)))
        z = torch.nn.functional.adaptive_avg_pool2d(f, 1)
        z = torch.relu(torch.nn.functional.conv2d(z, w[f'A{lvl}_2.conv1.weight'], w[f'A{lvl}_2.conv1.bias']))
        channel.append(torch.sigmoid(torch.nn.functional.conv2d(z, w[f'A{lvl}_2.conv2.weight'], w[f'A{lvl}_2.conv2.bias'])))
    pooled = O.attention_pooled(feats, spatial, channel)
    ws = [t(rs_randn(44 + i, tuple(p.shape))) for i, p in enumerate(pooled)]
    sum((p * wi).sum() for p, wi in zip(pooled, ws)).backward()
    for p, k in zip(pooled, ('pooled3', 'pooled4', 'pooled5')):
        close(p, g[k], rtol=1e-5, atol=1e-7)
    close(spatial[0], g['s3'])
    close(feats[2].grad, g['df5'], rtol=1e-4, atol=1e-8)
    close(sub(feats[0].grad), g['df3'], rtol=1e-4, atol=1e-8)


def _masks():
    return [t(1.0 / (1.0 + np.exp(-2.0 * rs_randn(50 + l, (3, 1, hw, hw))))).float()
            for l, hw in enumerate((56, 28, 14))]


LEVELS = ((8, 64, 5), (16, 128, 3), (32, 256, 1))


@pytest.mark.parametrize('ncls', [200, 8142])
def test_apcnn_roi_select(ncls):
    g = load('apcnn_roi')
    for lvl, (m, (s, a, k)) in enumerate(zip(_masks(), LEVELS)):
        roi = O.att_roi(m, s, a, 448, 448, ncls, 0.05, k)
        np.testing.assert_array_equal(roi.numpy(), g[f'roi_c{ncls}_l{lvl + 3}'])


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_apcnn_roi_crop(mode):
    g = load('apcnn_crop')
    rois = [t(g['roi3']), t(g['roi4']), t(g['roi5'])]
    x2 = t(rs_randn(60, (3, 8, 56, 56))).requires_grad_(True)
    random.seed(7)           # same python `random` stream as the reference run
    y = O.roi_crop_feat(x2, rois, 8, training=(mode == 'train'))
    (y * t(rs_randn(61, tuple(y.shape)))).sum().backward()
    close(sub(y, 61), g[f'y_{mode}'], rtol=1e-5, atol=1e-7)
    close(sub(x2.grad, 61), g[f'dx_{mode}'], rtol=1e-5, atol=1e-7)
    if mode == 'train':      # injected decisions reproduce the same result
        drops = [None if l == 0 else (int(l), int(i)) for l, i in g['drops']]
        y2 = O.roi_crop_feat(x2.detach(), rois, 8, training=True, drops=drops)
        close(y2, y.detach(), rtol=0, atol=0)


def test_osme():
    g = load('osme_small')
    w = {k[2:].replace('__', '.'): t(g[k]) for k in g.files if k.startswith('w_')}
    x = t(rs_relu_randn(71, (3, 32, 7, 7))).requires_grad_(True)
    gates = [(w[f'blocks.{p}.block.0.weight'], w[f'blocks.{p}.block.0.bias'],
              w[f'blocks.{p}.block.2.weight'], w[f'blocks.{p}.block.2.bias']) for p in range(2)]
    fcs = [(w[f'fcs.{p}.weight'], w[f'fcs.{p}.bias']) for p in range(2)]
    f, parts = O.osme_forward(x, gates, fcs)
    ((parts * t(rs_randn(72, tuple(parts.shape)))).sum() + f.sum()).backward()
    close(f, g['f'], rtol=1e-5, atol=1e-6)
    close(parts, g['parts'], rtol=1e-5, atol=1e-6)
    close(x.grad, g['dx'], rtol=1e-4, atol=1e-6)


def test_mamc_npairs_loss():
    """oracle npairs_loss / mamc_loss (vectorised) vs the reference's per-anchor python loop (MAMC_loss.py:57-90),
    including the cases where positive or negative sets are empty."""
    from inputs import MAMC_CASES
    g = load('mamc_loss')
    for i, (name, (b, p, d, labels)) in enumerate(MAMC_CASES.items()):
        x = t(rs_randn(300 + i, (b, p, d))).requires_grad_(True)
        loss = O.npairs_loss(x, torch.tensor(labels))
        loss.backward()
        close(loss, g[name + '_loss'], rtol=2e-6, atol=1e-7)
        close(x.grad, g[name + '_dx'], rtol=1e-4, atol=2e-7)
    b, p, d, labels = MAMC_CASES['balanced']
    x = t(rs_randn(300, (b, p, d))).requires_grad_(True)
    pred = t(rs_randn(310, (b, 200))).requires_grad_(True)
    total = O.mamc_loss(pred, x, torch.tensor(labels), 0.5, True)
    total.backward()
    close(total, g['mamc_total'], rtol=2e-6, atol=1e-7)
    close(pred.grad, g['mamc_dpred'], rtol=1e-5, atol=1e-8)
    close(x.grad, g['mamc_dx'], rtol=1e-4, atol=2e-7)


def test_cin_channel_interaction_and_loss():
    """oracle channel_interaction / cin_loss vs the reference's ChannelInteractionModule (train + eval) and CINLoss."""
    g = load('cin_small')
    w = {k[2:].replace('__', '.'): t(g[k]).requires_grad_(True) for k in g.files if k.startswith('w_')}
    x = t(rs_relu_randn(410, (4, 24, 3, 4))).requires_grad_(True)
    z, zc = O.channel_interaction(x, w['conv.weight'], w['conv.bias'], w['fc.weight'], w['fc.bias'], True)
    ((z * t(rs_randn(411, tuple(z.shape)))).sum() + (zc * t(rs_randn(412, tuple(zc.shape)))).sum()).backward()
    close(z, g['z'], rtol=1e-5, atol=1e-6)
    close(zc, g['z_cci'], rtol=1e-5, atol=1e-6)
    close(x.grad, g['dx'], rtol=1e-4, atol=1e-6)
    for k, v in w.items():
        close(v.grad, g['g_' + k.replace('.', '__')], rtol=1e-4, atol=1e-6)
    ze = O.channel_interaction(x.detach(), w['conv.weight'], w['conv.bias'], w['fc.weight'], w['fc.bias'], False)
    close(ze, g['z_eval'], rtol=1e-5, atol=1e-6)
    for name, labels in (('pairs', [1, 3, 1, 1]), ('nopairs', [1, 3, 0, 2])):
        logits = t(rs_randn(422, (4, 5))).requires_grad_(True)
        zc2 = t(rs_randn(423, (4, 24, 12))).requires_grad_(True)
        loss = O.cin_loss((logits, zc2), torch.tensor(labels), t(g['h_w']), t(g['h_b']), 2.0, 0.5)
        loss.backward()
        close(loss, g[f'loss_{name}'], rtol=1e-5, atol=1e-6)
        close(logits.grad, g[f'loss_{name}_dlogits'], rtol=1e-5, atol=1e-7)
        close(zc2.grad if zc2.grad is not None else torch.zeros_like(zc2), g[f'loss_{name}_dz'], rtol=1e-4, atol=1e-6)
