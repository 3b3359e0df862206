"""GPU parity of the opt-in variants that were written after round 1's GPU budget was spent: validated against the
oracle on the CPU emulation tier (tests/test_emu_parity.py collects these cases too), first run on a real MI355X
by the round-end `-m gpu` pass.  The file sorts after the validated suites on purpose.  Each variant is OFF by default
in the library (environment switch named in the test) until it has a measured number.
"""
import numpy as np
import pytest
import torch

import hawkeye_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = (b.detach() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def F():
    import hawkeye_amd.functional as F_
    from hawkeye_amd import _lib
    lib = _lib.load()
    assert b'gfx950' in lib.hk_version()
    return F_


@pytest.mark.parametrize('b,d,itn', [(3, 70, 5), (2, 128, 5), (9, 64, 3), (2, 33, 2)])
def test_ns_symmetric_tile_mode(F, b, d, itn, monkeypatch):
    """HK_NS_SYM=1: upper-triangle tiles only + mirrored stores for the symmetric products of the Newton-Schulz chain,
    and Z Y taken as the transpose of Y Z in the backward (38 -> 34 launches).  Same tolerances as the full products."""
    x = torch.relu(torch.randn(b, d, 6, 7, generator=torch.Generator().manual_seed(d))) + 0.01
    xo = x.clone().requires_grad_(True)
    yo = O.triuvec(O.sqrtm(O.covpool(xo), itn))
    wt = torch.randn(yo.shape, generator=torch.Generator().manual_seed(1))
    (yo * wt).sum().backward()
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('HK_NS_SYM', flag)
        xg = x.clone().to(DEV).requires_grad_(True)
        cov = F.covpool(xg)
        s = F.sqrtm(cov, itn)
        yg = F.triuvec(s)
        (yg * wt.to(DEV)).sum().backward()
        assert rel(yg, yo) < 1e-5 and rel(xg.grad, xo.grad) < 1e-4
        if flag == '1':
            assert rel(s, s.transpose(1, 2)) < 1e-6            # off-diagonal tiles mirrored, diagonal tiles computed
        res.append((yg.detach(), xg.grad))
    assert rel(res[1][0], res[0][0]) < 5e-6 and rel(res[1][1], res[0][1]) < 5e-5


@pytest.mark.parametrize('b,j,k,bias', [(3, 1000, 7, True), (64, 4096, 200, True), (5, 333, 130, False), (10, 6272, 96, True),
                                        (1, 40, 1, True)])
def test_linear_split_k(F, b, j, k, bias, monkeypatch):
    """hk_linear_fwd/bwd (classifier on the pooled vector) vs torch's Linear in fp64; slab counts forced
    through HK_LINEAR_SLABS cover one slab, ragged last slabs and the automatic choice."""
    gen = torch.Generator().manual_seed(b * 1000 + j)
    y = torch.randn(b, j, generator=gen)
    w = torch.randn(k, j, generator=gen) / j ** 0.5
    bv = torch.randn(k, generator=gen) if bias else None
    g = torch.randn(b, k, generator=gen)
    y64, w64 = y.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = bv.double().requires_grad_(True) if bias else None
    o64 = torch.nn.functional.linear(y64, w64, b64)
    (o64 * g.double()).sum().backward()
    for slabs in (None, '1', '3', '7'):
        if slabs is None:
            monkeypatch.delenv('HK_LINEAR_SLABS', raising=False)
        else:
            monkeypatch.setenv('HK_LINEAR_SLABS', slabs)
        yg, wg = y.clone().to(DEV).requires_grad_(True), w.clone().to(DEV).requires_grad_(True)
        bg = bv.clone().to(DEV).requires_grad_(True) if bias else None
        og = F.linear(yg, wg, bg)
        (og * g.to(DEV)).sum().backward()
        assert rel(og, o64) < 2e-6, slabs
        assert rel(yg.grad, y64.grad) < 2e-6 and rel(wg.grad, w64.grad) < 2e-6
        if bias:
            assert rel(bg.grad, b64.grad) < 2e-6


_MODEL_CFG = {
    'BCNN': dict(stage=2, num_classes=200),
    'CBCNN': dict(stage=2, num_classes=200, input_channel=512, output_channel=6000),
    'MPN': dict(iter_num=5, is_sqrt=True, is_vec=True, input_dim=2048, dimension_reduction=256, num_classes=200),
}


@pytest.mark.parametrize('name', ['BCNN', 'CBCNN', 'MPN'])
def test_models_with_hip_classifier(F, name, monkeypatch):
    """HAWKEYE_HIP_LINEAR=1: the classifier on the pooled vector runs on hk_linear_* - logits still match the REFERENCE
    model's (tests/golden/model_logits.npz) and a train step gives the same classifier gradient as torch's Linear."""
    import os

    import hawkeye_amd.model  # noqa: F401
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.model.registry import MODEL
    from inputs import rs_randn, seeded_init
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'model_logits.npz'))
    m = MODEL.get(name)(CfgNode(dict(name=name, **_MODEL_CFG[name])))
    seeded_init(m, 900)
    m = m.to(DEV).eval()
    x = torch.from_numpy(np.ascontiguousarray(rs_randn(901, (2, 3, 64, 64)))).to(DEV)
    grads = []
    for flag in ('0', '1'):
        monkeypatch.setenv('HAWKEYE_HIP_LINEAR', flag)
        m.zero_grad()
        y = m(x)
        assert rel(y, g[name]) < 1e-4 and y.argmax(1).cpu().tolist() == g[name].argmax(1).tolist()
        torch.nn.functional.cross_entropy(y, torch.tensor([3, 77], device=y.device)).backward()
        grads.append((m.classifier.weight.grad.clone(), m.classifier.bias.grad.clone(),
                      next(m.backbone.parameters()).grad.clone()))
    for p, q in zip(grads[0], grads[1]):
        assert rel(q, p) < 1e-5


def test_mamc_npairs_loss_vs_reference_goldens(F):
    """hk_npairs_loss vs the REFERENCE's NPairsLoss / MAMCLoss (tests/golden/mamc_loss.npz) and the
    oracle, including empty positive / negative sets."""
    import os

    from inputs import MAMC_CASES, rs_randn
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.model.loss import MAMCLoss
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mamc_loss.npz'))
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for i, (name, (b, p, d, labels)) in enumerate(MAMC_CASES.items()):
        x = tt(rs_randn(300 + i, (b, p, d))).to(DEV).requires_grad_(True)
        loss = F.npairs_loss(x, torch.tensor(labels).to(DEV))
        (3.0 * loss).backward()
        xo = tt(rs_randn(300 + i, (b, p, d))).requires_grad_(True)
        lo = O.npairs_loss(xo, torch.tensor(labels))
        lo.backward()
        assert abs(float(loss) - float(g[name + '_loss'])) <= 2e-6 * max(1.0, abs(float(g[name + '_loss']))), name
        assert abs(float(loss) - float(lo)) <= 2e-6 * max(1.0, abs(float(lo)))
        if float(np.abs(g[name + '_dx']).max()) > 0:
            assert rel(x.grad / 3.0, g[name + '_dx']) < 2e-5 and rel(x.grad / 3.0, xo.grad) < 2e-5, name
        else:
            assert float(x.grad.abs().max()) < 1e-7
    b, p, d, labels = MAMC_CASES['balanced']
    x = tt(rs_randn(300, (b, p, d))).to(DEV).requires_grad_(True)
    pred = tt(rs_randn(310, (b, 200))).to(DEV).requires_grad_(True)
    total = MAMCLoss(CfgNode(dict(lambda_a=0.5, use_mamc=True)))((pred, x), torch.tensor(labels).to(DEV))
    total.backward()
    assert abs(float(total) - float(g['mamc_total'])) <= 2e-6 * abs(float(g['mamc_total']))
    assert rel(pred.grad, g['mamc_dpred']) < 1e-5 and rel(x.grad, g['mamc_dx']) < 2e-5


def test_mamc_npairs_loss_larger_batch(F):
    """n = 96 anchors (32 samples x 3 attentions, 7 classes): more than one 64-row tile in both GEMMs."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(32, 3, 200, generator=gen)
    y = torch.randint(0, 7, (32,), generator=gen)
    xo = x.clone().requires_grad_(True)
    lo = O.npairs_loss(xo, y)
    lo.backward()
    xg = x.clone().to(DEV).requires_grad_(True)
    lg = F.npairs_loss(xg, y.to(DEV))
    lg.backward()
    assert abs(float(lg) - float(lo)) <= 5e-6 * abs(float(lo)) and rel(xg.grad, xo.grad) < 2e-5


@pytest.mark.parametrize('b,d,itn', [(2, 128, 5), (3, 200, 3), (9, 256, 2)])
def test_ns_128_tile_gemm_variant(F, b, d, itn, monkeypatch):
    """HK_NS_GEMM=4: the Newton-Schulz products on bgemm128_kernel (128x128 tile, 8 waves, two-chunk prefetch through
    two register sets).  Same k order as the 64x64 kernel, so results agree to rounding of the tile boundaries only."""
    x = torch.relu(torch.randn(b, d, 5, 6, generator=torch.Generator().manual_seed(d + 1))) + 0.01
    xo = x.clone().requires_grad_(True)
    yo = O.sqrtm(O.covpool(xo), itn)
    wt = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2))
    (yo * wt).sum().backward()
    res = []
    for flag in ('0', '4'):
        monkeypatch.setenv('HK_NS_GEMM', flag)
        xg = x.clone().to(DEV).requires_grad_(True)
        yg = F.sqrtm(F.covpool(xg), itn)
        (yg * wt.to(DEV)).sum().backward()
        assert rel(yg, yo) < 1e-5 and rel(xg.grad, xo.grad) < 1e-4
        res.append((yg.detach(), xg.grad))
    assert rel(res[1][0], res[0][0]) < 2e-6 and rel(res[1][1], res[0][1]) < 2e-5
