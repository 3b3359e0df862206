"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported from
/root/reference, CPU, fp32) on seeded inputs.  Runs only in the build
container; the fixtures it writes are committed and travel to the GPU box.

    python oracle/gen_golden.py            # rewrites tests/golden/

Inputs are drawn from numpy's legacy ``RandomState`` (MT19937: stable across
numpy versions) and are NOT stored - tests regenerate them with
``tests/golden/inputs.py``; outputs are stored in full when small and as a
strided subsample + sums when large.
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('HAWKEYE_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, '_stubs'))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, HERE)

from inputs import rs_randn, rs_relu_randn, rs_signed_channels, sub  # noqa: E402  (tests/golden/inputs.py)

torch.manual_seed(0)
torch.set_num_threads(8)

import model  # noqa: E402  (reference package; registers all methods)
from model.registry import MODEL  # noqa: E402

M_BCNN = sys.modules['model.methods.BCNN']
M_CBCNN = sys.modules['model.methods.CBCNN']
M_MPN = sys.modules['model.methods.MPNCOV']
M_AP = sys.modules['model.methods.APCNN']
M_OSME = sys.modules['model.methods.OSME']

OUT = os.path.join(ROOT, 'tests', 'golden')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + '.npz'),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                           for k, v in arrs.items()})
    print('wrote', name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape)
                          for k, v in arrs.items()})


# ---------------------------------------------------------------- BCNN pool
def gen_bcnn():
    pool = M_BCNN.BilinearPooling()
    # small ragged case, stored in full
    x = t(rs_relu_randn(11, (3, 32, 5, 7))).requires_grad_(True)
    y = pool(x)
    w = t(rs_randn(12, tuple(y.shape)))
    (y * w).sum().backward()
    save('bcnn_small', y=y, dx=x.grad)
    # headline shape C=512, HW=196, subsampled
    x = t(rs_relu_randn(1234, (2, 512, 14, 14))).requires_grad_(True)
    y = pool(x)
    w = t(rs_randn(1235, tuple(y.shape)))
    (y * w).sum().backward()
    save('bcnn_512', y_sub=sub(y), y_sum=y.double().sum(), y_argmax=y.argmax(dim=1),
         y_rownorm=y.norm(dim=1), dx_sub=sub(x.grad), dx_sum=x.grad.double().sum(),
         dx_abs=x.grad.double().abs().sum())


def gen_bcnn_ssqrt():
    """The reference's OTHER normalisation: BCNN.py:23-24 are commented out next to the `sqrt(x + 1e-5)` of :21.  The
    class is rebuilt from the reference's own source text with that pair of lines swapped in, then run."""
    import inspect
    src = inspect.getsource(M_BCNN.BilinearPooling)
    live, dead = '        x = torch.sqrt(x + 1e-5)\n', '        # x = torch.sign(x) * torch.sqrt(torch.abs(x) + 1e-10)\n'
    assert live in src and dead in src, 'reference BCNN.py changed'
    src = src.replace(live, '').replace(dead, dead.replace('# ', '', 1))
    ns = {'torch': torch}
    exec(src, ns)
    pool = ns['BilinearPooling']()
    x = t(rs_signed_channels(15, (3, 32, 5, 7))).requires_grad_(True)  # signed channels: negative Gram entries, none near 0
    y = pool(x)
    w = t(rs_randn(16, tuple(y.shape)))
    (y * w).sum().backward()
    save('bcnn_ssqrt_small', y=y, dx=x.grad)
    x = t(rs_signed_channels(1240, (2, 512, 14, 14))).requires_grad_(True)
    y = pool(x)
    w = t(rs_randn(1241, tuple(y.shape)))
    (y * w).sum().backward()
    save('bcnn_ssqrt_512', y_sub=sub(y), y_sum=y.double().sum(), y_abs=y.double().abs().sum(), y_rownorm=y.norm(dim=1),
         dx_sub=sub(x.grad), dx_sum=x.grad.double().sum(), dx_abs=x.grad.double().abs().sum())


# ---------------------------------------------------------------- CBP
def gen_cbp():
    cbp = M_CBCNN.CompactBilinearPooling(16, 16, 64)
    x = t(rs_relu_randn(21, (2, 16, 3, 5))).requires_grad_(True)
    y = cbp(x)
    w = t(rs_randn(22, tuple(y.shape)))
    (y * w).sum().backward()
    save('cbp_small', y=y, dx=x.grad)
    # dense strictly positive input: no bin is exactly 0, so the backward is well defined
    cbp = M_CBCNN.CompactBilinearPooling(16, 16, 64)
    x = t(np.abs(rs_randn(23, (2, 16, 3, 5))) + 0.1).requires_grad_(True)
    y = cbp(x)
    w = t(rs_randn(24, tuple(y.shape)))
    (y * w).sum().backward()
    save('cbp_small_dense', y=y, dx=x.grad)
    # the forms Hawkeye's own CBCNN never calls: two DIFFERENT inputs (CBCNN.py:96-102) and sum_pool = False (:127-130,
    # [B,H,W,D] with F.normalize running along H), one and two inputs
    # (a location's 256 signed products fall into 64 bins: a few bins nearly cancel, and the reference's own float32 FFT
    #  gradient is ~1e-3 away from its float64 one there - so the module is run in both precisions and the float32 run's
    #  distance from the float64 one is stored as the yardstick, like gen_full's cbp_e32_*)
    res = {}
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for tag, two, sp in (('two', True, True), ('loc', False, False), ('loc_two', True, False)):
        runs = {}
        for dt in (torch.float32, torch.float64):
            cbp = M_CBCNN.CompactBilinearPooling(16, 16, 64, sum_pool=sp)
            cbp.sparse_sketch_matrix1 = cbp.sparse_sketch_matrix1.to(dt)
            cbp.sparse_sketch_matrix2 = cbp.sparse_sketch_matrix2.to(dt)
            x1 = t(np.abs(rs_randn(25, (2, 16, 3, 5))) + 0.1).to(dt).requires_grad_(True)
            x2 = t(np.abs(rs_randn(26, (2, 16, 3, 5))) + 0.1).to(dt).requires_grad_(True)
            y = cbp(x1, x2) if two else cbp(x1)
            (y * t(rs_randn(27, tuple(y.shape))).to(dt)).sum().backward()
            runs[dt] = (y.detach(), x1.grad.clone(), x2.grad.clone() if two else None)
        (y, d1, d2), (y64, d164, d264) = runs[torch.float32], runs[torch.float64]
        res[f'y_{tag}'], res[f'dx1_{tag}'], res[f'dx1_64_{tag}'] = y, d1, d164.float()
        res[f'e32_dx1_{tag}'] = np.array([rel(d1, d164)])
        if two:
            res[f'dx2_{tag}'], res[f'dx2_64_{tag}'], res[f'e32_dx2_{tag}'] = d2, d264.float(), np.array([rel(d2, d264)])
        print('cbp_forms', tag, 'reference fp32 vs fp64: y', rel(y, y64), 'dx1', rel(d1, d164), ('dx2 %g' % rel(d2, d264)) if two else '')
    save('cbp_forms', **res)
    cbp = M_CBCNN.CompactBilinearPooling(512, 512, 6000)
    x = t(rs_relu_randn(1234, (2, 512, 14, 14))).requires_grad_(True)
    y = cbp(x)
    w = t(rs_randn(1236, tuple(y.shape)))
    (y * w).sum().backward()
    s1 = cbp.sparse_sketch_matrix1
    s2 = cbp.sparse_sketch_matrix2
    h1 = s1.abs().argmax(dim=1)
    h2 = s2.abs().argmax(dim=1)
    save('cbp_512', y=y, dx_sub=sub(x.grad), dx_sum=x.grad.double().sum(),
         dx_abs=x.grad.double().abs().sum(),
         h1=h1, sgn1=s1[torch.arange(512), h1], h2=h2, sgn2=s2[torch.arange(512), h2])


def gen_cbp_rect():
    """CompactBilinearPooling with input_dim1 != input_dim2 (CBCNN.py:68-94: one sketch matrix per input width; :104-105 wants
    bottom1 / bottom2 of those widths): (24, 16, 64) in full and (384, 512, 4096) subsampled, sum_pool True and False, outputs
    and both input gradients - the reference in float32 and float64, the float32 run's own distance stored as the yardstick
    (as gen_cbp's cbp_forms) -> cbp_rect.npz."""
    res = {}
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for tag, (c1, c2, d), (b, h, w), stride in (('s', (24, 16, 64), (2, 3, 5), 1), ('L', (384, 512, 4096), (2, 7, 7), 13)):
        for sp in (True, False):
            runs = {}
            for dt in (torch.float32, torch.float64):
                cbp = M_CBCNN.CompactBilinearPooling(c1, c2, d, sum_pool=sp)
                cbp.sparse_sketch_matrix1 = cbp.sparse_sketch_matrix1.to(dt)
                cbp.sparse_sketch_matrix2 = cbp.sparse_sketch_matrix2.to(dt)
                x1 = t(np.abs(rs_randn(1250, (b, c1, h, w))) + 0.1).to(dt).requires_grad_(True)
                x2 = t(np.abs(rs_randn(1251, (b, c2, h, w))) + 0.1).to(dt).requires_grad_(True)
                y = cbp(x1, x2)
                (y * t(rs_randn(1252, tuple(y.shape))).to(dt)).sum().backward()
                runs[dt] = (y.detach(), x1.grad.clone(), x2.grad.clone())
            (y, d1, d2), (y64, d164, d264) = runs[torch.float32], runs[torch.float64]
            k = f'{tag}_{"sum" if sp else "loc"}'
            res[f'y_{k}'], res[f'y_shape_{k}'] = sub(y, stride), np.array(y.shape)
            res[f'dx1_{k}'], res[f'dx2_{k}'] = sub(d1, stride), sub(d2, stride)
            res[f'dx1_64_{k}'], res[f'dx2_64_{k}'] = sub(d164.float(), stride), sub(d264.float(), stride)
            res[f'e32_{k}'] = np.array([rel(y, y64), rel(d1, d164), rel(d2, d264)])
            print('cbp_rect', k, 'reference fp32 vs fp64: y %.2e dx1 %.2e dx2 %.2e' % tuple(res[f'e32_{k}']))
            # the LINEAR part on its own - the sketch before the signed square root, CBCNN.py:113-130 with the module's own
            # matrices, float64 - and its gradient under a linear functional: what the kernels compute, free of the
            # conditioning of 1 / sqrt|c| at the bins that nearly cancel
            x1 = t(np.abs(rs_randn(1250, (b, c1, h, w))) + 0.1).double().requires_grad_(True)
            x2 = t(np.abs(rs_randn(1251, (b, c2, h, w))) + 0.1).double().requires_grad_(True)
            sk1 = x1.permute(0, 2, 3, 1).contiguous().view(-1, c1).mm(cbp.sparse_sketch_matrix1)
            sk2 = x2.permute(0, 2, 3, 1).contiguous().view(-1, c2).mm(cbp.sparse_sketch_matrix2)
            c = torch.fft.ifft(torch.fft.fft(sk1) * torch.fft.fft(sk2)).real.view(b, h, w, d)
            if sp:
                c = c.sum(dim=1).sum(dim=1)
            (c * t(rs_randn(1253, tuple(c.shape))).double()).sum().backward()
            res[f'c_{k}'], res[f'dc1_{k}'], res[f'dc2_{k}'] = sub(c.float(), stride), sub(x1.grad.float(), stride), sub(x2.grad.float(), stride)
            # and what ANY float32 evaluation of the identity by direct summation is away from float64 through the square
            # root (the oracle's Gram-identity formula, bins filled by index_add): the yardstick beside the FFT route's own
            h1, h2 = cbp.sparse_sketch_matrix1.abs().argmax(1), cbp.sparse_sketch_matrix2.abs().argmax(1)
            sg1 = cbp.sparse_sketch_matrix1[torch.arange(c1), h1]
            sg2 = cbp.sparse_sketch_matrix2[torch.arange(c2), h2]
            kk = ((h1[:, None] + h2[None, :]) % d).reshape(-1)
            dd = {}
            for dt in (torch.float32, torch.float64):
                a1 = t(np.abs(rs_randn(1250, (b, c1, h, w))) + 0.1).to(dt).requires_grad_(True)
                a2 = t(np.abs(rs_randn(1251, (b, c2, h, w))) + 0.1).to(dt).requires_grad_(True)
                f1 = a1.permute(0, 2, 3, 1).reshape(-1, c1) * sg1.to(dt)
                f2 = a2.permute(0, 2, 3, 1).reshape(-1, c2) * sg2.to(dt)
                prod = (f1[:, :, None] * f2[:, None, :]).reshape(f1.shape[0], -1)
                cc = torch.zeros(f1.shape[0], d, dtype=dt).index_add(1, kk, prod).view(b, h, w, d)
                if sp:
                    cc = cc.sum(dim=1).sum(dim=1)
                yy = torch.nn.functional.normalize(torch.sign(cc) * torch.sqrt(torch.abs(cc) + 1e-10))
                (yy * t(rs_randn(1252, tuple(yy.shape))).to(dt)).sum().backward()
                dd[dt] = (a1.grad.clone(), a2.grad.clone())
            res[f'd32_{k}'] = np.array([rel(dd[torch.float32][0], dd[torch.float64][0]), rel(dd[torch.float32][1], dd[torch.float64][1])])
            print('   direct float32 summation vs float64: dx1 %.2e dx2 %.2e' % tuple(res[f'd32_{k}']))
    save('cbp_rect', **res)


# ---------------------------------------------------------------- MPN-COV
def gen_mpn():
    for tag, shape, iters in (('mpn_small', (2, 16, 4, 5), (5, 3, 2, 1)),
                              ('mpn_256', (2, 256, 14, 14), (5,))):
        out = {}
        for it in iters:
            x = t(rs_relu_randn(31, shape)).requires_grad_(True)
            cov = M_MPN.Covpool.apply(x)
            cov.retain_grad()
            sq = M_MPN.Sqrtm.apply(cov, it)
            sq.retain_grad()
            tv = M_MPN.Triuvec.apply(sq)
            w = t(rs_randn(32, tuple(tv.shape)))
            (tv * w).sum().backward()
            big = shape[1] > 64
            f = sub if big else (lambda a: a)
            out.update({f'cov_it{it}': f(cov), f'sqrtm_it{it}': f(sq), f'triu_it{it}': f(tv),
                        f'dcov_it{it}': f(cov.grad), f'dsq_it{it}': f(sq.grad), f'dx_it{it}': f(x.grad),
                        f'triu_sum_it{it}': tv.double().sum(), f'dx_abs_it{it}': x.grad.double().abs().sum(),
                        f'trace0_it{it}': cov[0].trace()})
        out['triu_shape'] = np.array(tv.shape)
        save(tag, **out)


# ---------------------------------------------------------------- AP-CNN
class _Dummy:
    pass


def gen_apcnn():
    # pyramid attention with a small channel count (weights stored: tiny)
    apn = M_AP.PyramidAttentions(channel_size=32)
    sd = {k: v.clone() for k, v in apn.state_dict().items()}
    feats = [t(rs_randn(41 + i, (2, 32, s, s))).requires_grad_(True) for i, s in enumerate((28, 14, 7))]
    a3, a4, a5, s3, s4, s5 = apn(feats)
    pooled = [a.mean(dim=(2, 3)) for a in (a3, a4, a5)]
    w = [t(rs_randn(44 + i, tuple(p.shape))) for i, p in enumerate(pooled)]
    sum((p * wi).sum() for p, wi in zip(pooled, w)).backward()
    save('apcnn_apn', **{'w_' + k.replace('.', '__'): v for k, v in sd.items()},
         pooled3=pooled[0], pooled4=pooled[1], pooled5=pooled[2],
         s3=s3, s4=s4, s5=s5, df3=sub(feats[0].grad), df4=sub(feats[1].grad), df5=feats[2].grad,
         df3_abs=feats[0].grad.double().abs().sum(), df4_abs=feats[1].grad.double().abs().sum())

    # ROI selection: reference get_att_roi on random sigmoid-like masks (no ties)
    out = {}
    for ncls in (200, 8142):
        d = _Dummy()
        d.num_classes = ncls
        for lvl, (hw, stride, size, topk) in enumerate(((56, 8, 64, 5), (28, 16, 128, 3), (14, 32, 256, 1))):
            m = t(1.0 / (1.0 + np.exp(-2.0 * rs_randn(50 + lvl, (3, 1, hw, hw))))).float()
            roi = M_AP.ResNet.get_att_roi(d, m, stride, size, 448, 448, iou_thred=0.05, topk=topk)
            out[f'roi_c{ncls}_l{lvl + 3}'] = roi
    save('apcnn_roi', **out)

    # ROI crop / drop / resize (training uses python `random`; seed recorded)
    d = _Dummy()
    d.num_classes = 200
    masks = [t(1.0 / (1.0 + np.exp(-2.0 * rs_randn(50 + l, (3, 1, hw, hw))))).float()
             for l, hw in enumerate((56, 28, 14))]
    rois = [M_AP.ResNet.get_att_roi(d, m, s, a, 448, 448, iou_thred=0.05, topk=k)
            for m, (s, a, k) in zip(masks, ((8, 64, 5), (16, 128, 3), (32, 256, 1)))]
    res = {}
    for mode in ('train', 'eval'):
        d.training = mode == 'train'
        x2 = t(rs_randn(60, (3, 8, 56, 56))).requires_grad_(True)
        random.seed(7)
        # record the reference's own draw sequence so tests can inject it
        state = random.getstate()
        y, _ = M_AP.ResNet.get_roi_crop_feat(d, x2, rois, 8)
        w = t(rs_randn(61, tuple(y.shape)))
        (y * w).sum().backward()
        res[f'y_{mode}'] = sub(y, 61)
        res[f'y_abs_{mode}'] = y.double().abs().sum()
        res[f'dx_{mode}'] = sub(x2.grad, 61)
        res[f'dx_abs_{mode}'] = x2.grad.double().abs().sum()
        if mode == 'train':
            random.setstate(state)
            picks = []
            for i in range(3):
                n3 = int((rois[0][:, 0] == i).sum())
                n4 = int((rois[1][:, 0] == i).sum())
                pr = random.random()
                if pr < 0.3:
                    picks.append((3, random.randint(0, n3 - 1)))
                elif pr < 0.6:
                    picks.append((4, random.randint(0, n4 - 1)))
                else:
                    picks.append((0, -1))
            res['drops'] = np.array(picks)
    save('apcnn_crop', roi3=rois[0], roi4=rois[1], roi5=rois[2], **res)


# ---------------------------------------------------------------- OSME
def gen_osme():
    osme = M_OSME.OSME(32, 8, feature_shape=7, num_attention=2)
    sd = {k: v.clone() for k, v in osme.state_dict().items()}
    x = t(rs_relu_randn(71, (3, 32, 7, 7))).requires_grad_(True)
    f, parts = osme(x)
    w = t(rs_randn(72, tuple(parts.shape)))
    ((parts * w).sum() + f.sum()).backward()
    save('osme_small', **{'w_' + k.replace('.', '__'): v for k, v in sd.items()},
         f=f, parts=parts, dx=x.grad)


# ---------------------------------------------------------------- MAMC loss
MAMC_CASES = {          # name: (batch, parts, dim, labels)
    'balanced': (10, 2, 1024, [3, 3, 7, 7, 1, 1, 9, 9, 4, 4]),        # configs/OSMENet.yaml: 5 classes x 2 samples
    'mixed': (6, 3, 16, [2, 5, 2, 2, 0, 5]),
    'all_distinct': (4, 1, 8, [0, 1, 2, 3]),                          # no same-class pairs: two of the three terms empty
    'one_class': (3, 2, 8, [6, 6, 6]),                                # no negatives of another class
}


def gen_mamc():
    from yacs.config import CfgNode as CN
    from model.loss.MAMC_loss import MAMCLoss, NPairsLoss
    out = {}
    for i, (name, (b, p, d, labels)) in enumerate(MAMC_CASES.items()):
        x = t(rs_randn(300 + i, (b, p, d))).requires_grad_(True)
        y = torch.tensor(labels)
        loss = NPairsLoss()(x, y)
        loss.backward()
        out[name + '_loss'] = loss.detach()
        out[name + '_dx'] = x.grad.clone()
    b, p, d, labels = MAMC_CASES['balanced']
    x = t(rs_randn(300, (b, p, d))).requires_grad_(True)
    pred = t(rs_randn(310, (b, 200))).requires_grad_(True)
    total = MAMCLoss(CN(dict(lambda_a=0.5, use_mamc=True)))((pred, x), torch.tensor(labels))
    total.backward()
    out['mamc_total'] = total.detach()
    out['mamc_dpred'] = pred.grad.clone()
    out['mamc_dx'] = x.grad.clone()
    save('mamc_loss', **out)


# ---------------------------------------------------------------- CIN channel interaction
def gen_cin():
    from yacs.config import CfgNode as CN
    M_CIN = sys.modules['model.methods.CIN']
    from model.loss.CIN_loss import CINLoss
    torch.manual_seed(7)
    cim = M_CIN.ChannelInteractionModule(in_channel=24, spatial_size=(3, 4))
    with torch.no_grad():
        for i, p_ in enumerate(cim.parameters()):
            p_.copy_(t(rs_randn(400 + i, tuple(p_.shape))) * (0.05 if p_.dim() > 1 else 0.01))
    sd = {k: v.clone() for k, v in cim.state_dict().items()}
    out = {'w_' + k.replace('.', '__'): v for k, v in sd.items()}
    cim.train()
    x = t(rs_relu_randn(410, (4, 24, 3, 4))).requires_grad_(True)
    z, zc = cim(x)
    ((z * t(rs_randn(411, tuple(z.shape)))).sum() + (zc * t(rs_randn(412, tuple(zc.shape)))).sum()).backward()
    out.update(z=z, z_cci=zc, dx=x.grad.clone(),
               **{'g_' + k.replace('.', '__'): v.grad.clone() for k, v in cim.named_parameters()})
    cim.eval()
    out['z_eval'] = cim(t(rs_relu_randn(410, (4, 24, 3, 4))))
    # the criterion (its own Linear h)
    crit = CINLoss(CN(dict(alpha=2.0, beta=0.5, channel=24, feature_size=12, r_channel=8)))
    with torch.no_grad():
        crit.h.weight.copy_(t(rs_randn(420, tuple(crit.h.weight.shape))) * 0.1)
        crit.h.bias.copy_(t(rs_randn(421, tuple(crit.h.bias.shape))) * 0.1)
    logits = t(rs_randn(422, (4, 5))).requires_grad_(True)
    zc2 = t(rs_randn(423, (4, 24, 12))).requires_grad_(True)
    for name, labels in (('pairs', [1, 3, 1, 1]), ('nopairs', [1, 3, 0, 2])):
        logits.grad = zc2.grad = None
        loss = crit((logits, zc2), torch.tensor(labels))
        loss.backward()
        out[f'loss_{name}'] = loss.detach()
        out[f'loss_{name}_dlogits'] = logits.grad.clone()
        out[f'loss_{name}_dz'] = zc2.grad.clone()
    out['h_w'], out['h_b'] = crit.h.weight.detach().clone(), crit.h.bias.detach().clone()
    save('cin_small', **out)


def gen_cin_model():
    """Whole reference CIN (ResNet-50 + channel interaction + classifier) at 224x224: eval logits, train-mode outputs
    (batch statistics, contrastive branch) and the criterion value; plus its state_dict keys."""
    from yacs.config import CfgNode as CN
    from inputs import seeded_init
    from model.loss.CIN_loss import CINLoss
    M_CIN = sys.modules['model.methods.CIN']
    real_r50 = M_CIN.resnet50
    M_CIN.resnet50 = lambda pretrained=True: real_r50(pretrained=False)
    m = MODEL.get('CIN')(CN(dict(num_classes=200)))
    keys_path = os.path.join(OUT, 'state_dict_keys.json')
    keys = json.load(open(keys_path))
    keys['CIN'] = {
        'state_dict': [[k, list(v.shape)] for k, v in m.state_dict().items()],
        'children': [n for n, _ in m.named_children()],
        'n_params': sum(p.numel() for p in m.parameters()),
    }
    json.dump(keys, open(keys_path, 'w'))
    seeded_init(m, 930)
    x = t(rs_randn(931, (4, 3, 224, 224)))
    m.eval()
    with torch.no_grad():
        logits_eval = m(x)
    m.train()
    with torch.no_grad():
        logits_train, z_cci = m(x)
    torch.manual_seed(3)
    crit = CINLoss(CN(dict(alpha=2.0, beta=0.5, channel=2048, feature_size=49, r_channel=16)))
    with torch.no_grad():
        crit.h.weight.copy_(t(rs_randn(932, tuple(crit.h.weight.shape))) * 1e-3)
        crit.h.bias.zero_()
        loss = crit((logits_train, z_cci), torch.tensor([5, 9, 5, 9]))
    # a REAL backward through the plugin at its own width (C = 2048, 7 x 7): the trunk in eval mode (running statistics: no
    # batch-of-4 BatchNorm amplification in front of the head), the interaction module in train mode, the criterion on
    # (logits, Z_CCI) -> gradients of the classifier, the interaction module's conv / fc, and at the head's input.
    m.eval()
    m.ChannelInteraction.train()
    feats = []
    def keep(mod, inp, out):
        out.retain_grad()
        feats.append(out)
    hook = m.backbone.register_forward_hook(keep)
    m.zero_grad()
    lg, zc = m(x)
    hook.remove()
    crit.train()
    crit((lg, zc), torch.tensor([5, 9, 5, 9])).backward()
    grads = dict(hy_logits=lg.detach(), hy_z_cci_sub=sub(zc, 97), hy_dfeat_sub=sub(feats[0].grad, 13),
                 hy_dfeat_norm=feats[0].grad.double().norm())
    for k, v in list(m.ChannelInteraction.named_parameters()) + list(m.classifier.named_parameters()):
        key = k.replace('.', '__')
        grads['hy_g_' + key] = sub(v.grad, 1009 if v.numel() > 500000 else 7)
        grads['hy_gn_' + key] = v.grad.double().norm()
    grads['hy_g_h'] = sub(crit.h.weight.grad, 97)
    save('model_cin', logits_eval=logits_eval, logits_train=logits_train, z_cci_sub=sub(z_cci, 97), z_cci_sum=z_cci.double().sum(),
         loss=loss, **grads)


def gen_cin_448():
    """CIN at a 448 x 448 input: 14 x 14 maps, the kernels of hk_cin_sci_fwd / bwd for the larger map sizes.
    (a) the whole reference model in EVAL mode (its train mode is tied to 7 x 7 maps: CIN.py:22, the fc behind the
    contrastive branch has 2 * 2048 * 49 inputs) -> model_cin_448.npz; (b) the reference ChannelInteractionModule built for
    14 x 14 maps on 128 channels, train mode with the contrastive branch, outputs and every gradient -> cin_14x14.npz."""
    from yacs.config import CfgNode as CN
    from inputs import seeded_init
    M_CIN = sys.modules['model.methods.CIN']
    real_r50 = M_CIN.resnet50
    M_CIN.resnet50 = lambda pretrained=True: real_r50(pretrained=False)
    m = MODEL.get('CIN')(CN(dict(num_classes=200)))
    M_CIN.resnet50 = real_r50
    seeded_init(m, 930)
    m.eval()
    with torch.no_grad():
        x = t(rs_randn(941, (2, 3, 448, 448)))
        logits = m(x)
        z = m.ChannelInteraction(m.backbone(x))
    save('model_cin_448', logits_eval=logits, z_sub=sub(z, 97), z_sum=z.double().sum())

    cim = M_CIN.ChannelInteractionModule(in_channel=128, spatial_size=(14, 14))
    with torch.no_grad():
        for i, p_ in enumerate(cim.parameters()):
            p_.copy_(t(rs_randn(950 + i, tuple(p_.shape))) * (0.02 if p_.dim() > 1 else 0.01))
    cim.train()
    xm = t(rs_relu_randn(960, (4, 128, 14, 14))).requires_grad_(True)
    z, zc = cim(xm)
    ((z * t(rs_randn(961, tuple(z.shape)))).sum() + (zc * t(rs_randn(962, tuple(zc.shape)))).sum()).backward()
    out = dict(z=sub(z, 7), z_cci=sub(zc, 7), dx=sub(xm.grad, 7), dx_norm=xm.grad.double().norm())
    for k, v in cim.named_parameters():
        out['g_' + k.replace('.', '__')] = sub(v.grad, 7)
        out['gn_' + k.replace('.', '__')] = v.grad.double().norm()
    cim.eval()
    out['z_eval'] = sub(cim(t(rs_relu_randn(960, (4, 128, 14, 14)))), 7)
    save('cin_14x14', **out)


def gen_cin_2048():
    """The reference ChannelInteractionModule exactly as the plugin builds it (CIN.py:99: 2048 channels, 7 x 7 maps =
    configs/CIN.yaml:15), B = 4, train mode with the contrastive branch: Z, Z_CCI, dX and every parameter gradient
    (subsampled + norms) -> cin_2048.npz.  The backward kernels this shape dispatches walk 32 column blocks."""
    M_CIN = sys.modules['model.methods.CIN']
    cim = M_CIN.ChannelInteractionModule(in_channel=2048, spatial_size=(7, 7))
    with torch.no_grad():
        for i, (k, p_) in enumerate(cim.named_parameters()):
            scale = {'conv.weight': 0.005, 'fc.weight': 0.002}.get(k, 0.01)
            p_.copy_(t(rs_randn(970 + i, tuple(p_.shape))) * scale)
    cim.train()
    xm = t(rs_relu_randn(980, (4, 2048, 7, 7))).requires_grad_(True)
    z, zc = cim(xm)
    ((z * t(rs_randn(981, tuple(z.shape)))).sum() + (zc * t(rs_randn(982, tuple(zc.shape)))).sum()).backward()
    out = dict(z=sub(z, 7), z_cci=sub(zc, 7), dx=sub(xm.grad, 7), dx_norm=xm.grad.double().norm(),
               z_norm=z.double().norm(), z_cci_norm=zc.double().norm())
    for k, v in cim.named_parameters():
        out['g_' + k.replace('.', '__')] = sub(v.grad, 1009 if v.numel() > 500000 else 7)
        out['gn_' + k.replace('.', '__')] = v.grad.double().norm()
    cim.eval()
    out['z_eval'] = sub(cim(t(rs_relu_randn(980, (4, 2048, 7, 7)))), 7)
    save('cin_2048', **out)


# ---------------------------------------------------------------- key contracts
def gen_keys():
    from yacs.config import CfgNode as CN
    real_vgg16 = M_BCNN.vgg16
    M_BCNN.vgg16 = lambda pretrained=True: real_vgg16(pretrained=False)
    M_CBCNN.vgg16 = lambda pretrained=True: real_vgg16(pretrained=False)
    real_r50 = M_MPN.resnet50
    M_MPN.resnet50 = lambda pretrained=True: real_r50(pretrained=False)
    real_r101 = M_OSME.resnet101
    M_OSME.resnet101 = lambda pretrained=True: real_r101(pretrained=False)
    models = {
        'BCNN': MODEL.get('BCNN')(CN(dict(stage=2, num_classes=200))),
        'CBCNN': MODEL.get('CBCNN')(CN(dict(stage=2, num_classes=200, input_channel=512, output_channel=6000))),
        'MPN': MODEL.get('MPN')(CN(dict(iter_num=5, is_sqrt=True, is_vec=True, input_dim=2048,
                                        dimension_reduction=256, num_classes=200))),
        'APCNN': M_AP.resnet50(200),
        'APCNN_8142': M_AP.resnet50(8142),
        'OSMENet': MODEL.get('OSMENet')(CN(dict(num_attention=2, num_classes=200))),
    }
    keys_path = os.path.join(OUT, 'state_dict_keys.json')
    keys = json.load(open(keys_path)) if os.path.isfile(keys_path) else {}       # (gen_cin_model adds its own entry: keep it)
    for name, m in models.items():
        keys[name] = {
            'state_dict': [[k, list(v.shape)] for k, v in m.state_dict().items()],
            'children': [n for n, _ in m.named_children()],
            'n_params': sum(p.numel() for p in m.parameters()),
        }
    with open(keys_path, 'w') as f:
        json.dump(keys, f)
    print('wrote state_dict_keys.json', {k: len(v['state_dict']) for k, v in keys.items()})
    return models


# ---------------------------------------------------------------- whole-model logits
def gen_models(models):
    """Tiny end-to-end pins: 64x64 images through the reference BCNN / CBCNN / MPN with
    a seeded re-initialisation that tests/golden/inputs.py:seeded_init reproduces."""
    from inputs import seeded_init
    res = {}
    for name in ('BCNN', 'CBCNN', 'MPN'):
        m = models[name]
        seeded_init(m, 900)
        m.eval()
        x = t(rs_randn(901, (2, 3, 64, 64)))
        with torch.no_grad():
            res[name] = m(x)
    save('model_logits', **res)
    # AP-CNN (eval mode: no python-random drop) and OSMENet at 224x224
    m = models['APCNN']
    seeded_init(m, 910)
    m.eval()
    x = t(rs_randn(911, (2, 3, 224, 224)))
    with torch.no_grad():
        out_mean, out_list, mask_cat, roi_list = m(x, None)
    save('model_apcnn', out_mean=out_mean, out_list=torch.stack(out_list), mask_cat=sub(mask_cat, 13),
         roi3=roi_list[0], roi4=roi_list[1], roi5=roi_list[2])
    m = models['OSMENet']
    seeded_init(m, 920)
    m.eval()
    with torch.no_grad():
        logits, parts = m(t(rs_randn(921, (2, 3, 224, 224))))
    save('model_osme', logits=logits, parts=parts)


def gen_models_448():
    """The same three reference models at the CONFIGS' input size: two 448 x 448 images -> 14 x 14 feature maps, where the
    plugins dispatch the panel / fused kernels (bcnn_gram_panel_kernel<196>, cbp_fused_kernel<196>, the covariance panel
    kernel + nsmm chain at d = 256) and the wide-classifier kernels at their real widths (262144 / 6000 / 32896 -> 200).
    Pins eval logits, and - through a cross-entropy on targets (3, 77) - the classifier's own gradients, and the gradient
    AT THE HEAD'S INPUT as the reference's autograd produces it in-model (`*_feat_grad`: d loss / d backbone(x); for MPN
    also `MPN_dr_grad`, the gradient at the covariance's input behind the 1x1 reduction) - the pool's dX at 14 x 14 maps
    without MIOpen's backward in between.  The same model is also run in float64; the float32 run's distance from it is
    stored per tensor (`*_e32_feat_grad`) as the yardstick of what float32 can pin.
    `BCNN_S1*`: the reference BCNN with `stage=1` (BCNN.py:45-52: frozen trunk, features detached) - logits and the
    classifier's gradients; no parameter of the trunk may receive a gradient."""
    from inputs import seeded_init
    models = gen_keys()
    res = {}
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())

    def run(m, dt):
        keep = {}
        hooks = [m.backbone.register_forward_hook(lambda _m, _i, o: (o.retain_grad(), keep.__setitem__('feat', o))[0])]
        if hasattr(m, 'pool'):
            hooks.append(m.pool.conv_dr_block.register_forward_hook(
                lambda _m, _i, o: (o.retain_grad(), keep.__setitem__('dr', o))[0]))
        m = m.to(dt)
        m.zero_grad()
        x = t(rs_randn(931, (2, 3, 448, 448))).to(dt)
        y = m(x)
        torch.nn.functional.cross_entropy(y, torch.tensor([3, 77])).backward()
        for h in hooks:
            h.remove()
        return y.detach(), {k: v.grad.clone() for k, v in keep.items()}

    for name in ('BCNN', 'CBCNN', 'MPN'):
        m = models[name]
        seeded_init(m, 930)
        m.eval()
        for p_ in m.parameters():
            p_.requires_grad_(True)
        y, g32 = run(m, torch.float32)
        res[name] = y
        res[name + '_cls_w_grad'] = sub(m.classifier.weight.grad, 1009)
        res[name + '_cls_w_grad_abs'] = m.classifier.weight.grad.abs().sum().reshape(1)
        res[name + '_cls_b_grad'] = m.classifier.bias.grad.clone()
        w0 = next(m.backbone.parameters())
        res[name + '_conv0_grad'] = w0.grad.clone()
        res[name + '_feat_grad'] = sub(g32['feat'], 7)
        res[name + '_feat_grad_abs'] = g32['feat'].double().abs().sum().reshape(1)
        if 'dr' in g32:
            res[name + '_dr_grad'] = sub(g32['dr'], 3)
            res[name + '_dr_grad_abs'] = g32['dr'].double().abs().sum().reshape(1)
        if name == 'CBCNN':                        # sketch matrices are plain attributes: cast by hand
            m.bilinear_pooling.sparse_sketch_matrix1 = m.bilinear_pooling.sparse_sketch_matrix1.double()
            m.bilinear_pooling.sparse_sketch_matrix2 = m.bilinear_pooling.sparse_sketch_matrix2.double()
        y64, g64 = run(m, torch.float64)
        res[name + '_e32_feat_grad'] = np.array([rel(g32['feat'], g64['feat'])])
        res[name + '_e32_logits'] = np.array([rel(y, y64)])
        res[name + '_feat_grad64'] = sub(g64['feat'], 7).float()
        if 'dr' in g32:
            res[name + '_e32_dr_grad'] = np.array([rel(g32['dr'], g64['dr'])])
        print(name, 'fp32 reference vs fp64 reference: logits', rel(y, y64), 'feat grad', rel(g32['feat'], g64['feat']),
              ('dr grad %g' % rel(g32['dr'], g64['dr'])) if 'dr' in g32 else '')
        models[name] = None
        del m

    # BCNN stage 1 (BASELINE configs[0]): BCNN.py:45-52
    from yacs.config import CfgNode as CN
    m = MODEL.get('BCNN')(CN(dict(stage=1, num_classes=200)))
    seeded_init(m, 930)
    m.eval()
    y = m(t(rs_randn(931, (2, 3, 448, 448))))
    torch.nn.functional.cross_entropy(y, torch.tensor([3, 77])).backward()
    assert all(p_.grad is None for p_ in m.backbone.parameters())
    res['BCNN_S1'] = y.detach()
    res['BCNN_S1_cls_w_grad'] = sub(m.classifier.weight.grad, 1009)
    res['BCNN_S1_cls_w_grad_abs'] = m.classifier.weight.grad.abs().sum().reshape(1)
    res['BCNN_S1_cls_b_grad'] = m.classifier.bias.grad.clone()
    save('model_logits_448', **res)


APCNN_448_TRAIN_BATCH = 4


def gen_apcnn_448():
    """AP-CNN as configs[4] runs it: 448 x 448 input, num_classes = 8142 -> hidden_num = 256 (APCNN.py:360-363) and the
    0.1 - 0.9 border band of get_att_roi (APCNN.py:451-455).  Eval mode on two images (out_mean, the 8 logits, mask_cat,
    the three ROI tables: APCNN.py:540-599) and train mode at batch 4 with the python-`random` drop block seeded (as
    gen_apcnn_train: pinned in float64, the float32 run's own distance stored as the yardstick)."""
    import random
    from inputs import seeded_init
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    m = M_AP.resnet50(8142)
    seeded_init(m, 940)
    m.eval()
    with torch.no_grad():
        out_mean, out_list, mask_cat, roi_list = m(t(rs_randn(941, (2, 3, 448, 448))), None)
    res = dict(out_mean=sub(out_mean, 1), out_list=torch.stack(out_list), mask_cat=sub(mask_cat, 13),
               roi3=roi_list[0], roi4=roi_list[1], roi5=roi_list[2])
    print('eval rois', [tuple(r.shape) for r in roi_list], 'argmax', out_mean.argmax(1).tolist())
    runs = {}
    nb = APCNN_448_TRAIN_BATCH
    for dt in (torch.float64, torch.float32):
        m = M_AP.resnet50(8142)
        seeded_init(m, 940)
        m = m.to(dt).train()
        x = t(rs_randn(942, (nb, 3, 448, 448))).to(dt)
        wt = t(rs_randn(943, (nb, 8142))).to(dt)
        random.seed(5)
        om, ol, _mc, rl = m(x, None)
        (om * wt).sum().backward()
        runs[dt] = (om.detach(), torch.stack(ol).detach(), rl,
                    {n: p.grad for n, p in m.named_parameters() if p.grad is not None})
    om, ol, rois, grads = runs[torch.float64]
    om32, ol32, rois32, grads32 = runs[torch.float32]
    assert all(torch.equal(a[:, :5].float(), b[:, :5]) for a, b in zip(rois, rois32)), 'fp32 / fp64 picked different ROIs'
    keep = ['conv1.weight', 'layer2.0.conv1.weight', 'layer4.2.conv3.weight', 'cls_concate.1.weight', 'cls3.6.weight']
    stride = lambda k: max(7, grads[k].numel() // 2000 | 1)
    res.update(t_out_mean=om, t_out_list=ol, t_roi3=rois[0].float(), t_roi4=rois[1].float(), t_roi5=rois[2].float(),
               t_e32_out_list=np.array([rel(ol32[i], ol[i]) for i in range(8)]), t_e32_out_mean=np.array([rel(om32, om)]),
               t_grad_names=np.array(keep), **{'t_g%d' % i: sub(grads[k], stride(k)) for i, k in enumerate(keep)},
               **{'t_gn%d' % i: grads[k].norm().reshape(1) for i, k in enumerate(keep)},
               t_e32_g=np.array([rel(grads32[k], grads[k]) for k in keep]))
    print('train: fp32 reference vs fp64 reference: out_list', [rel(ol32[i], ol[i]) for i in range(8)],
          'grads', [rel(grads32[k], grads[k]) for k in keep])
    save('model_apcnn_448', **res)


APCNN_TRAIN_BATCH = 8


def gen_apcnn_train():
    """AP-CNN in TRAIN mode (BatchNorm on batch statistics, the python-`random` drop block of APCNN.py:485-504 with a
    fixed seed): logits of both stages and a few gradients of  sum(out_mean * wt)  - the whole two-stage model with the
    ROI refinement's backward, as the reference's own autograd computes it.
    BatchNorm over a batch of TWO samples divides by the difference of two nearly equal pooled features: rounding
    differences of 1e-7 in a feature come out as 1e-3 in a logit.  The reference is therefore run twice, in float64
    (the pinned values) and in float32 (what it actually executes); the float32 run's own distance from the float64
    one is stored per tensor (`e32_*`) and is the yardstick of the test's tolerances."""
    import random
    from inputs import seeded_init
    runs = {}
    for dt in (torch.float64, torch.float32):
        m = M_AP.resnet50(200)
        seeded_init(m, 910)
        m = m.to(dt).train()
        x = t(rs_randn(911, (APCNN_TRAIN_BATCH, 3, 224, 224))).to(dt)
        wt = t(rs_randn(912, (APCNN_TRAIN_BATCH, 200))).to(dt)
        random.seed(3)
        out_mean, out_list, mask_cat, roi_list = m(x, None)
        (out_mean * wt).sum().backward()
        runs[dt] = (out_mean.detach(), torch.stack(out_list).detach(), roi_list,
                    {n: p.grad for n, p in m.named_parameters() if p.grad is not None})
    om, ol, rois, grads = runs[torch.float64]
    om32, ol32, rois32, grads32 = runs[torch.float32]
    assert all(torch.equal(a[:, :5].float(), b[:, :5]) for a, b in zip(rois, rois32)), 'fp32 / fp64 picked different ROIs'
    keep = ['conv1.weight', 'layer2.0.conv1.weight', 'layer4.2.conv3.weight', 'cls_concate.1.weight']
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    stride = lambda k: max(7, grads[k].numel() // 2000 | 1)
    save('model_apcnn_train', out_mean=om, out_list=ol, roi3=rois[0].float(), roi4=rois[1].float(), roi5=rois[2].float(),
         e32_out_list=np.array([rel(ol32[i], ol[i]) for i in range(8)]), e32_out_mean=np.array([rel(om32, om)]),
         grad_names=np.array(keep), **{'g%d' % i: sub(grads[k], stride(k)) for i, k in enumerate(keep)},
         **{'gn%d' % i: grads[k].norm().reshape(1) for i, k in enumerate(keep)},
         e32_g=np.array([rel(grads32[k], grads[k]) for k in keep]))
    print('fp32 reference vs fp64 reference: out_list', [rel(ol32[i], ol[i]) for i in range(8)],
          'grads', [rel(grads32[k], grads[k]) for k in keep])


# ---------------------------------------------------------------- full BASELINE dispatch shapes
# The shapes the BASELINE configs actually launch (per-GPU batch 64 / 16): the kernels pick other tilings, grids and
# queue counts there than at the B = 2 pins above, so the reference is run at the full batch and pinned per sample.
FULL_PICK = [0, 7, 15, 31, 32, 47, 63]            # first / middle / last sample of each half (two-queue dispatch), and of the first 16
FULL_PICK16 = [0, 7, 15]                          # AP-CNN (batch 16)


def gen_full():
    out = {'pick': np.array(FULL_PICK), 'pick16': np.array(FULL_PICK16)}
    per = lambda a: a.double().reshape(a.shape[0], -1)

    # Fast MPN-COV head of configs/MPN.yaml at the benchmark batch: covariance (C = 256), Newton-Schulz (d = 256, 5
    # iterations), triuvec, and the whole backward  (MPNCOV.py:105-230)
    x = t(rs_relu_randn(3101, (64, 256, 14, 14))).requires_grad_(True)
    cov = M_MPN.Covpool.apply(x)
    cov.retain_grad()
    sq = M_MPN.Sqrtm.apply(cov, 5)
    sq.retain_grad()
    tv = M_MPN.Triuvec.apply(sq)
    (tv * t(rs_randn(3102, tuple(tv.shape)))).sum().backward()
    out.update(mpn_cov_sum=per(cov).sum(1), mpn_cov_abs=per(cov).abs().sum(1), mpn_sq_sum=per(sq).sum(1),
               mpn_sq_abs=per(sq).abs().sum(1), mpn_tv_sum=per(tv).sum(1), mpn_dcov_abs=per(cov.grad).abs().sum(1),
               mpn_dx_sum=per(x.grad).sum(1), mpn_dx_abs=per(x.grad).abs().sum(1))
    for s_ in FULL_PICK:
        out[f'mpn_cov_{s_}'] = sub(cov[s_], 61)
        out[f'mpn_sq_{s_}'] = sub(sq[s_], 61)
        out[f'mpn_dcov_{s_}'] = sub(cov.grad[s_], 61)
        out[f'mpn_dx_{s_}'] = sub(x.grad[s_], 61)
    print('mpn full done')

    # compact bilinear pooling of configs/CBCNN_S2.yaml, D = 6000, 64 samples (the yaml batch of 16 = the first 16);
    # the reference's FFT route is per-sample independent: run in slices of 4 to bound its [B*196, 6000] complex buffers
    cbp = M_CBCNN.CompactBilinearPooling(512, 512, 6000)
    xn, wn = rs_relu_randn(3201, (64, 512, 14, 14)), rs_randn(3202, (64, 6000))
    # The gradient passes through dc = du / (2 sqrt(|c| + 1e-10)): round-off of the fp32 FFTs in the small bins is
    # amplified, and the reference's own float32 gradient is ~1e-4 away from its float64 one.  So the reference is run
    # twice - float32 (what it executes; pins y) and float64 (same module, sketch matrices cast; pins dX) - and the
    # float32 run's distance from the float64 one is stored as the yardstick (cbp_e32_*), like gen_apcnn_train.
    runs = {}
    for dt in (torch.float32, torch.float64):
        cbp.sparse_sketch_matrix1 = cbp.sparse_sketch_matrix1.to(dt)
        cbp.sparse_sketch_matrix2 = cbp.sparse_sketch_matrix2.to(dt)
        ys, dxs = [], []
        for i in range(0, 64, 4):
            xs = t(xn[i:i + 4]).to(dt).requires_grad_(True)
            y = cbp(xs)
            (y * t(wn[i:i + 4]).to(dt)).sum().backward()
            ys.append(y.detach())
            dxs.append(xs.grad)
        runs[dt] = (torch.cat(ys), torch.cat(dxs))
    (y, dx), (y64, dx64) = runs[torch.float32], runs[torch.float64]
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    out.update(cbp_y_sum=per(y).sum(1), cbp_y_abs=per(y).abs().sum(1), cbp_dx64_abs=per(dx64).abs().sum(1),
               cbp_e32_dx=np.array([rel(dx[i], dx64[i]) for i in range(64)]),
               cbp_e32_y=np.array([rel(y[i], y64[i]) for i in range(64)]))
    print('cbp: fp32 reference vs fp64 reference, dx: max', out['cbp_e32_dx'].max(), 'y: max', out['cbp_e32_y'].max())
    for s_ in FULL_PICK:
        out[f'cbp_y_{s_}'] = y[s_]
        out[f'cbp_dx_{s_}'] = sub(dx[s_], 61)
        out[f'cbp_dx64_{s_}'] = sub(dx64[s_], 61).float()
    # Conditioning of the gradient per sample (a YARDSTICK for the tolerance, not a pin): a bin c_k is a signed sum of
    # ~44 Gram entries; summed in float32 by ANY route it carries an error ~ eps32 * S_k (S_k = sum of their
    # magnitudes), and dc_k = du_k / (2 sqrt|c_k|) turns that into a relative change eps32 * S_k / (2 |c_k|) of dc_k -
    # unbounded as a bin cancels.  dX is linear in dc, so the first-order float32 error of dX is
    # |L(dc * eps32 S / 2|c| * xi)| / |L(dc)| with random signs xi; evaluated in float64 through the count-sketch identity.
    # Of the 64 samples here four have a bin that cancels to 1e-2 .. 4e-1 (the reference's own float32 run is 1e-2 .. 6e-1
    # off on exactly those), the median is 5e-4.
    import hawkeye_oracle as O_
    h1, s1, h2, s2 = O_.sketch_hashes(512, 512, 6000)
    assert (torch.from_numpy(h1) == cbp.sparse_sketch_matrix1.abs().argmax(dim=1)).all()     # the reference's own hashes
    bins = ((torch.from_numpy(h1)[:, None] + torch.from_numpy(h2)[None, :]) % 6000).reshape(-1)
    sign = (torch.from_numpy(s1)[:, None] * torch.from_numpy(s2)[None, :]).double().reshape(-1)
    conds = []
    for i in range(0, 64, 8):
        X = t(xn[i:i + 8]).double().reshape(8, 512, 196)
        Gm = torch.bmm(X, X.transpose(1, 2)).reshape(8, -1)
        c = torch.zeros(8, 6000, dtype=torch.float64).index_add(1, bins, Gm * sign).requires_grad_(True)
        S = torch.zeros(8, 6000, dtype=torch.float64).index_add(1, bins, Gm.abs())
        u = torch.sign(c) * torch.sqrt(c.abs() + 1e-10)
        (torch.nn.functional.normalize(u) * t(wn[i:i + 8]).double()).sum().backward()
        dc = c.grad
        xi = t(np.random.RandomState(3203 + i).choice([-1.0, 1.0], size=(8, 6000)))
        ddc = dc * (2.0 ** -24 * S / (2 * (c.detach().abs() + 1e-10))) * xi
        lin = lambda v: torch.bmm((v[:, bins] * sign).reshape(8, 512, 512) + (v[:, bins] * sign).reshape(8, 512, 512).transpose(1, 2), X)
        conds.append(lin(ddc).reshape(8, -1).norm(dim=1) / lin(dc).reshape(8, -1).norm(dim=1))
    out['cbp_cond'] = torch.cat(conds)
    print('cbp: conditioning estimate max', float(out['cbp_cond'].max()), 'median', float(out['cbp_cond'].median()),
          '; reference fp32 error / (1e-4 + 2 cond): max', float((torch.from_numpy(out['cbp_e32_dx']) / (1e-4 + 2 * out['cbp_cond'])).max()))
    print('cbp full done')

    # AP-CNN at the yaml batch (16) with the iNat2018 class count (8142: border band 0.1 - 0.9, APCNN.py:451-455):
    # pyramid attention + GAP on 256-channel maps, ROI selection, ROI crop / drop / resize of the 512 x 56 x 56 map
    torch.manual_seed(11)
    apn = M_AP.PyramidAttentions(channel_size=256)
    with torch.no_grad():
        for i, (k, p_) in enumerate(apn.named_parameters()):
            if k.endswith('_1.conv.weight'):            # spatial gates (ConvTranspose2d [256,1,3,3]): keep the sigmoid unsaturated
                p_.copy_(t(rs_randn(3300 + i, tuple(p_.shape))) * np.float32(0.1 / 256 ** 0.5))
    for k, v in apn.state_dict().items():
        out['apn_w_' + k.replace('.', '__')] = v.clone()
    feats = [t(rs_randn(3310 + i, (16, 256, s, s))).requires_grad_(True) for i, s in enumerate((56, 28, 14))]
    a3, a4, a5, s3, s4, s5 = apn(feats)
    pooled = [a.mean(dim=(2, 3)) for a in (a3, a4, a5)]
    sum((p_ * t(rs_randn(3320 + i, tuple(p_.shape)))).sum() for i, p_ in enumerate(pooled)).backward()
    for lvl, (p_, m_, f_) in enumerate(zip(pooled, (s3, s4, s5), feats)):
        out[f'apn_pooled{lvl + 3}'] = p_
        out[f'apn_mask{lvl + 3}_sum'] = per(m_).sum(1)
        out[f'apn_df{lvl + 3}_abs'] = per(f_.grad).abs().sum(1)
        for s_ in FULL_PICK16:
            out[f'apn_df{lvl + 3}_{s_}'] = sub(f_.grad[s_], 211)
    d = _Dummy()
    d.num_classes = 8142
    masks = [t(1.0 / (1.0 + np.exp(-2.0 * rs_randn(3330 + l, (16, 1, hw, hw))))).float() for l, hw in enumerate((56, 28, 14))]
    rois = [M_AP.ResNet.get_att_roi(d, m_, s_, a_, 448, 448, iou_thred=0.05, topk=k_)
            for m_, (s_, a_, k_) in zip(masks, ((8, 64, 5), (16, 128, 3), (32, 256, 1)))]
    out.update(ap_roi3=rois[0], ap_roi4=rois[1], ap_roi5=rois[2])
    for mode in ('train', 'eval'):
        d.training = mode == 'train'
        x2 = t(rs_randn(3340, (16, 512, 56, 56))).requires_grad_(True)
        random.seed(13)
        state = random.getstate()
        yc, _ = M_AP.ResNet.get_roi_crop_feat(d, x2, rois, 8)
        (yc * t(rs_randn(3341, tuple(yc.shape)))).sum().backward()
        out[f'ap_y_abs_{mode}'] = per(yc).abs().sum(1)
        out[f'ap_dx_abs_{mode}'] = per(x2.grad).abs().sum(1)
        for s_ in FULL_PICK16:
            out[f'ap_y_{mode}_{s_}'] = sub(yc[s_], 211)
            out[f'ap_dx_{mode}_{s_}'] = sub(x2.grad[s_], 211)
        if mode == 'train':                       # the reference's own draw sequence (APCNN.py:494-504), replayed
            random.setstate(state)
            picks = []
            for i in range(16):
                n3, n4 = int((rois[0][:, 0] == i).sum()), int((rois[1][:, 0] == i).sum())
                pr = random.random()
                if pr < 0.3:
                    picks.append((3, random.randint(0, n3 - 1)))
                elif pr < 0.6:
                    picks.append((4, random.randint(0, n4 - 1)))
                else:
                    picks.append((0, -1))
            out['ap_drops'] = np.array(picks)
    save('full_shapes', **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:                       # regenerate selected fixtures only, e.g. `gen_golden.py mamc`
        for which in sys.argv[1:]:
            globals()['gen_' + which]()
        sys.exit(0)
    gen_mamc()
    gen_bcnn()
    gen_bcnn_ssqrt()
    gen_cbp()
    gen_mpn()
    gen_apcnn()
    gen_osme()
    ms = gen_keys()
    gen_models(ms)
