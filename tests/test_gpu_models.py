"""GPU: the registered plugins end to end vs logits of the REFERENCE models
(tests/golden/model_*.npz: reference run on CPU with `seeded_init` weights).
Backbones run on MIOpen, so the tolerance covers conv-algorithm differences of a
50-100 layer fp32 trunk; argmax must match exactly (north_star)."""
import os

import numpy as np
import pytest
import torch

from inputs import rs_randn, seeded_init, sub

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


def load(name):
    return np.load(os.path.join(G, name + '.npz'))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = (b.detach().cpu() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).double().reshape(-1)
    return float((a - b).norm() / b.norm())


def build(name, **kw):
    import hawkeye_amd.model  # noqa: F401
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.model.registry import MODEL
    return MODEL.get(name)(CfgNode(dict(name=name, **kw)))


# AP-CNN train-mode bounds, in units of the reference's OWN float32-vs-float64 distance per tensor (train-mode BatchNorm
# amplifies rounding).  Measured on the MI355X (MIOpen convolutions, another summation order; profiles/r5_parity_edges.log):
# worst 1.52 at 224 x 224 / batch 8 / 200 classes, 5.43 at 448 x 448 / batch 4 / 8142 classes (torch CPU: 1.06 / 1.09);
# the bounds are about twice that (round 4 allowed 20).  Round 6 (test_apcnn_train_distance_is_the_trunks_not_the_heads,
# profiles/r6_gpu_tests_mid.txt): the same model on the same device with the attention pooling and the ROI crop / resize on
# torch's own ops measures 9.1 where the hk kernels measure 5.2 (7.3 in another run of the same session: MIOpen's weight
# gradients are atomic sums) - the excess over torch-CPU is the convolution library's summation order under train-mode
# BatchNorm, not the hand-written heads.
APCNN_TRAIN_K = 4.0
APCNN_TRAIN_K_448 = 12.0

CFG = {
    'BCNN': dict(stage=2, num_classes=200),
    'CBCNN': dict(stage=2, num_classes=200, input_channel=512, output_channel=6000),
    'MPN': dict(iter_num=5, is_sqrt=True, is_vec=True, input_dim=2048, dimension_reduction=256, num_classes=200),
}


@pytest.mark.parametrize('name', ['BCNN', 'CBCNN', 'MPN'])
def test_logits_match_reference(name):
    g = load('model_logits')
    m = build(name, **CFG[name])
    seeded_init(m, 900)
    m = m.to(DEV).eval()
    with torch.no_grad():
        y = m(t(rs_randn(901, (2, 3, 64, 64))).to(DEV))
    assert rel(y, g[name]) < 1e-4
    assert y.argmax(1).cpu().tolist() == g[name].argmax(1).tolist()


@pytest.mark.parametrize('name', ['BCNN', 'CBCNN', 'MPN', 'BCNN-channels_last', 'CBCNN-channels_last'])
def test_models_at_config_input_size_vs_reference(name):
    """Whole plugins at the CONFIGS' input size (two 448 x 448 images -> 14 x 14 maps): here the heads dispatch what the
    benchmarked configs dispatch - bcnn_gram_panel_kernel<196> / gram_bwd3_kernel, cbp_fused_kernel<196> / cbp_bwd3_kernel,
    the covariance panel kernel + Newton-Schulz chain at d = 256 - and the classifier runs on linear_skinny_kernel /
    linear_bwd64_kernel at its real width (262144 / 6000 / 32896 -> 200).  Against the REFERENCE models on the same seeded
    weights and images (tests/golden/model_logits_448.npz, oracle/gen_golden.py::gen_models_448): eval logits 1e-4 +
    argmax; through a cross-entropy, the classifier's gradients 1e-4; and the gradient AT THE HEAD'S INPUT - the pool's dX
    as the reference's autograd produces it in-model, before any MIOpen backward (d loss / d backbone(x); for MPN also
    behind the 1x1 reduction, at the covariance's input) - 1e-4 (CBCNN: the reference's own float32 run is 4.8e-5 from
    its float64 one there - sqrt'(|c|) of small bins - so its bound is 1e-4 + 2 e32).  The first convolution's gradient
    (behind MIOpen's whole backward) is printed and only bounded against gross disagreement."""
    g = load('model_logits_448')
    # `-channels_last`: the memory format bench.py trains in - there the VGG trunk runs the fused epilogues behind its
    # convolutions (csrc/trunk.hip, ConvStack), so this is the benchmarked path against the reference's numbers
    name, _, fmt = name.partition('-')
    m = build(name, **CFG[name])
    seeded_init(m, 930)
    m = m.to(DEV).eval()
    x = t(rs_randn(931, (2, 3, 448, 448))).to(DEV)
    if fmt:
        m, x = m.to(memory_format=torch.channels_last), x.contiguous(memory_format=torch.channels_last)
    keep = {}
    m.backbone.register_forward_hook(lambda _m, _i, o: (o.retain_grad(), keep.__setitem__('feat', o))[0])
    if name == 'MPN':
        m.pool.conv_dr_block.register_forward_hook(lambda _m, _i, o: (o.retain_grad(), keep.__setitem__('dr', o))[0])
    y = m(x)
    if fmt:
        assert keep['feat'].is_contiguous(memory_format=torch.channels_last)
    assert rel(y, g[name]) < 1e-4, rel(y, g[name])
    assert y.argmax(1).cpu().tolist() == g[name].argmax(1).tolist()
    torch.nn.functional.cross_entropy(y, torch.tensor([3, 77], device=DEV)).backward()
    gw = m.classifier.weight.grad
    assert rel(gw.reshape(-1)[::1009], g[name + '_cls_w_grad']) < 1e-4
    assert abs(float(gw.abs().sum()) - float(g[name + '_cls_w_grad_abs'][0])) < 1e-4 * float(g[name + '_cls_w_grad_abs'][0])
    assert rel(m.classifier.bias.grad, g[name + '_cls_b_grad']) < 1e-4
    # head boundary: the pool's own dX, in-model
    fg = keep['feat'].grad
    e32 = float(g[name + '_e32_feat_grad'][0])
    ef, ef64 = rel(sub(fg.cpu(), 7), g[name + '_feat_grad']), rel(sub(fg.cpu(), 7), g[name + '_feat_grad64'])
    print(f'[448 {name} {fmt}] head-input gradient vs reference: {ef:.2e} (vs its float64 run {ef64:.2e}; reference fp32 vs fp64 {e32:.2e})')
    assert ef < (1e-4 + 2 * e32 if name == 'CBCNN' else 1e-4), ef
    assert abs(float(fg.double().abs().sum()) / float(g[name + '_feat_grad_abs'][0]) - 1) < 1e-4
    if name == 'MPN':
        ed = rel(sub(keep['dr'].grad.cpu(), 3), g['MPN_dr_grad'])
        print(f'[448 MPN] covariance-input gradient vs reference: {ed:.2e}')
        assert ed < 1e-4, ed
        assert abs(float(keep['dr'].grad.double().abs().sum()) / float(g['MPN_dr_grad_abs'][0]) - 1) < 1e-4
    w0 = next(m.backbone.parameters())
    e0 = rel(w0.grad, g[name + '_conv0_grad'])
    print(f'[448 {name} {fmt}] first-conv gradient vs reference: {e0:.2e}')
    assert e0 < 2e-2, e0


def test_bcnn_stage1_at_config_input_size_vs_reference():
    """BASELINE configs[0] (BCNN_S1: frozen trunk, features detached, BCNN.py:45-52) at 448 x 448 against the reference
    model built with stage=1: logits, and the classifier's gradients - the dW / db-only backward of the classifier
    kernel (hk_linear_bwd with dx = NULL, MODE 2) in-model; no trunk parameter may get a gradient and the pool's
    backward must not run (its input does not require grad)."""
    g = load('model_logits_448')
    m = build('BCNN', stage=1, num_classes=200)
    seeded_init(m, 930)
    m = m.to(DEV).eval()
    y = m(t(rs_randn(931, (2, 3, 448, 448))).to(DEV))
    assert rel(y, g['BCNN_S1']) < 1e-4
    assert y.argmax(1).cpu().tolist() == g['BCNN_S1'].argmax(1).tolist()
    torch.nn.functional.cross_entropy(y, torch.tensor([3, 77], device=DEV)).backward()
    assert all(p.grad is None for p in m.backbone.parameters())
    gw = m.classifier.weight.grad
    assert rel(gw.reshape(-1)[::1009], g['BCNN_S1_cls_w_grad']) < 1e-4
    assert abs(float(gw.abs().sum()) / float(g['BCNN_S1_cls_w_grad_abs'][0]) - 1) < 1e-4
    assert rel(m.classifier.bias.grad, g['BCNN_S1_cls_b_grad']) < 1e-4


def test_bcnn_signed_sqrt_whole_model():
    """BCNN with the reference's OTHER normalisation (BCNN.py:23-24 enabled: `bilinear_pooling.signed_sqrt = True`) end to
    end: the plugin then runs pooling + classifier as one node with the l2 scale folded into the classifier
    (model/utils.py::pooled_classifier -> F.ssqrt_pool_linear).  Same 28 state_dict keys as the default model;
    logits and every gradient equal to the two-node composition (signed-sqrt pooling kernel, then torch's nn.Linear)
    on the same weights; and the pooled vector of that composition against the reference's own source with the two
    lines swapped in (tests/golden/bcnn_ssqrt_small.npz pins the pooling itself in test_gpu_parity)."""
    import json
    from emu.harness import restore_wide_linear, set_wide_linear
    keys = json.load(open(os.path.join(G, 'state_dict_keys.json')))['BCNN']['state_dict']
    m = build('BCNN', stage=2, num_classes=12)
    m.bilinear_pooling.signed_sqrt = True
    assert [k for k, _ in keys] == list(m.state_dict().keys())
    seeded_init(m, 950)
    m = m.to(DEV).eval()
    x = t(rs_randn(951, (3, 3, 64, 64))).to(DEV)
    tgt = torch.tensor([1, 7, 4], device=DEV)
    res = []
    for kernel in (False, True):                       # torch Linear on the kernel's pooled vector / the fused node
        saved = set_wide_linear(kernel)
        m.zero_grad()
        y = m(x)
        torch.nn.functional.cross_entropy(y, tgt).backward()
        res.append((y.detach().clone(), m.classifier.weight.grad.clone(), m.classifier.bias.grad.clone(),
                    next(m.backbone.parameters()).grad.clone()))
        restore_wide_linear(saved)
    (y0, w0, b0, c0), (y1, w1, b1, c1) = res
    assert rel(y1, y0) < 1e-5 and y1.argmax(1).tolist() == y0.argmax(1).tolist()
    assert rel(w1, w0) < 1e-5 and rel(b1, b0) < 1e-5
    e = rel(c1, c0)
    print(f'[BCNN signed sqrt] fused node vs pooling kernel + nn.Linear: first-conv gradient {e:.2e}')
    assert e < 2e-2, e                                # (behind MIOpen's backward; 2 x 2 maps: the signed sqrt's slope at G ~ 0)


def test_pyramid_attentions_module_vs_reference_golden():
    """SURVEY row A7 at MODULE level: the plugin's PyramidAttentions (spatial gate on MIOpen, hk_att_pool, channel gates
    averaged bottom-up, pooled = sgap + a_c * gap) with the REFERENCE's weights, against what the reference's
    PyramidAttentions + GAP returned (tests/golden/apcnn_apn.npz): pooled vectors, attention masks, and the gradient
    w.r.t. all three pyramid levels (APCNN.py:236-268,561-563)."""
    from hawkeye_amd.model.methods.APCNN import PyramidAttentions
    g = load('apcnn_apn')
    apn = PyramidAttentions(channel_size=32)
    apn.load_state_dict({k[2:].replace('__', '.'): t(g[k]) for k in g.files if k.startswith('w_')}, strict=True)
    apn = apn.to(DEV)
    feats = [t(rs_randn(41 + i, (2, 32, s, s))).to(DEV).requires_grad_(True) for i, s in enumerate((28, 14, 7))]
    pooled, gaps, masks = apn(feats)
    ws = [t(rs_randn(44 + i, tuple(p.shape))).to(DEV) for i, p in enumerate(pooled)]
    sum((p * wi).sum() for p, wi in zip(pooled, ws)).backward()
    for p, k in zip(pooled, ('pooled3', 'pooled4', 'pooled5')):
        assert rel(p, g[k]) < 1e-5, k
    for m, k in zip(masks, ('s3', 's4', 's5')):
        assert rel(m, g[k]) < 1e-5, k
    for f, gk in zip(gaps, feats):                           # (means of zero-mean maps: cancellation, hence 1e-5)
        assert rel(f, gk.detach().double().mean(dim=(2, 3))) < 1e-5
    assert rel(feats[2].grad, g['df5']) < 1e-4
    assert rel(sub(feats[0].grad.cpu()), g['df3']) < 1e-4 and rel(sub(feats[1].grad.cpu()), g['df4']) < 1e-4
    assert abs(float(feats[0].grad.double().abs().sum()) / float(g['df3_abs']) - 1) < 1e-5
    assert abs(float(feats[1].grad.double().abs().sum()) / float(g['df4_abs']) - 1) < 1e-5


def test_apcnn_eval_matches_reference():
    g = load('model_apcnn')
    m = build('APCNN', num_classes=200)
    seeded_init(m, 910)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out_mean, out_list, mask_cat, rois = m(t(rs_randn(911, (2, 3, 224, 224))).to(DEV), None)
    # ROI pyramid: same cells picked, scores to fp32 conv tolerance
    for got, key in zip(rois, ('roi3', 'roi4', 'roi5')):
        got = got.cpu().numpy()
        assert got.shape == g[key].shape
        np.testing.assert_array_equal(got[:, :5], g[key][:, :5])
        np.testing.assert_allclose(got[:, 5], g[key][:, 5], rtol=1e-4)
    np.testing.assert_allclose(sub(mask_cat.cpu(), 13).numpy(), g['mask_cat'], rtol=1e-3, atol=1e-5)
    assert rel(torch.stack(out_list), g['out_list']) < 1e-4
    assert rel(out_mean, g['out_mean']) < 1e-4
    assert out_mean.argmax(1).cpu().tolist() == g['out_mean'].argmax(1).tolist()


def test_osmenet_eval_matches_reference():
    g = load('model_osme')
    m = build('OSMENet', num_attention=2, num_classes=200)
    seeded_init(m, 920)
    m = m.to(DEV).eval()
    with torch.no_grad():
        logits, parts = m(t(rs_randn(921, (2, 3, 224, 224))).to(DEV))
    assert rel(logits, g['logits']) < 1e-4 and rel(parts, g['parts']) < 1e-4
    assert logits.argmax(1).cpu().tolist() == g['logits'].argmax(1).tolist()


@pytest.mark.parametrize('name,size', [('BCNN', 128), ('CBCNN', 128), ('MPN', 128), ('APCNN', 224), ('OSMENet', 224)])
def test_train_step_runs_and_updates(name, size):
    kw = dict(CFG.get(name, {}))
    if name == 'APCNN':
        kw = dict(num_classes=200)
    if name == 'OSMENet':
        kw = dict(num_attention=2, num_classes=200)
    torch.manual_seed(0)
    m = build(name, **kw).to(DEV).train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9)
    x = torch.randn(4, 3, size, size, device=DEV)
    y = torch.randint(0, 200, (4,), device=DEV)
    before = [p.detach().clone() for p in m.parameters()]
    if name == 'APCNN':
        _, out_list, _, _ = m(x, y)
        loss = sum(torch.nn.functional.cross_entropy(o, y) for o in out_list)
    elif name == 'OSMENet':
        loss = torch.nn.functional.cross_entropy(m(x)[0], y)
    else:
        loss = torch.nn.functional.cross_entropy(m(x), y, label_smoothing=0.1)
    opt.zero_grad()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
    opt.step()
    assert torch.isfinite(loss)
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, m.parameters()))
    assert changed > 0.9 * len(before)


def test_apcnn_train_mode_matches_reference():
    """AP-CNN in TRAIN mode against the reference (tests/golden/model_apcnn_train.npz, oracle/gen_golden.py
    gen_apcnn_train; batch 8): BatchNorm on batch statistics, the drop block of APCNN.py:485-504 driven by python
    `random` with the same seed (exact_random_stream replays the reference's draws), both stages' logits, and the
    gradient of sum(out_mean * wt) at four depths of the network - i.e. through hk_roi_crop_resize_bwd,
    hk_att_pool_bwd and the ROI selection, as the reference's own autograd computes it.  The pinned values are the
    reference run in float64; the tolerances are multiples of the distance of the reference's OWN float32 run from
    them (`e32_*`: 1e-5 on the logits, 4e-3 on the first convolution's gradient - train-mode BatchNorm amplifies
    rounding; at batch 2 the reference's float32 gradients are 18 % away from its float64 ones)."""
    import random
    g = load('model_apcnn_train')
    n = g['out_mean'].shape[0]
    m = build('APCNN', num_classes=200)
    seeded_init(m, 910)
    m = m.to(DEV).train()
    m.exact_random_stream = True
    x = t(rs_randn(911, (n, 3, 224, 224))).to(DEV)
    wt = t(rs_randn(912, (n, 200))).to(DEV)
    random.seed(3)
    out_mean, out_list, _mask, rois = m(x, None)
    (out_mean * wt).sum().backward()
    for got, key in zip(rois, ('roi3', 'roi4', 'roi5')):       # same cells picked
        got = got.cpu().numpy()
        assert got.shape == g[key].shape
        np.testing.assert_array_equal(got[:, :5], g[key][:, :5])
    k = APCNN_TRAIN_K
    worst = 0.0
    for i, o in enumerate(out_list):
        r = float(rel(o, g['out_list'][i]))
        worst = max(worst, r / max(float(g['e32_out_list'][i]), 1e-5))
        assert r < k * max(float(g['e32_out_list'][i]), 1e-5), (i, r, float(g['e32_out_list'][i]))
    assert rel(out_mean, g['out_mean']) < k * max(float(g['e32_out_mean'][0]), 1e-5)
    grads = dict(m.named_parameters())
    for i, name in enumerate(g['grad_names']):
        gr = grads[str(name)].grad
        r = float(rel(sub(gr.cpu(), max(7, gr.numel() // 2000 | 1)), g['g%d' % i]))
        worst = max(worst, r / max(float(g['e32_g'][i]), 1e-5))
        assert r < k * max(float(g['e32_g'][i]), 1e-5), (str(name), r, float(g['e32_g'][i]))
        assert abs(float(gr.double().norm()) / float(g['gn%d' % i][0]) - 1) < k * max(float(g['e32_g'][i]), 1e-5), str(name)
    print(f'[apcnn 224 train] worst distance / reference fp32-vs-fp64 distance: {worst:.2f} (bound {k})')


def test_apcnn_at_config_shape_eval_vs_reference():
    """AP-CNN as BASELINE configs[4] runs it - 448 x 448 input, 8142 classes (hidden_num = 256, APCNN.py:360-363; the
    0.1 - 0.9 border band of get_att_roi, APCNN.py:451-455) - against the reference model in eval mode
    (tests/golden/model_apcnn_448.npz, gen_apcnn_448): the three ROI tables cell for cell, mask_cat, the 8 logits,
    out_mean and its argmax."""
    g = load('model_apcnn_448')
    m = build('APCNN', num_classes=8142)
    seeded_init(m, 940)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out_mean, out_list, mask_cat, rois = m(t(rs_randn(941, (2, 3, 448, 448))).to(DEV), None)
    for got, key in zip(rois, ('roi3', 'roi4', 'roi5')):
        got = got.cpu().numpy()
        assert got.shape == g[key].shape
        np.testing.assert_array_equal(got[:, :5], g[key][:, :5])
        np.testing.assert_allclose(got[:, 5], g[key][:, 5], rtol=1e-4)
    np.testing.assert_allclose(sub(mask_cat.cpu(), 13).numpy(), g['mask_cat'], rtol=1e-3, atol=1e-5)
    assert rel(torch.stack(out_list), g['out_list']) < 1e-4
    assert rel(out_mean, g['out_mean']) < 1e-4
    assert out_mean.argmax(1).cpu().tolist() == g['out_mean'].reshape(2, -1).argmax(1).tolist()


def _apcnn_448_train_distances():
    """One train-mode forward + backward of the AP-CNN plugin as test_apcnn_at_config_shape_train_vs_reference runs it
    (448 x 448, 8142 classes, batch 4, the reference's python-`random` drop sequence, seed 5) -> {tensor name: distance to the
    reference's float64 run / the reference's own float32-vs-float64 distance}; ROI cells asserted bit-exact on the way."""
    import random
    g = load('model_apcnn_448')
    n = g['t_out_mean'].shape[0]
    m = build('APCNN', num_classes=8142)
    seeded_init(m, 940)
    m = m.to(DEV).train()
    m.exact_random_stream = True
    x = t(rs_randn(942, (n, 3, 448, 448))).to(DEV)
    wt = t(rs_randn(943, (n, 8142))).to(DEV)
    random.seed(5)
    out_mean, out_list, _mask, rois = m(x, None)
    (out_mean * wt).sum().backward()
    for got, key in zip(rois, ('t_roi3', 't_roi4', 't_roi5')):
        got = got.cpu().numpy()
        assert got.shape == g[key].shape
        np.testing.assert_array_equal(got[:, :5], g[key][:, :5])
    dist = {}
    for i, o in enumerate(out_list):
        dist['out_list[%d]' % i] = float(rel(o, g['t_out_list'][i])) / max(float(g['t_e32_out_list'][i]), 1e-5)
    dist['out_mean'] = float(rel(out_mean, g['t_out_mean'])) / max(float(g['t_e32_out_mean'][0]), 1e-5)
    grads = dict(m.named_parameters())
    for i, name in enumerate(g['t_grad_names']):
        gr = grads[str(name)].grad
        dist['d ' + str(name)] = float(rel(sub(gr.cpu(), max(7, gr.numel() // 2000 | 1)), g['t_g%d' % i])) / max(float(g['t_e32_g'][i]), 1e-5)
        dist['|d ' + str(name) + '|'] = abs(float(gr.double().norm()) / float(g['t_gn%d' % i][0]) - 1) / max(float(g['t_e32_g'][i]), 1e-5)
    return dist


def test_apcnn_at_config_shape_train_vs_reference():
    """The same model in TRAIN mode at batch 4 with the reference's python-`random` drop sequence (seed 5,
    exact_random_stream): ROI cells, both stages' logits and gradients at five depths, pinned on the reference's float64
    run with the reference's own float32 distance as the yardstick (as test_apcnn_train_mode_matches_reference)."""
    dist = _apcnn_448_train_distances()
    k = APCNN_TRAIN_K_448
    worst = max(dist.values())
    for name, r in dist.items():
        assert r < k, (name, r)
    print(f'[apcnn 448 train] worst distance / reference fp32-vs-fp64 distance: {worst:.2f} (bound {k})')


def _torch_heads(monkeypatch):
    """Replace the hand-written kernels on AP-CNN's differentiable path by the reference's own torch formulas ON THE DEVICE
    (APCNN.py:256-266 + the GAP of cls*, :478-531 with the boxes the plugin hands over) - a checker for the test below, not
    a product path.  The ROI selection (no gradient, bit-exact against the reference cell for cell) stays on its kernel, so
    both runs crop the same boxes."""
    import hawkeye_amd.functional as HF

    def att_pool(f, a_s=None):
        return f.mean((2, 3)), (None if a_s is None else (a_s * f).mean((2, 3)))

    def att_pool_levels(feats, atts):
        return (torch.stack([f.mean((2, 3)) for f in feats]), torch.stack([(a * f).mean((2, 3)) for f, a in zip(feats, atts)]))

    def roi_crop_resize(x, box, drop, training):
        n, c, hh, ww = x.shape
        bx, dr = box.cpu(), drop.cpu()
        outs = []
        for i in range(n):
            x1, y1, x2, y2 = (int(v) for v in bx[i].long())                   # .long() truncation, :507
            if training:
                mask = torch.ones(c, hh, ww, dtype=x.dtype, device=x.device)
                if float(dr[i, 2]) >= 0:                                       # a level-3 / level-4 ROI was drawn (:494-504)
                    d = dr[i].long()
                    mask[:, int(d[1]):int(d[3]), int(d[0]):int(d[2])] = 0
                crop = (x[i] * mask)[:, y1:y2, x1:x2].contiguous().unsqueeze(0)
                rate = c * (bx[i, 3] - bx[i, 1]) * (bx[i, 2] - bx[i, 0]) / mask[:, y1:y2, x1:x2].sum().cpu()   # :509-511
                crop = crop * rate.to(x.device)
            else:
                crop = x[i, :, y1:y2, x1:x2].contiguous().unsqueeze(0)
            outs.append(torch.nn.functional.interpolate(crop, (hh, ww), mode='bilinear', align_corners=False))
        return torch.cat(outs, 0)

    for name, fn in (('att_pool', att_pool), ('att_pool_levels', att_pool_levels), ('roi_crop_resize', roi_crop_resize)):
        monkeypatch.setattr(HF, name, fn)


def test_apcnn_train_distance_is_the_trunks_not_the_heads(monkeypatch):
    """Where does the distance of test_apcnn_at_config_shape_train_vs_reference (5.4 x the reference's own fp32-vs-fp64
    distance at 448 x 448 / 8142 classes, against 1.1 for torch on the CPU) come from?  The same plugin, weights, input and
    drop sequence run twice on the MI355X: with the hk kernels, and with the attention pooling and the ROI crop / resize
    replaced by the reference's torch formulas on the device (everything else - MIOpen trunk, FPN, gates, classifier heads -
    identical).  If the two worst distances agree, the excess is the convolution library's summation order under train-mode
    BatchNorm, not the hand-written heads; both are printed per tensor."""
    d_hk = _apcnn_448_train_distances()
    _torch_heads(monkeypatch)
    d_th = _apcnn_448_train_distances()
    print('[apcnn 448 train, distance / reference fp32-vs-fp64 distance]  tensor: hk heads | torch-op heads on the device')
    for name in d_hk:
        if not name.startswith('|'):
            print(f'    {name:48s} {d_hk[name]:6.2f} | {d_th[name]:6.2f}')
    w_hk, w_th = max(d_hk.values()), max(d_th.values())
    print(f'[apcnn 448 train] worst: hk heads {w_hk:.2f}, torch-op heads {w_th:.2f} (ratio {w_hk / w_th:.2f})')
    # (MIOpen's weight gradients use atomics: the same configuration measured 5.2 and 7.3 in one session - so no tight ratio
    #  here.  What the comparison has to show is the REGIME: with torch's own ops in place of every hand-written kernel on the
    #  differentiable path the distance does not drop back towards torch-CPU's 1.1)
    assert w_hk < APCNN_TRAIN_K_448 and w_th < APCNN_TRAIN_K_448 and w_th > w_hk / 3.0, (w_hk, w_th)


def test_apcnn_exact_random_stream_mode():
    """exact_random_stream=True consumes python `random` exactly like the reference (APCNN.py:494-501):
    random() per image, then randint(0, n-1) on the level-3 / level-4 ROI count."""
    import random
    m = build('APCNN', num_classes=200)
    seeded_init(m, 910)
    m = m.to(DEV).train()
    m.exact_random_stream = True
    x = t(rs_randn(911, (2, 3, 224, 224))).to(DEV)
    random.seed(3)
    out = m(x, None)
    state_after = random.getstate()
    assert torch.isfinite(out[0]).all()
    (r3, n3), (r4, n4), _ = out[3].tables
    random.seed(3)
    for i in range(2):
        pr = random.random()
        if pr < 0.3:
            random.randint(0, int(n3[i]) - 1)
        elif pr < 0.6:
            random.randint(0, int(n4[i]) - 1)
    assert random.getstate() == state_after


def test_reducer_on_gpu_single_rank_matches_plain_sgd():
    from hawkeye_amd import ddp
    torch.manual_seed(0)
    m1 = build('BCNN', stage=1, num_classes=10).to(DEV)
    m2 = build('BCNN', stage=1, num_classes=10).to(DEV)
    m2.load_state_dict(m1.state_dict())
    red = ddp.GradientAllReducer(m1)
    assert [n for n, _ in red.describe()] == [2]            # stage 1: only classifier weight+bias train
    x = torch.randn(2, 3, 64, 64, device=DEV)
    y = torch.randint(0, 10, (2,), device=DEV)
    for m, use in ((m1, True), (m2, False)):
        if use:
            red.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), y)
        loss.backward()
        if use:
            red.finish()
    torch.testing.assert_close(m1.classifier.weight.grad, m2.classifier.weight.grad)


_RCCL_ONE_RANK = r"""
import json, os, sys
sys.path.insert(0, os.environ['HK_ROOT'])
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['HK_PORT'])
import torch
import torch.distributed as dist
from hawkeye_amd import ddp
rank, world, local = ddp.init_from_env('nccl', force=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1
torch.manual_seed(0)
net = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 8, 3, padding=1),
                                  torch.nn.Flatten(), torch.nn.Linear(8 * 16 * 16, 300000 // 2048 * 8)).cuda()
m1, m2 = net(), net()
m2.load_state_dict(m1.state_dict())
red = ddp.GradientAllReducer(m1, bucket_mb=0.25, trace=True)
x = torch.randn(4, 3, 16, 16, device='cuda')
red.zero_grad()
m1(x).square().mean().backward()
red.finish()
m2(x).square().mean().backward()
err = max(float((p.grad - q.grad).abs().max() / q.grad.abs().max()) for p, q in zip(m1.parameters(), m2.parameters()))
print(json.dumps({'max_abs_diff': err, 'buckets': red.describe(), 'timeline': red.timeline()}))
dist.destroy_process_group()
"""


def test_rccl_path_executes_on_one_gpu(tmp_path):
    """The bucketed all-reduce over RCCL (backend "nccl") with a single-rank process group: every bucket really goes
    through ncclAllReduce on RCCL's stream from the post-accumulate hooks and is joined before the step - the code path
    of an 8-GPU run, executed on the one GPU this box has.  Own process: the group is process-global state."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HK_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HK_PORT=str(port))
    p = subprocess.run([sys.executable, '-c', _RCCL_ONE_RANK], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['max_abs_diff'] < 1e-5                       # all-reduce over one rank = identity (MIOpen's weight-gradient
                                                            # kernels are not bitwise repeatable: two runs differ at 1e-8)
    assert len(out['buckets']) >= 2
    tl = out['timeline']
    issued = [b[2] for b in tl['buckets']]
    assert all(v is not None for v in issued) and issued == sorted(issued)       # last layer's bucket first
    assert issued[0] <= tl['backward_end_ms'] <= tl['joined_ms']


@pytest.mark.parametrize('world', [2, 8])
def test_bench_spawns_its_own_ranks(tmp_path, world):
    """`python bench.py --gpus N` (the driver's invocation shape, no torchrun around it) re-executes itself with one
    process per rank - 2, and 8 as on the scaling node; here all ranks share the box's single GPU over gloo.  Rank 0 prints
    the one JSON line, which carries the bucket timeline without being asked (N > 1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['HAWKEYE_BENCH_DETAIL'] = str(tmp_path / 'bench_detail.json')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--share-gpu', '--backend', 'gloo',
                        '--steps', '2', '--warmup', '1', '--batch', '2', '--image', '64'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['config']['global_batch'] == 2 * world and out['scaling'] == 'weak' and out['value'] > 0
    assert len(lines[0]) < 4096 and p.stdout.rstrip().splitlines()[-1] == lines[0]        # short, and the LAST stdout line
    d = out['ddp']                             # bucket 0 (the classifier: last layer, first gradient) goes out before backward ends
    assert d['buckets_mib'][0] > d['buckets_mib'][-1] or len(d['buckets_mib']) == 1
    assert d['issued_ms'][0] is not None and d['issued_ms'][0] <= d['backward_end_ms'] <= d['joined_ms']
    # who ran: every rank reported in, the communicator spans them, and (here: --share-gpu) they sat on ONE device -
    # on the scaling node the same keys must read distinct_devices == world, which bench.py asserts by itself
    assert d['ranks_seen'] == list(range(world)) and d['comm_size'] == world and d['backend'] == 'gloo'
    assert d['distinct_devices'] == 1 and len(d['devices']) == min(world, 8)
    assert 1 <= d['host_threads_per_rank'] <= max(1, (os.cpu_count() or 1) // world)


def test_vgg_trunk_with_fused_epilogues_equals_the_plain_stack():
    """ConvStack (model/backbone/vgg.py: nn.Sequential whose forward fuses bias + ReLU (+ 2 x 2 max-pool) behind each MIOpen
    convolution, csrc/trunk.hip) against the SAME children run one by one as nn.Sequential does (= what the reference's
    `features` executes, model/backbone/vgg.py:24-57) on a channels_last batch: the output, the input gradient and every
    parameter gradient.  Not bit-identity: the convolution is called without its bias here, and MIOpen may pick another
    algorithm for that problem; the epilogues themselves are bit-exact (test_trunk_epilogues_equal_the_ops_they_replace)."""
    import copy
    from hawkeye_amd.model.backbone import vgg16
    from hawkeye_amd.model.backbone.vgg import ConvStack
    torch.manual_seed(3)
    fused = vgg16(pretrained=False).features.to(DEV).to(memory_format=torch.channels_last)
    assert isinstance(fused, ConvStack) and isinstance(fused, torch.nn.Sequential)
    plain = torch.nn.Sequential(*copy.deepcopy(fused).children())
    assert list(plain.state_dict().keys()) == list(fused.state_dict().keys())
    x = torch.randn(4, 3, 96, 64, device=DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(4, 512, 3, 2, device=DEV)
    # MIOpen's backward is not repeatable on this stack: tools/probe/miopen_repeat_probe.py runs the PLAIN stack (torch ops and
    # MIOpen only) sixteen times on one input and finds gradients 1e-3 .. 1e-2 away from the other runs in 1 of 8 (4 x 3 x 96 x 64)
    # to 7 of 15 (8 x 3 x 448 x 448) of them - input gradient and first convolution mostly.  A wrong epilogue would be wrong every
    # time; so the comparison is repeated and has to hold in at least one of five attempts (it holds in ~7 of 8).
    tries = []
    for attempt in range(5):
        outs = []
        for net in (fused, plain):
            xi = x.clone().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            y = net(xi)
            (y * wt).sum().backward()
            outs.append((y.detach(), xi.grad, [p_.grad for p_ in net.parameters()]))
        (yf, gxf, gpf), (yp, gxp, gpp) = outs
        ey, ex = rel(yf, yp), rel(gxf, gxp)
        ep = max(rel(a, b) for a, b in zip(gpf, gpp))
        tries.append((ey, ex, ep))
        print(f'[VGG trunk, fused epilogues vs plain stack, attempt {attempt}] output {ey:.2e}  input gradient {ex:.2e}  worst parameter gradient {ep:.2e}')
        assert ey < 1e-5                                           # the forward is repeatable
        if ex < 1e-4 and ep < 1e-4:
            break
    else:
        raise AssertionError(f'gradients of the fused stack never matched the plain one: {tries}')
    # NCHW memory, or a child with a forward hook: the children run one by one (hooks fire)
    seen = []
    h = fused[1].register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    with torch.no_grad():
        y2 = fused(x)
    h.remove()
    assert seen == [(4, 64, 96, 64)] and rel(y2, yp) < 1e-5
    with torch.no_grad():
        assert rel(fused(x.contiguous()), yp) < 1e-5


def test_resnet_bottleneck_with_fused_residual_equals_the_plain_block():
    """Bottleneck (model/backbone/resnet.py) ends in `out += identity; relu(out)`: on the MI355X one pass (hk_add_relu_fwd); a
    forward hook on the block's ReLU switches the framework's two ops back in - same values, same gradients (the fused op is
    bit-identical to add_ + relu_: test_add_relu_equals_the_two_ops_it_replaces; BatchNorm and the convolutions run the same
    kernels either way)."""
    import copy
    from hawkeye_amd.model.backbone.resnet import Bottleneck
    torch.manual_seed(5)
    blk = Bottleneck(256, 64).to(DEV).to(memory_format=torch.channels_last).train()
    ref = copy.deepcopy(blk)
    h = ref.relu.register_forward_hook(lambda m, i, o: None)           # any hook: the plain path
    x = torch.randn(4, 256, 14, 14, device=DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(4, 256, 14, 14, device=DEV)
    tries = []
    for attempt in range(5):           # (repeated for the reason given in test_vgg_trunk_with_fused_epilogues_equals_the_plain_stack)
        outs = []
        for net in (blk, ref):
            xi = x.clone().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            y = net(xi)
            (y * wt).sum().backward()
            outs.append((y.detach(), xi.grad, [p_.grad for p_ in net.parameters()]))
        (y1, g1, p1), (y2, g2, p2) = outs
        tries.append((rel(y1, y2), rel(g1, g2), max(rel(a, b) for a, b in zip(p1, p2))))
        assert tries[-1][0] < 1e-6
        if tries[-1][1] < 1e-5 and tries[-1][2] < 1e-5:
            break
    else:
        raise AssertionError(f'gradients of the fused block never matched the plain one: {tries}')
    h.remove()


def test_bench_line_is_last_on_stdout_with_rccl(tmp_path):
    """The driver parses bench.py's LAST stdout line.  With the `nccl` (RCCL) backend the library leaves its version banner
    in the C stdio buffer of stdout until the process exits - behind everything python printed (observed on the MI355X box
    with stdout on a pipe, which is how the driver reads it).  One rank over RCCL (--force-pg), stdout on a pipe: the JSON
    line must still be the last one, and it carries what RCCL logged about the communicator (`ddp.rccl`)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'NCCL_DEBUG_FILE')}
    env['HAWKEYE_BENCH_DETAIL'] = str(tmp_path / 'bench_detail.json')
    env['MASTER_PORT'] = '29517'
    for rccl_log in ('1', '0'):                # with the log armed (banner in the file) and without (banner on stdout at exit)
        env['HAWKEYE_RCCL_LOG'] = rccl_log
        p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--force-pg', '--steps', '2', '--warmup', '1', '--batch', '2',
                            '--image', '64', '--no-cpu-baseline', '--no-kernels', '--no-other-models'],
                           capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        last = p.stdout.rstrip().splitlines()[-1]
        assert last.startswith('{'), p.stdout[-600:]
        out = json.loads(last)
        assert out['n_gpus'] == 1 and out['value'] > 0 and 'issued_ms' in out['ddp']
        if rccl_log == '1':
            assert 'RCCL' in (out['ddp']['rccl']['version'] or ''), out['ddp']
            print('[bench --force-pg] ddp.rccl =', out['ddp']['rccl'])


def test_trainer_runs_one_synthetic_epoch(tmp_path):
    """The reference's Trainer flow (build from yaml -> train -> validate -> checkpoint) on the MI355X heads."""
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.examples.BCNN import BCNNTrainer
    cfg = CfgNode.load_cfg(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), 'configs',
                                             'BCNN_S2_synthetic.yaml')))
    cfg.experiment.log_dir = str(tmp_path)
    cfg.dataset.samples = 8
    cfg.dataset.batch_size = 4
    cfg.dataset.num_workers = 0
    cfg.dataset.transformer.image_size = 64
    cfg.train.save_frequence = 1
    cfg.train.epoch = 2
    cfg.freeze()
    tr = BCNNTrainer(cfg)
    tr.train()
    assert len(tr.performance_meters['train']['loss'].values) == 2
    # the reference never writes a periodic checkpoint after the FIRST epoch (train.py:296: `epoch != 0 and ...`)
    assert not os.path.isfile(os.path.join(tr.log_root, 'BCNN_epoch_1.pth'))
    assert os.path.isfile(os.path.join(tr.log_root, 'BCNN_epoch_2.pth'))
    sd = torch.load(os.path.join(tr.log_root, 'BCNN_epoch_2.pth'), map_location='cpu')
    assert 'classifier.weight' in sd and not any(k.startswith('module.') for k in sd)
