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
    b = torch.from_numpy(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / b.norm())


def build(name, **kw):
    import hawkeye_amd.model  # noqa: F401
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.model.registry import MODEL
    return MODEL.get(name)(CfgNode(dict(name=name, **kw)))


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
    cfg.freeze()
    tr = BCNNTrainer(cfg)
    tr.train()
    assert len(tr.performance_meters['train']['loss'].values) == 1
    assert os.path.isfile(os.path.join(tr.log_root, 'BCNN_epoch_1.pth'))
    sd = torch.load(os.path.join(tr.log_root, 'BCNN_epoch_1.pth'), map_location='cpu')
    assert 'classifier.weight' in sd and not any(k.startswith('module.') for k in sd)
