"""Host-side contracts of the plugin surface, checked on CPU: registry semantics,
yacs-compatible config, constructor contracts, child order, and byte-compatible
state_dict keys/shapes vs the REFERENCE's models (tests/golden/state_dict_keys.json,
written by oracle/gen_golden.py from /root/reference)."""
import copy
import json
import os

import pytest
import torch

from hawkeye_amd.config import CfgNode
from hawkeye_amd.model.registry import MODEL, install_into
from hawkeye_amd.utils.repository import Repository
import hawkeye_amd.model  # noqa: F401

KEYS = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'state_dict_keys.json')))

CONFIGS = {
    'BCNN': dict(name='BCNN', stage=2, num_classes=200),
    'CBCNN': dict(name='CBCNN', stage=2, num_classes=200, input_channel=512, output_channel=6000),
    'MPN': dict(name='MPN', iter_num=5, is_sqrt=True, is_vec=True, input_dim=2048, dimension_reduction=256, num_classes=200),
    'APCNN': dict(name='APCNN', num_classes=200),
    'APCNN_8142': dict(name='APCNN', num_classes=8142),
    'OSMENet': dict(name='OSMENet', num_attention=2, num_classes=200),
    'CIN': dict(name='CIN', num_classes=200),
}


def test_registry_semantics():
    r = Repository()

    @r.register
    def foo():
        return 1

    assert r.get('foo') is foo and r['foo']() == 1
    with pytest.raises(AssertionError):
        r.register(foo)                      # uniqueness (utils/repository.py:11)
    assert sorted(MODEL) == ['APCNN', 'BCNN', 'CBCNN', 'CIN', 'MPN', 'OSMENet', 'ResNet101', 'ResNet50']
    ref = Repository()
    ref.register(foo)
    ref['BCNN'] = object()
    install_into(ref)
    assert ref['BCNN'] is MODEL['BCNN'] and 'foo' in ref


def test_cfgnode_yacs_compat(tmp_path):
    text = 'experiment:\n  name: x\n  cuda: [0]\nmodel:\n  name: BCNN\n  num_classes: 200\n'
    cfg = CfgNode.load_cfg(text)
    assert cfg.model.name == 'BCNN' and cfg['model']['num_classes'] == 200
    assert 'stage' not in cfg.model and 'name' in cfg.model
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.model.name = 'y'
    assert CfgNode.load_cfg(str(cfg)).to_dict() == cfg.to_dict()
    p = tmp_path / 'c.yaml'
    p.write_text(text)
    from hawkeye_amd.config import setup_config
    assert setup_config(['--config', str(p)]).experiment.cuda == [0]
    assert copy.deepcopy(cfg).is_frozen()


@pytest.mark.parametrize('name', list(CONFIGS))
def test_state_dict_keys_match_reference(name):
    cfg = CfgNode(CONFIGS[name])
    cfg.freeze()
    model = MODEL.get(cfg.name)(cfg)
    ref = KEYS[name]
    got = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert got == ref['state_dict']
    assert [n for n, _ in model.named_children()] == ref['children']
    assert sum(p.numel() for p in model.parameters()) == ref['n_params']


def test_bcnn_contract_details():
    m1 = MODEL.get('BCNN')(CfgNode(dict(stage=1, num_classes=7)))
    assert m1.stage == 1 and all(not p.requires_grad for p in m1.backbone.parameters())
    assert all(p.requires_grad for p in m1.classifier.parameters())
    m2 = MODEL.get('BCNN')(CfgNode(dict(num_classes=7)))          # stage optional -> 2 (BCNN.py:36)
    assert m2.stage == 2 and all(p.requires_grad for p in m2.parameters())
    m3 = copy.deepcopy(m2)                                        # PeerLearningNet deep-copies the base model
    assert m3.classifier.weight.shape == (7, 512 * 512)
    assert len(list(m2.bilinear_pooling.parameters())) == 0


def test_cbcnn_has_no_sketch_state_and_is_copyable():
    m = MODEL.get('CBCNN')(CfgNode(CONFIGS['CBCNN']))
    assert len(list(m.bilinear_pooling.named_buffers())) == 0 and len(list(m.bilinear_pooling.parameters())) == 0
    assert m.bilinear_pooling.rand_h_1[:4].tolist() == [5157, 235, 3980, 5192]
    c = copy.deepcopy(m)
    assert (c.bilinear_pooling.rand_h_2 == m.bilinear_pooling.rand_h_2).all()


def test_apcnn_optimizer_split_contract():
    """Examples/APCNN.py:38-42 splits children()[:7] / [7:]."""
    m = MODEL.get('APCNN')(CfgNode(CONFIGS['APCNN']))
    kids = [n for n, _ in m.named_children()]
    assert kids[:7] == ['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3']
    assert kids[7:] == ['layer4', 'fpn', 'apn', 'cls5', 'cls4', 'cls3', 'cls_concate']
    assert m.cls3[3].out_features == 512
    assert MODEL.get('APCNN')(CfgNode(CONFIGS['APCNN_8142'])).cls3[3].out_features == 256


def test_forward_on_cpu_fails_loudly():
    from hawkeye_amd._lib import HawkeyeHipError
    m = MODEL.get('BCNN')(CfgNode(dict(num_classes=3)))
    with pytest.raises(HawkeyeHipError):
        m(torch.randn(1, 3, 64, 64))


@pytest.mark.skipif(not os.path.isdir('/root/reference/configs'), reason='reference checkout not present')
def test_reference_yaml_configs_load_unchanged():
    """Every yaml of the reference loads through the yacs-compatible CfgNode (config.py:13-17 semantics)."""
    import glob
    from hawkeye_amd.config import load_config
    files = sorted(glob.glob('/root/reference/configs/*.yaml'))
    assert len(files) >= 20
    for f in files:
        cfg = load_config(f)
        assert cfg.is_frozen() and 'name' in cfg.model and 'experiment' in cfg
    cfg = load_config('/root/reference/configs/BCNN_S2.yaml')
    assert cfg.model.stage == 2 and cfg.dataset.transformer.image_size == 448 and cfg.experiment.cuda == [0]


@pytest.mark.parametrize('name', ['Baseline', 'BCNN_S1', 'BCNN_S2', 'CBCNN_S2', 'MPN', 'APCNN'])
def test_in_repo_yaml_configs(name, monkeypatch):
    """The BASELINE configs (and the default configs/Baseline.yaml that setup_config falls back to) ship with the repo:
    the GPU box has no /root/reference.  Each loads, names a registered plugin that constructs from its model node,
    and - where the reference checkout exists - parses to exactly the reference's tree."""
    from hawkeye_amd.config import load_config, setup_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, 'configs', name + '.yaml'))
    assert cfg.is_frozen()
    model = MODEL.get(cfg.model.name)(cfg.model)
    assert sum(p.numel() for p in model.parameters()) > 1e6
    ref = f'/root/reference/configs/{name}.yaml'
    if os.path.exists(ref):
        assert cfg.to_dict() == load_config(ref).to_dict()
    if name == 'Baseline':
        monkeypatch.chdir(root)
        assert setup_config([]).to_dict() == cfg.to_dict()


def test_example_trainers_keep_the_reference_optimizer_groups():
    """Optimiser param-group logic of the per-method trainers, without running them (no GPU here)."""
    import hawkeye_amd.examples.APCNN as EA
    import hawkeye_amd.examples.MPN as EM

    class Shell:           # minimal stand-in for a constructed Trainer
        pass

    m = MODEL.get('APCNN')(CfgNode(CONFIGS['APCNN']))
    sh = Shell()
    sh.model = m
    sh.get_model_module = lambda model=None: m
    opt = EA.APCNNTrainer.get_optimizer(sh, CfgNode(dict(lr=0.0005, weight_decay=0.0005)))
    n_head = sum(p.numel() for k in list(m.children())[7:] for p in k.parameters())
    assert sum(p.numel() for p in opt.param_groups[0]['params']) == n_head
    assert opt.param_groups[1]['lr'] == pytest.approx(0.00005)
    m = MODEL.get('MPN')(CfgNode(CONFIGS['MPN']))
    sh.model = m
    opt = EM.MPNTrainer.get_optimizer(sh, CfgNode(dict(lr=8e-5, weight_decay=2e-5)))
    assert [g['lr'] for g in opt.param_groups] == [8e-5, 8e-5, pytest.approx(1.6e-5)]


def test_vgg_conv_stack_is_a_sequential_with_the_reference_keys():
    """ConvStack (the VGG trunk whose forward fuses the epilogues behind each convolution on an MI355X) is an nn.Sequential
    with the reference's children under the reference's indices (model/backbone/vgg.py:24-57: keys `{0,2,5,...,28}.{weight,bias}`),
    and off the GPU - CPU tensors here - it runs them one by one: the same values as a plain nn.Sequential of the same children."""
    from hawkeye_amd.model.backbone import vgg16
    from hawkeye_amd.model.backbone.vgg import ConvStack
    torch.manual_seed(0)
    feats = vgg16(pretrained=False).features
    assert isinstance(feats, ConvStack) and isinstance(feats, torch.nn.Sequential) and len(feats) == 31
    convs = [i for i, m in enumerate(feats) if isinstance(m, torch.nn.Conv2d)]
    assert convs == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
    assert list(feats.state_dict().keys()) == [f'{i}.{p}' for i in convs for p in ('weight', 'bias')]
    plain = torch.nn.Sequential(*feats.children())
    x = torch.randn(1, 3, 32, 32)
    for inp in (x, x.contiguous(memory_format=torch.channels_last)):
        assert torch.equal(feats(inp), plain(inp))
    m = MODEL.get('BCNN')(CfgNode(CONFIGS['BCNN']))
    assert isinstance(m.backbone, ConvStack) and isinstance(m.backbone[:7], torch.nn.Sequential)     # slicing keeps working
