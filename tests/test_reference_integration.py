"""Route A of INTEGRATION.md, executed against the REFERENCE'S OWN code: `install_into(MODEL)` swaps the plugins into
the reference's registry and the reference's `Trainer` (train.py) / `Tester` (test.py) - their `__init__`, `get_model`
(train.py:158-169, test.py:65-76), `to_device`, `train()` loop, `batch_training`, `validate`, `save_model` - drive the
MI355X heads unchanged.  Only what the build container lacks is replaced: `torchvision` / `tensorboardX` / `yacs` are
not installed (stub modules), there is no image data (the dataset builders are overridden with a synthetic set), and
there is no GPU here (the heads run on the CPU emulation of the kernel sources; `experiment.cuda: []` makes the
reference take its CPU branch).  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get('HAWKEYE_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DRIVER = r'''
import os, sys, types, json
REF, ROOT, WORK, NAME = sys.argv[1:5]
sys.argv = [os.path.abspath(__file__), '--config', os.path.join(WORK, 'cfg.yaml')]
sys.path[:0] = [REF, ROOT, os.path.join(ROOT, 'tests')]

# ---- packages the container does not have: just enough surface for the reference's import statements
def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m
class InterpolationMode:
    BILINEAR = 'bilinear'
class _Unavailable:
    def __init__(self, *a, **k):
        raise RuntimeError('torchvision is not installed: the test supplies tensors, not images')
tvf = mod('torchvision.transforms.functional', InterpolationMode=InterpolationMode)
tva = mod('torchvision.transforms.autoaugment')
tvtt = mod('torchvision.transforms.transforms')
tvt = mod('torchvision.transforms', functional=tvf, autoaugment=tva, transforms=tvtt, Compose=_Unavailable,
          Resize=_Unavailable, CenterCrop=_Unavailable, ToTensor=_Unavailable, Normalize=_Unavailable)
mod('torchvision', transforms=tvt)
class SummaryWriter:
    def __init__(self, *a, **k): self.scalars = []
    def add_scalar(self, tag, value, global_step=None): self.scalars.append((tag, float(value), global_step))
    def close(self): pass
mod('tensorboardX', SummaryWriter=SummaryWriter)
import hawkeye_amd.config as _hc                       # yacs-compatible CfgNode (load_cfg / freeze / attribute access)
mod('yacs')
mod('yacs.config', CfgNode=_hc.CfgNode)

import torch
torch.set_num_threads(4)
import utils.utils as ref_utils                       # reference: TqdmHandler.emit sleeps 1 s per record (utils.py:76)
class _NoSleep:
    def __getattr__(self, k):
        import time
        return (lambda s: None) if k == 'sleep' else getattr(time, k)
ref_utils.time = _NoSleep()

import model                                           # the reference package: registers its own methods
from model.registry import MODEL
reference_cls = MODEL.get(NAME)
import hawkeye_amd.model                               # INTEGRATION.md, route A
from hawkeye_amd.model.registry import install_into
install_into(MODEL)
assert MODEL.get(NAME) is not reference_cls and MODEL.get(NAME).__module__.startswith('hawkeye_amd.')

from train import Trainer                              # the reference's trainer / tester, as they are
from test import Tester
from emu.harness import emulated

class Synthetic(torch.utils.data.Dataset):
    def __init__(self, n, size, classes, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, 3, size, size, generator=g)
        self.y = torch.randint(0, classes, (n,), generator=g)
    def __len__(self): return len(self.y)
    def __getitem__(self, i): return {'img': self.x[i], 'label': self.y[i]}

class SynTrainer(Trainer):                             # data side only; everything else is inherited
    def get_transformers(self, config): return {'train': None, 'val': None}
    def get_dataset(self, config):
        return {s: Synthetic(8, config.transformer.image_size, self.config.model.num_classes, i) for i, s in enumerate(('train', 'val'))}

class SynTester(Tester):
    def get_transformer(self, config): return None
    def get_dataset(self, config): return Synthetic(8, config.transformer.image_size, self.config.model.num_classes, 1)

with emulated():
    tr = SynTrainer()
    assert type(tr.model).__module__.startswith('hawkeye_amd.'), type(tr.model)
    before = [p.detach().clone() for p in tr.model.parameters()]
    tr.train()
    moved = sum(float((p.detach() - q).abs().sum()) for p, q in zip(tr.model.parameters(), before))
    ckpt = os.path.join(tr.log_root, f'{NAME}_epoch_2.pth')
    saved = os.path.isfile(ckpt)
    # the reference's Tester on the checkpoint the reference's Trainer wrote
    import yaml
    cfg = yaml.safe_load(open(sys.argv[2]))
    cfg['model']['load'] = ckpt
    yaml.safe_dump(cfg, open(sys.argv[2], 'w'))
    te = SynTester()
    te.test()
    same = all(torch.equal(a, b) for a, b in zip(tr.model.state_dict().values(), te.model.state_dict().values()))
    # same weights, same validation set, both in eval mode: the two drivers must report the same accuracy
    print(json.dumps({'moved': moved, 'saved': saved, 'train_loss': tr.performance_meters['train']['loss'].current_value,
                      'val_acc': tr.performance_meters['val']['acc'].current_value, 'test_acc': te.performance_meters['acc'].current_value,
                      'test_count': te.average_meters['acc'].count, 'same_weights': same,
                      'scalars': [s[0] for s in tr.tb_writer.scalars]}))
'''

_CFG = '''
experiment:
  name: ref_integration
  log_dir: {work}/logs
  cuda: []
  seed: 0
dataset:
  batch_size: 4
  num_workers: 0
  transformer:
    image_size: {size}
    resize_size: {size}
model:
{model}
train:
  epoch: 2
  save_frequence: 2
  optimizer:
    lr: 0.001
    weight_decay: 0.0
  scheduler:
    T_max: 2
    eta_min: 0.0
  criterion:
    name: CrossEntropyLoss
'''

_MODELS = {
    'BCNN': (64, '  name: BCNN\n  stage: 2\n  num_classes: 3\n'),
    'CBCNN': (64, '  name: CBCNN\n  stage: 2\n  num_classes: 3\n  input_channel: 512\n  output_channel: 256\n'),
    'MPN': (64, '  name: MPN\n  iter_num: 5\n  is_sqrt: True\n  is_vec: True\n  input_dim: 2048\n  dimension_reduction: 64\n'
                '  num_classes: 3\n'),
}


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')
@pytest.mark.parametrize('name', ['BCNN', 'CBCNN', 'MPN'])
def test_reference_trainer_and_tester_drive_the_plugins(name, tmp_path):
    import json

    from emu import build_emu
    if build_emu._compiler() is None:
        pytest.skip('no clang++ to build the emulated kernels')
    build_emu.build()
    size, model_yaml = _MODELS[name]
    (tmp_path / 'cfg.yaml').write_text(_CFG.format(work=tmp_path, size=size, model=model_yaml))
    driver = tmp_path / 'driver.py'
    driver.write_text(_DRIVER)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, str(driver), REF, ROOT, str(tmp_path), name], capture_output=True, text=True,
                       timeout=900, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['saved'] and out['moved'] > 0 and out['same_weights']
    assert out['train_loss'] == out['train_loss'] and out['train_loss'] > 0           # finite
    assert out['test_count'] == 8 and abs(out['test_acc'] - out['val_acc']) < 1e-9
    assert 'train/loss' in out['scalars'] and 'val/acc' in out['scalars']
