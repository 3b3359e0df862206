"""CPU tier: the whole-model cases of tests/test_gpu_models.py (reference logits goldens, train steps) with the backbone
on torch-CPU and every head running the emulated kernel sources (tests/emu).  Test infrastructure only."""
import importlib.util
import os

import pytest

from emu.harness import emulated

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location('_emu_cases_test_gpu_models', os.path.join(_here, 'test_gpu_models.py'))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_mod.DEV = 'cpu'

# device-runtime specific (RCCL reducer on a GPU stream, Trainer device placement)
_SKIP = {'test_reducer_on_gpu_single_rank_matches_plain_sgd', 'test_trainer_runs_one_synthetic_epoch',
         'test_rccl_path_executes_on_one_gpu', 'test_bench_spawns_its_own_ranks',
         # round 6: RCCL's stdout banner, MIOpen convolutions under the fused trunk epilogues, a same-device A/B of the AP-CNN heads
         'test_bench_line_is_last_on_stdout_with_rccl', 'test_vgg_trunk_with_fused_epilogues_equals_the_plain_stack',
         'test_apcnn_train_distance_is_the_trunks_not_the_heads', 'test_resnet_bottleneck_with_fused_residual_equals_the_plain_block'}
for _name, _obj in list(vars(_mod).items()):
    if _name.startswith('test_') and callable(_obj) and _name not in _SKIP:
        globals()[_name] = _obj


# 10 - 25 s each when emulated (HK_EMU_FULL=1 runs them)
_HEAVY = {'test_models_at_config_input_size_vs_reference[MPN]', 'test_models_at_config_input_size_vs_reference[BCNN-channels_last]',
          'test_models_at_config_input_size_vs_reference[CBCNN-channels_last]', 'test_train_step_runs_and_updates[MPN-128]', 'test_train_step_runs_and_updates[OSMENet-224]',
          'test_osmenet_eval_matches_reference', 'test_apcnn_at_config_shape_train_vs_reference',
          'test_apcnn_at_config_shape_eval_vs_reference', 'test_bcnn_signed_sqrt_whole_model'}


@pytest.fixture(autouse=True)
def _skip_heavy(request):
    if request.node.name in _HEAVY and os.environ.get('HK_EMU_FULL') != '1':
        pytest.skip('slow under emulation (HK_EMU_FULL=1 runs it)')


@pytest.fixture(autouse=True, scope='module')
def _emulated_heads():
    from emu import build_emu
    if build_emu._compiler() is None:
        pytest.skip('no clang++ to build the emulated kernels')
    with emulated():
        yield


@pytest.mark.parametrize('which', ['BCNN', 'OSMENet', 'CIN'])
def test_trainers_end_to_end_on_emulated_heads(which, tmp_path, monkeypatch):
    """Trainer flow from a yaml (build -> train -> validate -> checkpoint) with the device hook pointed at the CPU:
    BCNN (cross entropy) and OSMENet (class-balanced batches + MAMC criterion on hk_npairs_loss)."""
    import torch

    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.train import Trainer
    if which != 'BCNN' and os.environ.get('HK_EMU_FULL') != '1':
        pytest.skip('ResNet at 224x224 on the CPU: HK_EMU_FULL=1 runs it')
    monkeypatch.setattr(Trainer, 'select_device', lambda self, cfg: torch.device('cpu'))
    root = os.path.dirname(_here)
    if which == 'BCNN':
        from hawkeye_amd.examples.BCNN import BCNNTrainer as T
        cfg = CfgNode.load_cfg(open(os.path.join(root, 'configs', 'BCNN_S2_synthetic.yaml')))
        cfg.dataset.samples, cfg.dataset.batch_size, cfg.dataset.transformer.image_size = 8, 4, 64
    elif which == 'OSMENet':
        from hawkeye_amd.examples.OSMENet import OSMENetTrainer as T
        cfg = CfgNode.load_cfg(open(os.path.join(root, 'configs', 'OSMENet_synthetic.yaml')))
        cfg.dataset.samples, cfg.dataset.batch_size = 12, 4
        cfg.dataset.n_classes, cfg.dataset.n_samples, cfg.model.num_classes = 2, 2, 3
    else:
        from hawkeye_amd.examples.CIN import CINTrainer as T
        cfg = CfgNode.load_cfg(open(os.path.join(root, 'configs', 'CIN_synthetic.yaml')))
        cfg.dataset.samples, cfg.dataset.batch_size = 12, 4
        cfg.dataset.n_classes, cfg.dataset.n_samples, cfg.model.num_classes = 2, 2, 3
        cfg.train.criterion.r_channel = 8
    cfg.experiment.log_dir = str(tmp_path)
    cfg.dataset.num_workers = 0
    cfg.train.save_frequence = 1
    cfg.freeze()
    tr = T(cfg)
    tr.train()
    assert len(tr.performance_meters['train']['loss'].values) == 1
    # (no periodic checkpoint after the first epoch: `epoch != 0 and ...`, reference train.py:296)
    assert not os.path.isfile(os.path.join(tr.log_root, f'{which}_epoch_1.pth'))


def test_tester_evaluates_a_trainer_checkpoint(tmp_path, monkeypatch):
    """Trainer -> checkpoint -> Tester (reference test.py flow: strict load of `model.load`, validation pass, top-1)
    and a real image folder through the presets, with the input finalised on the device (uint8 from the workers)."""
    import numpy as np
    import torch
    from PIL import Image

    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.examples.BCNN import BCNNTrainer
    from hawkeye_amd.test import Tester
    from hawkeye_amd.train import Trainer
    monkeypatch.setattr(Trainer, 'select_device', lambda self, cfg: torch.device('cpu'))
    monkeypatch.setattr(Tester, 'select_device', lambda self, cfg: torch.device('cpu'))
    root = os.path.dirname(_here)
    # a tiny image folder in the reference's `label relpath` format
    img_root, meta = tmp_path / 'images', tmp_path / 'meta'
    (img_root / 'a').mkdir(parents=True)
    meta.mkdir()
    rs = np.random.RandomState(0)
    lines = []
    for i in range(8):
        Image.fromarray((rs.rand(90, 120, 3) * 255).astype(np.uint8)).save(img_root / 'a' / f'{i}.jpg')
        lines.append(f'{i % 3} a/{i}.jpg')
    for split in ('train', 'val'):
        (meta / f'{split}.txt').write_text('\n'.join(lines) + '\n')
    cfg = CfgNode.load_cfg(open(os.path.join(root, 'configs', 'BCNN_S2_synthetic.yaml')))
    cfg.experiment.log_dir = str(tmp_path / 'logs')
    cfg.dataset.name, cfg.dataset.root_dir, cfg.dataset.meta_dir = 'folder', str(img_root), str(meta)
    cfg.dataset.batch_size, cfg.dataset.num_workers = 4, 0
    cfg.dataset.transformer.image_size, cfg.dataset.transformer.resize_size = 64, 72
    cfg.dataset.transformer.device_finalize = True
    cfg.model.num_classes = 3
    cfg.train.save_frequence = 1
    cfg.train.epoch = 2
    cfg.freeze()
    tr = BCNNTrainer(cfg)
    tr.train()
    ckpt = os.path.join(tr.log_root, 'BCNN_epoch_2.pth')
    assert os.path.isfile(ckpt)
    tcfg = cfg.clone() if hasattr(cfg, 'clone') else CfgNode(cfg.to_dict())
    tcfg.defrost() if hasattr(tcfg, 'defrost') else None
    tcfg.model.load = ckpt
    te = Tester(tcfg)
    te.test()
    assert te.average_meters['acc'].count == 8 and 0.0 <= te.average_meters['acc'].avg <= 100.0
    # same weights in both
    for (k1, v1), (k2, v2) in zip(tr.model.state_dict().items(), te.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def _two_rank_trainer_worker(rank, world, port, log_dir, out_dir):
    import torch

    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.examples.BCNN import BCNNTrainer
    from hawkeye_amd.train import Trainer
    Trainer.select_device = lambda self, cfg: torch.device('cpu')
    root = os.path.dirname(_here)
    with emulated():
        cfg = CfgNode.load_cfg(open(os.path.join(root, 'configs', 'BCNN_S2_synthetic.yaml')))
        cfg.experiment.log_dir = log_dir
        cfg.experiment.debug = False                     # the start-up path with the "folder must not exist" check
        cfg.dataset.samples, cfg.dataset.batch_size, cfg.dataset.num_workers = 12, 3, 0
        cfg.dataset.transformer.image_size = 32
        cfg.model.num_classes = 3
        cfg.train.epoch, cfg.train.save_frequence = 1, 1
        cfg.freeze()
        tr = BCNNTrainer(cfg)
        tr.train()
        # what this rank alone saw in validation, recomputed without the cross-rank sync
        local = torch.zeros(2, dtype=torch.float64)
        tr.model.eval()
        with torch.no_grad():
            for data in tr.dataloaders['val']:
                out = tr.model(data['img'])
                local += torch.tensor([float((out.argmax(1) == data['label']).sum()), float(len(data['label']))])
        torch.save({'val_acc': tr.performance_meters['val']['acc'].current_value,
                    'train_loss': tr.performance_meters['train']['loss'].current_value,
                    'val_count': tr.average_meters['acc'].count, 'local': local,
                    'lr': tr.optimizer.param_groups[0]['lr'],
                    'w': tr.model.classifier.weight.detach().clone()}, os.path.join(out_dir, f'r{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_multi_rank_trainer_metrics_are_global(tmp_path, world):
    """Two - and eight - Trainer processes over gloo (heads on the emulated kernels), `debug: False`: rank 0 alone creates the
    experiment folder (the other rank waits), validation accuracy / train loss are the averages over BOTH shards on
    every rank (what the reference's single-process DataParallel run reports, and what ReduceLROnPlateau and the
    best-model rule consume), and the replicas end the epoch with identical weights."""
    import socket

    import torch
    import torch.multiprocessing as mp
    from emu import build_emu
    if build_emu._compiler() is None:
        pytest.skip('no clang++ to build the emulated kernels')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / 'out'
    out.mkdir()
    mp.spawn(_two_rank_trainer_worker, args=(world, port, str(tmp_path / 'logs'), str(out)), nprocs=world, join=True)
    rs = [torch.load(out / f'r{r}.pt') for r in range(world)]
    r0 = rs[0]
    for r in rs[1:]:
        assert r0['val_acc'] == r['val_acc'] and r0['train_loss'] == r['train_loss'] and r0['lr'] == r['lr']
        assert r0['val_count'] == r['val_count'] == 12                     # the whole validation set, not one shard
        assert torch.equal(r0['w'], r['w'])
    # 12 validation samples over 8 ranks do not divide: shards of 2 and 1, every sample counted ONCE (a padding sampler
    # would count 16 and move the accuracy that drives ReduceLROnPlateau / best_model.pth)
    both = sum(r['local'] for r in rs)
    assert both[1] == 12 and abs(r0['val_acc'] - 100.0 * float(both[0] / both[1])) < 1e-9
    assert os.path.isfile(tmp_path / 'logs' / 'bcnn_s2_synthetic' / 'train_config.yaml')
