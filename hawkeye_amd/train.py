"""Trainer runtime with the reference's override points (train.py:37-435):
get_model / get_transformers / get_dataset / get_dataloader / get_criterion /
get_optimizer / get_scheduler / batch_training / batch_validate /
do_scheduler_step / on_start_epoch / on_end_epoch, `@emergency_save`, checkpoint
layout `{epoch, model, optimizer, scheduler}`, `best_model.pth`.

What changed underneath is only the device / parallel layer: instead of
`to_device()` wrapping the model in single-process `nn.DataParallel`
(train.py:220-228) the trainer runs as ONE PROCESS PER GPU (torchrun) and
synchronises gradients with `hawkeye_amd.ddp.GradientAllReducer` (RCCL).  The
yaml's `dataset.batch_size` is the PER-GPU batch.

`dataset.name: synthetic` gives an on-device random dataset (no disk, no
torchvision) - that is what bench.py and the smoke runs use; image-folder data
goes through hawkeye_amd.data (PIL + torch ops, no torchvision).
"""
import logging
import os
import sys
import traceback
from shutil import copyfile

import torch

from . import ddp
from .config import setup_config
from .model.registry import MODEL
from .utils import (AverageMeter, PerformanceMeter, ScalarWriter, Timer, TqdmHandler, accuracy,
                    set_random_seed)


def emergency_save(func):
    """Save a checkpoint when training is interrupted or raises (train.py:17-34)."""

    def wrapped(self):
        try:
            func(self)
        except KeyboardInterrupt:
            self.logger.info('KeyboardInterrupt - try to save checkpoint ...')
            self.save_checkpoint()
        except Exception as e:  # noqa: BLE001
            self.logger.error(repr(e))
            self.logger.error(traceback.format_exc())
            self.logger.info('try to save checkpoint ...')
            self.save_checkpoint()

    return wrapped


class Trainer:
    def __init__(self, config=None):
        self.config = config if config is not None else setup_config()
        cfg = self.config
        self.rank, self.world, self.local_rank = ddp.init_from_env()
        self.is_main = self.rank == 0

        self.epoch = self.start_epoch = 0
        self.total_epoch = cfg.train.epoch
        self.resume = 'resume' in cfg.experiment and cfg.experiment.resume
        self.debug = cfg.experiment.debug if 'debug' in cfg.experiment else False
        self.log_root = os.path.join(cfg.experiment.log_dir, cfg.experiment.name)
        self.report_one_line = True

        # only rank 0 checks and creates the experiment folder; the others wait for it (a rank that created the folder
        # first would make rank 0's "already exists" check fail and leave everybody else hanging in the first broadcast)
        if self.is_main:
            if not self.resume and not self.debug:
                assert not os.path.exists(self.log_root), 'Experiment log folder already exists!!'
                os.makedirs(self.log_root)
                with open(os.path.join(self.log_root, 'train_config.yaml'), 'w') as f:
                    f.write(str(cfg))
                if os.path.isfile(sys.argv[0]):
                    copyfile(sys.argv[0], os.path.join(self.log_root, 'train.py'))
            os.makedirs(self.log_root, exist_ok=True)
        if self.world > 1:
            torch.distributed.barrier()

        self.logger = self.get_logger()
        self.tb_writer = ScalarWriter(self.log_root) if self.is_main else None
        self.logger.info(f'Train Config:\n{cfg}')

        self.device = self.select_device(cfg)
        self.logger.info(f'rank {self.rank}/{self.world} on {self.device}')

        if 'seed' in cfg.experiment and cfg.experiment.seed is not None:
            set_random_seed(cfg.experiment.seed + self.rank)
            self.logger.info(f'Using specific random seed: {cfg.experiment.seed} (+rank)')

        self.transformers = self.get_transformers(cfg.dataset.transformer)
        self.collate_fn = self.get_collate_fn()
        self.datasets = self.get_dataset(cfg.dataset)
        self.dataloaders = self.get_dataloader(cfg.dataset)
        self.logger.info(f'Building model {cfg.model.name} ...')
        self.model = self.to_device(self.get_model(cfg.model), parallel=True)
        self.logger.info(f'Building model {cfg.model.name} OK!')

        self.criterion = self.get_criterion(cfg.train.criterion)
        self.optimizer = self.get_optimizer(cfg.train.optimizer)
        self.scheduler = self.get_scheduler(cfg.train.scheduler)
        # gradient synchronisation (after the optimiser so that requires_grad flags are final)
        self.reducer = ddp.GradientAllReducer(self.model) if self.world > 1 else None
        # a criterion that owns parameters (CINLoss.h) is data-parallel state too
        crit_trainable = isinstance(self.criterion, torch.nn.Module) and \
            any(p.requires_grad for p in self.criterion.parameters())
        self.criterion_reducer = ddp.GradientAllReducer(self.criterion) if (self.world > 1 and crit_trainable) else None

        if self.resume:
            self.logger.info(f'Resuming from `{self.resume}`')
            self.load_checkpoint(cfg.experiment.resume)

        self.performance_meters = self.get_performance_meters()
        self.average_meters = self.get_average_meters()
        self.timer = Timer()
        self.logger.info('Training Preparation Done!')

    # ------------------------------------------------------------------ plumbing
    def select_device(self, cfg):
        """`experiment.cuda` non-empty list -> this process's GPU (LOCAL_RANK); [] / None or no GPU -> refuse: the HIP
        heads have no CPU path (the CPU reference lives in oracle/ as test infrastructure)."""
        want_gpu = isinstance(cfg.experiment.cuda, list) and len(cfg.experiment.cuda) > 0
        if not (want_gpu and torch.cuda.is_available()):
            raise RuntimeError('hawkeye_amd trains on MI355X only: set experiment.cuda: [0] and run on a GPU host')
        device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(device)
        return device

    def get_logger(self):
        logger = logging.getLogger()
        logger.handlers = []
        logger.setLevel(logging.INFO if self.is_main else logging.WARNING)
        h = TqdmHandler()
        h.setFormatter(logging.Formatter('[%(asctime)s] %(message)s'))
        logger.addHandler(h)
        if self.is_main:
            fh = logging.FileHandler(os.path.join(self.log_root, 'report.log'), encoding='utf8')
            fh.setFormatter(logging.Formatter('[%(asctime)s][%(levelname)s] %(message)s'))
            logger.addHandler(fh)
        return logger

    def get_performance_meters(self):
        return {'train': {m: PerformanceMeter(higher_is_better='loss' not in m) for m in ('acc', 'loss')},
                'val': {'acc': PerformanceMeter()}, 'val_first': {'acc': PerformanceMeter()}}

    def get_average_meters(self):
        return {m: AverageMeter() for m in ('acc', 'loss')}

    def reset_average_meters(self):
        for m in self.average_meters.values():
            m.reset()

    # ------------------------------------------------------------------ overridable builders
    def get_model(self, config):
        model = MODEL.get(config.name)(config)
        if 'load' in config and config.load != '':
            self.logger.info(f'Loading model from {config.load}')
            model.load_state_dict(torch.load(config.load, map_location='cpu'))
        return model

    def get_transformers(self, config):
        """The reference's presets (train.py:171-183): RandomResizedCrop + flip + TrivialAugmentWide + RandomErasing(0.1)
        for training, Resize + CenterCrop for validation - hawkeye_amd/transforms.py (no torchvision in the image)."""
        from . import transforms
        resize = config['resize_size'] if 'resize_size' in config else int(config['image_size'] * 8 / 7)
        # `device_finalize: True`: workers ship uint8 crops; float conversion + normalise + erase run on the GPU
        dev = bool(config['device_finalize']) if 'device_finalize' in config else False
        return {'train': transforms.ClassificationPresetTrain(crop_size=config['image_size'], auto_augment_policy='ta_wide',
                                                              random_erase_prob=0.1, device_finalize=dev),
                'val': transforms.ClassificationPresetEval(crop_size=config['image_size'], resize_size=resize,
                                                           device_finalize=dev)}

    def get_collate_fn(self):
        return {'train': None, 'val': None}

    def get_dataset(self, config):
        from . import data
        if config.name == 'synthetic':
            n_cls = self.config.model.num_classes
            size = config.transformer.image_size
            return {s: data.SyntheticDataset(config.samples if 'samples' in config else 64 * config.batch_size,
                                             size, n_cls, seed=i) for i, s in enumerate(('train', 'val'))}
        return {s: data.FGDataset(config.root_dir, os.path.join(config.meta_dir, s + '.txt'), self.transformers[s])
                for s in ('train', 'val')}

    def get_dataloader(self, config):
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        loaders = {}
        for s in ('train', 'val'):
            sampler = None
            if self.world > 1 and s == 'train':
                sampler = DistributedSampler(self.datasets[s], num_replicas=self.world, rank=self.rank, shuffle=True)
            elif self.world > 1:
                # validation: indices rank, rank + world, ... with NO padding (DistributedSampler repeats samples until
                # the set divides by the world size; the duplicates would be counted by the all-reduced meters and the
                # accuracy that drives ReduceLROnPlateau / best_model.pth would differ from the reference's
                # single-process number over the same set)
                sampler = list(range(self.rank, len(self.datasets[s]), self.world))
            loaders[s] = DataLoader(self.datasets[s], config.batch_size, num_workers=config.num_workers,
                                    pin_memory=True, shuffle=(s == 'train' and sampler is None), sampler=sampler,
                                    collate_fn=self.collate_fn[s])
        return loaders

    def get_criterion(self, config):
        return torch.nn.CrossEntropyLoss(label_smoothing=0.1)

    def get_optimizer(self, config):
        return torch.optim.Adam(self.model.parameters(), lr=config.lr, weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, config.T_max, config.eta_min)

    def to_device(self, m, parallel=False):
        """`experiment.channels_last: True` keeps images / conv weights in NHWC (MIOpen's fp32 kernels on gfx950 are
        NHWC implicit-GEMMs: this removes its layout transposes, +10 % on the BCNN step - DESIGN.md section 5; it is also the
        layout in which the VGG trunk runs its fused epilogues, csrc/trunk.hip: another +12 %)."""
        cl = 'channels_last' in self.config.experiment and self.config.experiment.channels_last
        if isinstance(m, dict) and 'u8' in m:              # uint8 crops from the workers (transformer.device_finalize)
            from . import functional as HF
            return HF.image_finalize(m['u8'].to(self.device, non_blocking=True), m['erase'].to(self.device, non_blocking=True),
                                     channels_last=bool(cl))
        if isinstance(m, torch.Tensor):
            m = m.to(self.device, non_blocking=True)
            return m.contiguous(memory_format=torch.channels_last) if (cl and m.dim() == 4) else m
        m = m.to(self.device)
        return m.to(memory_format=torch.channels_last) if cl else m

    def get_model_module(self, model=None):
        return self.model if model is None else model       # never wrapped: attribute access stays direct

    # ------------------------------------------------------------------ step helpers
    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
            if self.criterion_reducer is not None:
                self.criterion_reducer.zero_grad()
        else:
            self.optimizer.zero_grad()

    def backward_and_step(self, loss):
        """zero_grad -> backward (all-reduce overlapped) -> join -> optimizer.step()."""
        self.zero_grad()
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
            if self.criterion_reducer is not None:
                self.criterion_reducer.finish()
        self.optimizer.step()

    # ------------------------------------------------------------------ loops
    @emergency_save
    def train(self):
        config = self.config.train
        if 'val_first' in config and config.val_first:
            self.logger.info('Validate model before training.')
            self.validate()
            self.performance_meters['val_first']['acc'].update(self.average_meters['acc'].avg)
            self.report(epoch=0, split='val_first')
        self.model.train()
        for epoch in range(self.start_epoch, self.total_epoch):
            self.epoch = epoch
            self.reset_average_meters()
            self._on_start_epoch()
            self.logger.info(f'Starting epoch {epoch + 1} ...')
            self.timer.tick()
            loader = self.dataloaders['train']
            if hasattr(loader.sampler, 'set_epoch'):
                loader.sampler.set_epoch(epoch)
            for data in loader:
                self._on_start_forward()
                self.batch_training(data)
                self._on_end_forward()
            dt = self.timer.tick()
            self.logger.info(f'Training epoch {epoch + 1} took {dt:.1f}s')
            self.sync_average_meters()
            self.performance_meters['train']['acc'].update(self.average_meters['acc'].avg)
            self.performance_meters['train']['loss'].update(self.average_meters['loss'].avg)
            self.report(epoch=epoch + 1, split='train')

            self.reset_average_meters()
            self.validate()
            self.model.train()
            val_acc = self.average_meters['acc'].avg             # over the WHOLE validation set (validate() syncs ranks)
            m = self.performance_meters['val']['acc']
            is_best = epoch >= 5 and (not m.values or val_acc > m.best_value)     # train.py:284-288
            m.update(val_acc)
            self.report(epoch=epoch + 1, split='val')
            self.do_scheduler_step()

            if self.is_main:
                if epoch != 0 and (epoch + 1) % config.save_frequence == 0:       # train.py:296
                    self.save_model()
                if is_best:
                    self.save_model('best_model.pth')
            self._on_end_epoch()
        self.logger.info('Training done.')

    def sync_average_meters(self):
        """Multi-rank: every AverageMeter becomes the average over ALL ranks' samples (sum and count all-reduced), so
        that the logged metrics, the scheduler (ReduceLROnPlateau steps on val acc) and the best-model choice see what
        the reference's single-process DataParallel run computes over the full set - identically on every rank."""
        if self.world <= 1:
            return
        names = sorted(self.average_meters)
        buf = torch.tensor([[self.average_meters[n].sum, self.average_meters[n].count] for n in names],
                           dtype=torch.float64, device=self.device)
        torch.distributed.all_reduce(buf)
        for n, (s_, c_) in zip(names, buf.tolist()):
            meter = self.average_meters[n]
            meter.sum, meter.count = s_, c_
            meter.avg = s_ / c_ if c_ else 0.0

    def batch_training(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        outputs = self.model(images)
        loss = self.criterion(outputs, labels)
        self.backward_and_step(loss)
        self.average_meters['acc'].update(accuracy(outputs, labels, 1), images.size(0))
        self.average_meters['loss'].update(loss.item(), images.size(0))

    def validate(self):
        self.model.train(False)
        with torch.no_grad():
            for data in self.dataloaders['val']:
                self.batch_validate(data)
        self.sync_average_meters()
        self.model.train(True)

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        self.average_meters['acc'].update(accuracy(self.model(images), labels, 1), images.size(0))

    def do_scheduler_step(self):
        if self.scheduler is not None:
            self.scheduler.step()

    def report(self, epoch, split='train'):
        for metric, meter in self.performance_meters[split].items():
            self.logger.info(f'[{split}] epoch {epoch} {metric}: {meter.current_value:.4f} (best {meter.best_value:.4f})')
            if self.tb_writer is not None:
                self.tb_writer.add_scalar(f'{split}/{metric}', meter.current_value, epoch)

    # ------------------------------------------------------------------ checkpoints
    def save_model(self, name=None):
        name = name or f'{self.config.model.name}_epoch_{self.epoch + 1}.pth'
        torch.save(self.model.state_dict(), os.path.join(self.log_root, name))     # plain keys (never `module.`-prefixed)

    def save_checkpoint(self):
        if not self.is_main:
            return
        torch.save({'epoch': self.epoch, 'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                    'scheduler': self.scheduler.state_dict() if self.scheduler else None},
                   os.path.join(self.log_root, f'checkpoint_epoch_{self.epoch}.pth'))        # train.py:379

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location='cpu')
        self.start_epoch = ck['epoch']          # the interrupted epoch is redone from its start, as in train.py:391
        self.model.load_state_dict(ck['model'])
        self.optimizer.load_state_dict(ck['optimizer'])
        if self.scheduler is not None and ck.get('scheduler') is not None:
            self.scheduler.load_state_dict(ck['scheduler'])

    # ------------------------------------------------------------------ hooks
    def _on_start_epoch(self):
        self.on_start_epoch(self.config.hook.on_start_epoch if 'hook' in self.config and
                            'on_start_epoch' in self.config.hook else None)

    def _on_end_epoch(self):
        self.on_end_epoch(self.config.hook.on_end_epoch if 'hook' in self.config and
                          'on_end_epoch' in self.config.hook else None)

    def _on_start_forward(self):
        self.on_start_forward(self.config.hook.on_start_forward if 'hook' in self.config and
                              'on_start_forward' in self.config.hook else None)

    def _on_end_forward(self):
        self.on_end_forward(self.config.hook.on_end_forward if 'hook' in self.config and
                            'on_end_forward' in self.config.hook else None)

    def on_start_forward(self, config):
        pass

    def on_end_forward(self, config):
        pass

    def on_start_epoch(self, config):
        pass

    def on_end_epoch(self, config):
        pass


if __name__ == '__main__':
    Trainer().train()
