"""Pieces shared by the example trainers."""
import torch
from torch.utils.data import DataLoader

from ..data import BalancedBatchSampler
from ..train import Trainer
from ..utils import accuracy


def warmup_cosine(optimizer, config):
    """Linear warm-up then cosine annealing, stepped per epoch (Examples/CBCNN.py:35-45, MPN.py:20-30)."""
    main = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config['T_max'] - config['warmup_epochs'])
    warm = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=config['lr_warmup_decay'],
                                             total_iters=config['warmup_epochs'])
    return torch.optim.lr_scheduler.SequentialLR(optimizer, schedulers=[warm, main],
                                                 milestones=[config['warmup_epochs']])


def lr_groups(model, slow_attr, base_lr, slow_factor):
    """Two optimiser groups: `model.<slow_attr>` at `slow_factor * base_lr`, every other parameter at `base_lr`."""
    slow = list(getattr(model, slow_attr).parameters())
    slow_ids = {id(p) for p in slow}
    fast = [p for p in model.parameters() if id(p) not in slow_ids]
    return [{'params': slow, 'lr': slow_factor * base_lr}, {'params': fast, 'lr': base_lr}]


class PairBatchTrainer(Trainer):
    """Trainer for the methods whose criterion needs same-class images inside every batch (OSMENet's MAMC n-pairs
    loss, CIN's contrastive halves): when the dataset section of the yaml carries `n_classes` and `n_samples`, the
    training loader draws `n_classes x n_samples` class-balanced batches (Examples/OSMENet.py:17-31, CIN.py:17-30);
    the model returns `(logits, extra)` in training, the criterion consumes the pair, accuracy is taken on the logits.
    Subclasses provide `get_criterion` / `get_optimizer`."""

    def get_dataloader(self, config):
        loaders = super().get_dataloader(config)
        if 'n_classes' not in config or 'n_samples' not in config:
            return loaders
        exp = self.config.experiment
        seed = exp.seed if 'seed' in exp and exp.seed is not None else 0
        train_set = self.datasets['train']
        batches = BalancedBatchSampler(train_set.labels, config.n_classes, config.n_samples, seed=seed, rank=self.rank)
        loaders['train'] = DataLoader(train_set, batch_sampler=batches, num_workers=config.num_workers, pin_memory=True,
                                      collate_fn=self.collate_fn['train'])
        return loaders

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)

    def _batch(self, data):
        return self.to_device(data['img']), self.to_device(data['label'])

    def batch_training(self, data):
        images, labels = self._batch(data)
        outputs = self.model(images)
        loss = self.criterion(outputs, labels)
        self.backward_and_step(loss)
        count = images.size(0)
        self.average_meters['acc'].update(accuracy(outputs[0], labels, 1), count)
        self.average_meters['loss'].update(loss.item(), count)

    def batch_validate(self, data):
        images, labels = self._batch(data)
        out = self.model(images)
        logits = out[0] if isinstance(out, tuple) else out
        self.average_meters['acc'].update(accuracy(logits, labels, 1), images.size(0))
