"""CIN trainer (reference Examples/CIN.py): class-balanced batches whose two halves form the contrast pairs, the
CINLoss criterion (it owns a Linear, so it lives on the device and its parameters get their own optimiser group,
:33-42) and warm-up + cosine schedule."""
import torch
from torch.utils.data import DataLoader

from ..data import BalancedBatchSampler
from ..model.loss import CINLoss
from ..train import Trainer
from ..utils import accuracy
from .common import warmup_cosine


class CINTrainer(Trainer):
    def get_dataloader(self, config):
        loaders = super().get_dataloader(config)
        if 'n_classes' in config and 'n_samples' in config:
            exp = self.config.experiment
            sampler = BalancedBatchSampler(self.datasets['train'].labels, config.n_classes, config.n_samples,
                                           seed=exp.seed if 'seed' in exp and exp.seed is not None else 0, rank=self.rank)
            loaders['train'] = DataLoader(self.datasets['train'], num_workers=config.num_workers, pin_memory=True,
                                          batch_sampler=sampler, collate_fn=self.collate_fn['train'])
        return loaders

    def get_criterion(self, config):
        return CINLoss(config).to(self.device)

    def get_optimizer(self, config):
        return torch.optim.SGD([
            {'params': self.model.parameters(), 'lr': config.lr},
            {'params': self.criterion.parameters(), 'lr': config.lr},
        ], weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)

    def batch_training(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        outputs = self.model(images)
        loss = self.criterion(outputs, labels)
        self.backward_and_step(loss)
        self.average_meters['acc'].update(accuracy(outputs[0], labels, 1), images.size(0))
        self.average_meters['loss'].update(loss.item(), images.size(0))

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        self.average_meters['acc'].update(accuracy(self.model(images), labels, 1), images.size(0))


if __name__ == '__main__':
    CINTrainer().train()
