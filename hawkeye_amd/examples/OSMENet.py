"""OSMENet trainer (reference Examples/OSMENet.py): class-balanced batches (`dataset.n_classes` x
`dataset.n_samples`, :17-31), the MAMC criterion - cross entropy + lambda_a * n-pairs over the part features (:33,
:58-62) on the HIP kernel - and the backbone at 0.1x LR (:35-43)."""
import torch
from torch.utils.data import DataLoader

from ..data import BalancedBatchSampler
from ..model.loss import MAMCLoss
from ..train import Trainer
from ..utils import accuracy
from .common import warmup_cosine


class OSMENetTrainer(Trainer):
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
        return MAMCLoss(config)

    def get_optimizer(self, config):
        model = self.get_model_module()
        ids = set(map(id, model.backbone.parameters()))
        rest = [p for p in model.parameters() if id(p) not in ids]
        return torch.optim.SGD([
            {'params': model.backbone.parameters(), 'lr': 0.1 * config.lr},
            {'params': rest, 'lr': config.lr},
        ], weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)

    def batch_training(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        pred, parts = self.model(images)
        loss = self.criterion((pred, parts), labels)
        self.backward_and_step(loss)
        self.average_meters['acc'].update(accuracy(pred, labels, 1), images.size(0))
        self.average_meters['loss'].update(loss.item(), images.size(0))

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        self.average_meters['acc'].update(accuracy(self.model(images)[0], labels, 1), images.size(0))


if __name__ == '__main__':
    OSMENetTrainer().train()
