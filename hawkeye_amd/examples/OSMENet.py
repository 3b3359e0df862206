"""OSMENet trainer: backbone at 0.1x LR (reference Examples/OSMENet.py:35-43).  The MAMC n-pairs loss and the
balanced batch sampler are out of this round's scope: plain cross entropy on the
classifier logits is used."""
import torch

from ..train import Trainer
from ..utils import accuracy
from .common import warmup_cosine


class OSMENetTrainer(Trainer):
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
        pred, _ = self.model(images)
        loss = self.criterion(pred, labels)
        self.backward_and_step(loss)
        self.average_meters['acc'].update(accuracy(pred, labels, 1), images.size(0))
        self.average_meters['loss'].update(loss.item(), images.size(0))

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        self.average_meters['acc'].update(accuracy(self.model(images)[0], labels, 1), images.size(0))


if __name__ == '__main__':
    OSMENetTrainer().train()
