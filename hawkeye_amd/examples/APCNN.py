"""AP-CNN trainer: loss = sum of CE over the 8 logits, two LR groups split at children()[7]
(reference Examples/APCNN.py:36-59; note `relu`/`maxpool` count as children, so layer4 gets the head LR),
cosine schedule applied by hand at epoch start (:68-83)."""
import numpy as np
import torch
import torch.nn as nn

from ..train import Trainer
from ..utils import accuracy


class APCNNTrainer(Trainer):
    def get_optimizer(self, config):
        kids = list(self.get_model_module().children())
        return torch.optim.SGD([
            {'params': nn.Sequential(*kids[7:]).parameters(), 'lr': config.lr},
            {'params': nn.Sequential(*kids[:7]).parameters(), 'lr': config.lr / 10},
        ], momentum=0.9, weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return None

    def do_scheduler_step(self):
        pass

    def on_start_epoch(self, config):
        n, lr = self.config.train.epoch, self.config.train.optimizer.lr
        cur = float(lr / 2 * (np.cos(np.pi * (self.epoch % n) / n) + 1))
        self.optimizer.param_groups[0]['lr'] = cur
        self.optimizer.param_groups[1]['lr'] = cur / 10

    def batch_training(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        logits_mean, logits_list, _, _ = self.model(images, labels)
        loss = sum(self.criterion(l, labels) for l in logits_list)
        self.backward_and_step(loss)
        self.average_meters['acc'].update(accuracy(logits_mean, labels, 1), images.size(0))
        self.average_meters['loss'].update(loss.item(), images.size(0))

    def batch_validate(self, data):
        images, labels = self.to_device(data['img']), self.to_device(data['label'])
        self.average_meters['acc'].update(accuracy(self.model(images, labels)[0], labels, 1), images.size(0))


if __name__ == '__main__':
    APCNNTrainer().train()
