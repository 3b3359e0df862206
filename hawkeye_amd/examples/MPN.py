"""Fast MPN-COV trainer: Adam, backbone at 0.2x the head learning rate (reference Examples/MPN.py:13-18)."""
import torch

from ..train import Trainer
from .common import warmup_cosine


class MPNTrainer(Trainer):
    def get_optimizer(self, config):
        return torch.optim.Adam([
            {'params': self.model.classifier.parameters(), 'lr': config.lr},
            {'params': self.model.pool.parameters(), 'lr': config.lr},
            {'params': self.model.backbone.parameters(), 'lr': 0.2 * config.lr},
        ], weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)


if __name__ == '__main__':
    MPNTrainer().train()
