"""Fast MPN-COV trainer.

Optimiser layout of the reference (Examples/MPN.py:13-18): Adam with three groups - the classifier and the pooling
head (1x1 reduction conv + BN) at the configured learning rate, the ResNet-50 trunk at one fifth of it - followed by
the shared linear-warm-up + cosine schedule (hawkeye_amd/examples/common.py).  The groups are built from the plain
attributes `classifier` / `pool` / `backbone`, which is why the model is never wrapped by the data-parallel layer.
"""
import torch

from ..train import Trainer
from .common import warmup_cosine

TRUNK_LR_RATIO = 0.2


class MPNTrainer(Trainer):
    def get_optimizer(self, config):
        net = self.model
        head_lr, trunk_lr = config.lr, TRUNK_LR_RATIO * config.lr
        groups = [dict(params=net.classifier.parameters(), lr=head_lr),
                  dict(params=net.pool.parameters(), lr=head_lr),
                  dict(params=net.backbone.parameters(), lr=trunk_lr)]
        return torch.optim.Adam(groups, weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)


if __name__ == '__main__':
    MPNTrainer().train()
