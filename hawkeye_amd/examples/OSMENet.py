"""OSMENet trainer (reference Examples/OSMENet.py): class-balanced batches, the MAMC criterion - cross entropy +
lambda_a * n-pairs over the part features, on the HIP kernel - and the ResNet trunk at a tenth of the learning rate."""
import torch

from ..model.loss import MAMCLoss
from .common import PairBatchTrainer, lr_groups


class OSMENetTrainer(PairBatchTrainer):
    def get_criterion(self, config):
        return MAMCLoss(config)

    def get_optimizer(self, config):
        groups = lr_groups(self.get_model_module(), 'backbone', config.lr, 0.1)
        return torch.optim.SGD(groups, weight_decay=config.weight_decay)


if __name__ == '__main__':
    OSMENetTrainer().train()
