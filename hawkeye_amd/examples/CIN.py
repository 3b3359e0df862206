"""CIN trainer (reference Examples/CIN.py): class-balanced batches whose two halves are the contrast pairs, and the
CINLoss criterion.  The criterion owns a Linear, so it is moved to the device and its parameters get an optimiser
group of their own (and, under data parallelism, their own gradient all-reduce: Trainer.criterion_reducer)."""
import torch

from ..model.loss import CINLoss
from .common import PairBatchTrainer


class CINTrainer(PairBatchTrainer):
    def get_criterion(self, config):
        return CINLoss(config).to(self.device)

    def get_optimizer(self, config):
        groups = [{'params': list(owner.parameters()), 'lr': config.lr} for owner in (self.model, self.criterion)]
        return torch.optim.SGD(groups, weight_decay=config.weight_decay)


if __name__ == '__main__':
    CINTrainer().train()
