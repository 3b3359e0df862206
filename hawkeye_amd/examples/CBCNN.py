"""CBCNN trainer: stage-1 freeze is done here, not in the model (reference Examples/CBCNN.py:13-33)."""
import torch

from ..train import Trainer
from .common import warmup_cosine


class CBCNNTrainer(Trainer):
    def get_model(self, config):
        model = super().get_model(config)
        if config.stage == 1:
            for p in model.backbone.parameters():
                p.requires_grad = False
        return model

    def get_optimizer(self, config):
        params = self.model.classifier.parameters() if self.config.model.stage == 1 else self.model.parameters()
        return torch.optim.SGD(params, lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return warmup_cosine(self.optimizer, config)


if __name__ == '__main__':
    CBCNNTrainer().train()
