"""BCNN trainer: stage 1 trains the classifier only, stage 2 everything; SGD + ReduceLROnPlateau on val acc
(reference Examples/BCNN.py:32-48)."""
import torch

from ..train import Trainer


class BCNNTrainer(Trainer):
    def get_optimizer(self, config):
        model = self.get_model_module()
        params = model.classifier.parameters() if self.config.model.stage == 1 else model.parameters()
        return torch.optim.SGD(params, lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)

    def get_scheduler(self, config):
        return torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='max', factor=0.1, patience=3,
                                                          threshold=1e-4)

    def do_scheduler_step(self):
        self.scheduler.step(self.performance_meters['val']['acc'].current_value)


if __name__ == '__main__':
    BCNNTrainer().train()
