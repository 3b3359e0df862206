"""CIN criterion with the reference's contract and arithmetic (model/loss/CIN_loss.py:7-50): cross entropy on the
logits plus `alpha` times the contrastive term computed from `h(vec(Z_CCI))`.  The reference's arithmetic is kept as
it stands, including two things that look unintended but define what its checkpoints were trained with: the pair
mask compares the first half of the labels with the single label `target[B//2]` (:40), and the margin term is
overwritten by the square of the positive term (:42-44), so `beta` never enters the value.  A handful of tiny
reductions: plain PyTorch-ROCm ops, no kernel of ours."""
import torch
import torch.nn as nn

from ..utils import initialize_weights


class CINLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.alpha = config.alpha if 'alpha' in config else 2.0
        self.beta = config.beta if 'beta' in config else 0.5
        self.channel = config.channel if 'channel' in config else 2048
        self.feature_size = config.feature_size if 'feature_size' in config else 7 * 7
        self.r_channel = config.r_channel if 'r_channel' in config else 512
        self.pdist = nn.PairwiseDistance(p=2)
        self.ce_loss = nn.CrossEntropyLoss(label_smoothing=0.1)
        self.h = nn.Linear(self.channel * self.feature_size, self.r_channel)
        self.apply(initialize_weights)

    def forward(self, output, target):
        if not isinstance(output, tuple):
            return self.ce_loss(output, target)
        logits, z_cci = output
        batch = z_cci.size(0)
        half = batch // 2
        embed = self.h(z_cci.reshape(batch, -1))
        same = target[:half] == target[half]
        positive = torch.sum(self.pdist(embed[:half][same], embed[half:][same]) ** 2)
        contrastive = positive + positive ** 2
        return self.ce_loss(logits, target) + self.alpha * contrastive
