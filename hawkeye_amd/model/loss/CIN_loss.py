"""CIN criterion with the reference's contract and arithmetic (model/loss/CIN_loss.py:7-50): cross entropy on the
logits plus `alpha` times the contrastive term computed from `h(vec(Z_CCI))`.  The reference's arithmetic is kept as
it stands, including two things that look unintended but define what its checkpoints were trained with: the pair
mask compares the first half of the labels with the single label `target[B//2]` (:40), and the margin term is
overwritten by the square of the positive term (:42-44), so `beta` never enters the value.  A handful of tiny
reductions: plain PyTorch-ROCm ops, no kernel of ours."""
import torch.nn as nn

from ..utils import initialize_weights


# option -> default (reference CIN_loss.py:10-19): weight of the contrastive term, its margin, shape of Z_CCI, width of h
_OPTIONS = {'alpha': 2.0, 'beta': 0.5, 'channel': 2048, 'feature_size': 7 * 7, 'r_channel': 512}


class CINLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        for option, default in _OPTIONS.items():
            setattr(self, option, config[option] if option in config else default)
        self.pdist = nn.PairwiseDistance(p=2)
        self.ce_loss = nn.CrossEntropyLoss(label_smoothing=0.1)
        self.h = nn.Linear(self.channel * self.feature_size, self.r_channel)
        self.apply(initialize_weights)

    def contrastive(self, z_cci, target):
        """Squared distances between the embeddings of the pairs (i, i + B/2) that the mask selects, plus its square."""
        half = z_cci.size(0) // 2
        embed = self.h(z_cci.flatten(1))
        same = target[:half] == target[half]
        positive = self.pdist(embed[:half][same], embed[half:][same]).pow(2).sum()
        return positive + positive ** 2

    def forward(self, output, target):
        if not isinstance(output, tuple):
            return self.ce_loss(output, target)
        logits, z_cci = output
        return self.ce_loss(logits, target) + self.alpha * self.contrastive(z_cci, target)
