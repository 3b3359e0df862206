"""MAMC loss for OSMENet on the MI355X path - the reference's `model/loss/MAMC_loss.py` contract:
`MAMCLoss(config)((pred, parts), targets)` with `config.lambda_a` (default 0.5) and `config.use_mamc` (default True).
The n-pairs term (equation 11 of the MAMC paper, reference lines 34-90: a python loop over the B*P anchors) is one
call into the HIP library, loss and gradient together (`hawkeye_amd.functional.npairs_loss`)."""
import torch.nn as nn

from ... import functional as HF


class NPairsLoss(nn.Module):
    """inputs [B, P, D] part features, targets [B] labels -> scalar."""

    def forward(self, inputs, targets):
        return HF.npairs_loss(inputs, targets)


class MAMCLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.lambda_a = config.lambda_a if 'lambda_a' in config else 0.5
        self.use_mamc = config.use_mamc if 'use_mamc' in config else True
        self.ce_loss = nn.CrossEntropyLoss(label_smoothing=0.1)
        self.npair_loss = NPairsLoss()

    def forward(self, inputs, targets):
        pred, parts = inputs
        loss = self.ce_loss(pred, targets)
        if self.use_mamc:
            loss = loss + self.lambda_a * self.npair_loss(parts, targets)
        return loss
