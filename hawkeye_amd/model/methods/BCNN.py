"""Bilinear CNN plugin (mirrors model/methods/BCNN.py:8-55).

Same constructor contract (`config.stage` optional, `config.num_classes`), same
attributes (`backbone`, `bilinear_pooling`, `classifier`, `stage`) and the same 28
state_dict keys; the pooling itself runs on the gfx950 kernels
(hk_bcnn_pool_fwd/bwd) instead of bmm + 7 elementwise ATen ops."""
import torch.nn as nn

from ... import functional as HF
from ..backbone import vgg16
from ..registry import MODEL
from ..utils import initialize_weights


class BilinearPooling(nn.Module):
    """[B,C,H,W] -> [B,C*C]: sqrt(X X^T / HW + 1e-5) then l2-normalise (BCNN.py:13-27)."""

    def forward(self, x):
        return HF.bilinear_pool(x)


@MODEL.register
class BCNN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.stage = config.stage if 'stage' in config else 2     # BCNN.py:36
        # all 31 layers of VGG-16 `features`, last MaxPool included (BCNN.py:38-39)
        self.backbone = nn.Sequential(*list(vgg16(pretrained=True).features.children()))
        self.bilinear_pooling = BilinearPooling()
        self.classifier = nn.Linear(512 ** 2, config.num_classes)
        self.classifier.apply(initialize_weights)
        if self.stage == 1:
            for p in self.backbone.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.backbone(x)
        if self.stage == 1:
            x = x.detach()
        return self.classifier(self.bilinear_pooling(x))
