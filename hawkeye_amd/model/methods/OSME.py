"""OSME plugin (mirrors model/methods/OSME.py:8-64): ResNet-101 trunk, P excitation
gates, P fully connected part heads.  Squeeze (GAP) and the channel re-scaling for
all P gates run on the gfx950 kernels (hk_osme_gap / hk_osme_scale_*)."""
import torch
import torch.nn as nn

from ... import functional as HF
from ..backbone import resnet101
from ..registry import MODEL


class OSME_block(nn.Module):
    def __init__(self, channels, ratio):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)            # kept for attribute parity; GAP runs in hk_osme_gap
        self.block = nn.Sequential(
            nn.Linear(channels, channels // ratio), nn.ReLU(inplace=True),
            nn.Linear(channels // ratio, channels), nn.Sigmoid())

    def gate(self, z):
        return self.block(z)

    def forward(self, x):
        return HF.osme_scale(x, self.block(HF.osme_gap(x)).unsqueeze(0))[0]


class OSME(nn.Module):
    def __init__(self, in_channels, out_channels=1024, feature_shape=(7, 7), num_attention=2):
        super().__init__()
        hw = feature_shape[0] * feature_shape[1] if isinstance(feature_shape, tuple) else feature_shape * feature_shape
        self.blocks = nn.ModuleList([OSME_block(in_channels, 16) for _ in range(num_attention)])
        self.fcs = nn.ModuleList([nn.Linear(in_channels * hw, out_channels) for _ in range(num_attention)])

    def forward(self, x):
        n = x.size(0)
        z = HF.osme_gap(x)                                           # one squeeze shared by all gates
        m = torch.stack([blk.gate(z) for blk in self.blocks], dim=0)
        s = HF.osme_scale(x, m)                                      # [P,N,C,H,W], one pass over x
        feats = [fc(s[i].reshape(n, -1)) for i, fc in enumerate(self.fcs)]
        return sum(feats), torch.stack(feats, dim=1)


@MODEL.register
class OSMENet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_attention = config.num_attention
        self.num_classes = config.num_classes
        self.backbone = nn.Sequential(*list(resnet101(pretrained=True).children())[:-2])
        self.osme = OSME(2048, 1024, feature_shape=7, num_attention=self.num_attention)
        self.classifier = nn.Linear(1024, self.num_classes)

    def forward(self, x):
        x1, x_part = self.osme(self.backbone(x))
        return self.classifier(x1), x_part
