"""Fast MPN-COV plugin (mirrors model/methods/MPNCOV.py:23-102): ResNet-50 trunk ->
1x1 dimension reduction (PyTorch-ROCm) -> covariance -> Newton-Schulz sqrt ->
upper-triangle vector, the last three on the gfx950 kernels."""
import torch.nn as nn

from ... import functional as HF
from ..backbone import resnet50
from ..registry import MODEL
from ..utils import wide_linear


class MPNCOV(nn.Module):
    def __init__(self, iter_num=3, is_sqrt=True, is_vec=True, input_dim=2048, dimension_reduction=None):
        super().__init__()
        self.iterNum, self.is_sqrt, self.is_vec, self.dr = iter_num, is_sqrt, is_vec, dimension_reduction
        if self.dr is not None:
            self.conv_dr_block = nn.Sequential(
                nn.Conv2d(input_dim, self.dr, kernel_size=1, stride=1, bias=False),
                nn.BatchNorm2d(self.dr),
                nn.ReLU(inplace=True))
        d = self.dr if self.dr else input_dim
        self.output_dim = int(d * (d + 1) / 2) if is_vec else int(d * d)
        for m in self.modules():                                   # MPNCOV.py:77-83
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if self.dr is not None:
            x = self.conv_dr_block(x)
        x = HF.covpool(x)
        if self.is_sqrt and self.is_vec:                       # MPNCOV.py:88-92 back to back: one chain of launches
            return HF.sqrtm_triuvec(x, self.iterNum, symmetric=True)
        if self.is_sqrt:
            x = HF.sqrtm(x, self.iterNum, symmetric=True)      # a covariance: symmetric by construction
        if self.is_vec:
            x = HF.triuvec(x)
        return x


@MODEL.register
class MPN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.backbone = nn.Sequential(*list(resnet50(pretrained=True).children())[:-2])
        self.pool = MPNCOV(config.iter_num, config.is_sqrt, config.is_vec, config.input_dim,
                           config.dimension_reduction)
        self.classifier = nn.Linear(self.pool.output_dim, config.num_classes)

    def forward(self, x):
        x = self.pool(self.backbone(x))
        return wide_linear(self.classifier, x.view(x.size(0), -1))
