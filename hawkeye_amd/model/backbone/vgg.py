"""VGG-16 trunk on PyTorch-ROCm (MIOpen convolutions).  Module structure and
state_dict keys equal torchvision's / the reference's (model/backbone/vgg.py:24-86):
`features.{0,2,5,...,28}.{weight,bias}`, `classifier.{0,3,6}.*`."""
import torch.nn as nn

from ..registry import BACKBONE
from ..utils import initialize_weights
from . import pretrained as _pre

VGG16_LAYOUT = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')


def conv_stack(layout):
    mods, cin = [], 3
    for item in layout:
        if item == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        mods.append(nn.Conv2d(cin, item, kernel_size=3, padding=1))
        mods.append(nn.ReLU(inplace=True))
        cin = item
    return nn.Sequential(*mods)


class VGG(nn.Module):
    def __init__(self, features, num_classes=1000, init_weights=True):
        super().__init__()
        self.features = features
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, num_classes))
        if init_weights:
            self.apply(initialize_weights)

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(x.flatten(1))


@BACKBONE.register
def vgg16(pretrained=False, progress=True, **kwargs):
    model = VGG(conv_stack(VGG16_LAYOUT), **kwargs)
    if pretrained:
        sd = _pre.load('vgg16')
        if sd is not None:
            model.load_state_dict(sd)      # strict, like vgg.py:85
    return model
