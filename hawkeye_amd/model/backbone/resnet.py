"""ResNet-50/101 trunks on PyTorch-ROCm (MIOpen).  torchvision-v1.5 layout (stride on
the 3x3 conv), same child order and parameter names as the reference's
model/backbone/resnet.py:89-252 so checkpoints load both ways."""
import torch.nn as nn

from ..registry import BACKBONE, MODEL
from ..utils import load_state_dict
from . import pretrained as _pre


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        # `y += idt; relu(y)` as one pass over the block's output on an MI355X (hk_add_relu_fwd, csrc/trunk.hip: two reads and a
        # write where add_ + relu_ make three and two); the framework's ops for CPU tensors, other dtypes, a hooked ReLU
        from ... import functional as HF
        from ..utils import _hooked
        if HF.add_relu_ok(y, idt) and not _hooked(self.relu):
            return HF.add_relu(y, idt)
        y += idt
        return self.relu(y)


def make_stage(block, inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * block.expansion:
        down = nn.Sequential(
            nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * block.expansion))
    layers = [block(inplanes, planes, stride, down)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers), planes * block.expansion


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        c = 64
        self.layer1, c = make_stage(block, c, 64, layers[0], 1)
        self.layer2, c = make_stage(block, c, 128, layers[1], 2)
        self.layer3, c = make_stage(block, c, 256, layers[2], 2)
        self.layer4, c = make_stage(block, c, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(c, num_classes)
        for m in self.modules():                       # resnet.py:190-195
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def _build(arch, layers, pretrained, **kwargs):
    model = ResNet(Bottleneck, layers, **kwargs)
    if pretrained:
        sd = _pre.load(arch)
        if sd is not None:
            load_state_dict(model, sd)                 # lenient, like resnet.py:262-266
    return model


@BACKBONE.register
def resnet50(pretrained=False, progress=True, **kwargs):
    return _build('resnet50', [3, 4, 6, 3], pretrained, **kwargs)


@BACKBONE.register
def resnet101(pretrained=False, progress=True, **kwargs):
    return _build('resnet101', [3, 4, 23, 3], pretrained, **kwargs)


# The plain classifiers behind the reference's default config (configs/Baseline.yaml, model.name ResNet50): registered
# as MODELs taking the config node, like model/backbone/resnet.py:403-412.
@MODEL.register
def ResNet50(config):
    return resnet50(pretrained=config.pretrained if 'pretrained' in config else True, num_classes=config.num_classes)


@MODEL.register
def ResNet101(config):
    return resnet101(pretrained=config.pretrained if 'pretrained' in config else True, num_classes=config.num_classes)
