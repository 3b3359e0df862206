"""VGG-16 trunk on PyTorch-ROCm (MIOpen convolutions).  Module structure and
state_dict keys equal torchvision's / the reference's (model/backbone/vgg.py:24-86):
`features.{0,2,5,...,28}.{weight,bias}`, `classifier.{0,3,6}.*`."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from ..registry import BACKBONE
from ..utils import initialize_weights
from . import pretrained as _pre

VGG16_LAYOUT = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')


def _plain_pool(m):
    two = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    return (isinstance(m, nn.MaxPool2d) and two(m.kernel_size) == (2, 2) and two(m.stride) == (2, 2) and two(m.padding) == (0, 0)
            and two(m.dilation) == (1, 1) and not m.ceil_mode and not m.return_indices)


class ConvStack(nn.Sequential):
    """The VGG `features` stack (model/backbone/vgg.py:24-57): the same children under the same indices - `state_dict` keys,
    `children()` and slicing are nn.Sequential's - with one difference in how it RUNS on an MI355X: behind every
    convolution the bias add, the ReLU and, at the end of a stage, the 2 x 2 max-pool are one pass over the activation
    instead of one pass per op (hk_bias_relu_*, csrc/trunk.hip: the framework's elementwise kernels around the
    convolutions are 12.5 % of the BCNN training step, and the full-resolution activation in front of a pool is not
    even written).  The convolutions themselves are MIOpen's.  Anything the kernels do not cover - CPU tensors, NCHW
    memory, odd map sizes, other dtypes, a child with forward hooks - runs the children one by one as nn.Sequential does."""

    def forward(self, x):
        from ... import functional as HF
        from ..utils import _hooked
        mods = list(self)
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last)) or any(_hooked(m) for m in mods):
            return super().forward(x)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.Conv2d) and m.bias is not None and m.padding_mode == 'zeros' and i + 1 < len(mods)
                    and isinstance(mods[i + 1], nn.ReLU)):
                if not (i + 2 < len(mods) and _plain_pool(mods[i + 2])) and HF.conv1_ok(x, m):
                    # the first layer (3 -> 64 channels): convolution, bias and ReLU in ONE kernel - it is the write of its output
                    x = HF.conv1_bias_relu(x, m.weight, m.bias)
                    i += 2
                    continue
                y = TF.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
                pool = i + 2 < len(mods) and _plain_pool(mods[i + 2])
                if pool and HF.trunk_epilogue_ok(y, pool=True):
                    x = HF.bias_relu_pool(y, m.bias)
                    i += 3
                    continue
                if HF.trunk_epilogue_ok(y):
                    x = HF.bias_relu(y, m.bias)
                    i += 2
                    continue
                x = mods[i + 1](y + m.bias.view(1, -1, 1, 1))
                i += 2
                continue
            x = m(x)
            i += 1
        return x


def conv_stack(layout):
    mods, cin = [], 3
    for item in layout:
        if item == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        mods.append(nn.Conv2d(cin, item, kernel_size=3, padding=1))
        mods.append(nn.ReLU(inplace=True))
        cin = item
    return ConvStack(*mods)


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
