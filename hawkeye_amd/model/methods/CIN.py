"""Channel Interaction Network plugin on the MI355X path (contract of the reference's model/methods/CIN.py:
`ChannelInteractionModule` / `CINClassifier` / registered `CIN`, same attribute names, constructor arguments and
`state_dict` keys).  The interaction itself - softmax(-X X^T / HW) X and the contrastive |W_A - eta W_B| X between
the two halves of the batch - runs on the gfx950 kernels (hk_cin_sci_*, hk_cin_cci_*); the 3x3 convolution, the
residual sum, the 1-output fc that produces eta / gamma and the classifier stay on PyTorch-ROCm."""
import torch
import torch.nn as nn

from ... import functional as HF
from ..backbone import resnet50
from ..registry import MODEL
from ..utils import initialize_weights


class ChannelInteractionModule(nn.Module):
    def __init__(self, in_channel=2048, spatial_size=(7, 7)):
        super().__init__()
        self.in_channel, self.spatial_size = in_channel, spatial_size
        positions = spatial_size[0] * spatial_size[1]
        self.softmax = nn.Softmax()                       # parameter-free member of the reference (CIN.py:21), unused
        self.conv = nn.Conv2d(in_channel, in_channel, 3, 1, 1)
        self.fc = nn.Linear(2 * in_channel * positions, 1)

    def _refine(self, y, shape):
        b, c, w, h = shape
        return self.conv(y.view(b, c, w, h)).view(b, c, w * h)

    def forward(self, x):
        b, c, w, h = x.size()
        assert b % 2 == 0, 'batch size should not be odd!'
        feats = x.reshape(b, c, w * h)
        y, w_sci = HF.cin_sci(feats)                      # self-channel interaction (CIN.py:31-34)
        y = self._refine(y, (b, c, w, h))
        z = y + feats
        if not self.training:
            return z
        # contrastive channel interaction: sample i of the first half is paired with sample i of the second half
        flat, half = y.reshape(b, -1), b // 2
        first, second = flat[:half], flat[half:]
        eta = self.fc(torch.cat((first, second), dim=1))
        gamma = self.fc(torch.cat((second, first), dim=1))
        weight = torch.cat((eta, gamma), dim=0).reshape(b)
        y_cci = self._refine(HF.cin_cci(w_sci, feats, weight), (b, c, w, h))
        return z, y_cci + feats


class CINClassifier(nn.Module):
    def __init__(self, in_channel=2048, num_classes=200):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.classifier = nn.Linear(in_channel, num_classes)

    def _logits(self, z):
        return self.classifier(torch.squeeze(self.pool(z)))

    def forward(self, x):
        if isinstance(x, tuple):                          # training: classify Z, hand Z_CCI on to the criterion
            return self._logits(x[0]), x[1]
        return self._logits(x)


@MODEL.register
class CIN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_classes = config.num_classes if 'num_classes' in config else 200
        self.backbone = nn.Sequential(*list(resnet50(pretrained=True).children())[:-2])
        self.ChannelInteraction = ChannelInteractionModule(in_channel=2048, spatial_size=(7, 7))
        self.classifier = CINClassifier(in_channel=2048, num_classes=self.num_classes)
        self.ChannelInteraction.apply(initialize_weights)
        self.classifier.apply(initialize_weights)

    def forward(self, x):
        return self.classifier(self.ChannelInteraction(self.backbone(x)))
