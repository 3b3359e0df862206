"""Channel Interaction Network on the MI355X path.

Contract of the reference's model/methods/CIN.py (registered `CIN`, `ChannelInteractionModule`, `CINClassifier`;
same constructor arguments, attribute names and `state_dict` keys): the interaction itself - softmax(-X X^T / HW) X and,
in training, |W_A - eta W_B| X between the two halves of the batch - runs on the gfx950 kernels (hk_cin_sci_*,
hk_cin_cci_*); the 3x3 convolution, the residual sums, the one-output fc that produces eta / gamma and the classifier
are PyTorch-ROCm layers.
"""
import torch
import torch.nn as nn

from ... import functional as HF
from ..backbone import resnet50
from ..registry import MODEL
from ..utils import initialize_weights


def _swap_halves(t):
    """[first half; second half] -> [second half; first half] along dim 0 (the contrast partner of every sample)."""
    half = t.shape[0] // 2
    return torch.cat((t[half:], t[:half]), dim=0)


class ChannelInteractionModule(nn.Module):
    """forward(x [B,C,W,H]) -> Z [B,C,WH] in eval, (Z, Z_CCI) in training (B even)."""

    def __init__(self, in_channel=2048, spatial_size=(7, 7)):
        super().__init__()
        self.in_channel = in_channel
        self.spatial_size = spatial_size
        self.softmax = nn.Softmax()        # the reference carries this parameter-free member; nothing calls it
        self.conv = nn.Conv2d(in_channel, in_channel, kernel_size=3, stride=1, padding=1)
        self.fc = nn.Linear(2 * in_channel * spatial_size[0] * spatial_size[1], 1)

    def _conv_on_map(self, flat, map_shape):
        return self.conv(flat.view(map_shape)).view(flat.shape)

    def _pair_weights(self, y):
        """eta for the first half of the batch, gamma for the second: fc over [own features, partner's features]."""
        own = y.flatten(1)
        return self.fc(torch.cat((own, _swap_halves(own)), dim=1)).flatten()

    def forward(self, x):
        assert x.size(0) % 2 == 0, 'batch size should not be odd!'
        feats = x.flatten(2)
        y, w_sci = HF.cin_sci(feats)
        y = self._conv_on_map(y, x.shape)
        z = y + feats
        if not self.training:
            return z
        y_cci = HF.cin_cci(w_sci, feats, self._pair_weights(y))
        return z, self._conv_on_map(y_cci, x.shape) + feats


class CINClassifier(nn.Module):
    def __init__(self, in_channel=2048, num_classes=200):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.classifier = nn.Linear(in_channel, num_classes)

    def forward(self, x):
        z, z_cci = x if isinstance(x, tuple) else (x, None)
        logits = self.classifier(torch.squeeze(self.pool(z)))
        return logits if z_cci is None else (logits, z_cci)


@MODEL.register
class CIN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_classes = config.num_classes if 'num_classes' in config else 200
        trunk = resnet50(pretrained=True)
        self.backbone = nn.Sequential(*list(trunk.children())[:-2])
        self.ChannelInteraction = ChannelInteractionModule(in_channel=2048, spatial_size=(7, 7))
        self.classifier = CINClassifier(in_channel=2048, num_classes=self.num_classes)
        for head in (self.ChannelInteraction, self.classifier):
            head.apply(initialize_weights)

    def forward(self, x):
        return self.classifier(self.ChannelInteraction(self.backbone(x)))
