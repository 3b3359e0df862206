"""Bilinear CNN plugin (mirrors model/methods/BCNN.py:8-55).

Same constructor contract (`config.stage` optional, `config.num_classes`), same
attributes (`backbone`, `bilinear_pooling`, `classifier`, `stage`) and the same 28
state_dict keys; the pooling itself runs on the gfx950 kernels
(hk_bcnn_pool_fwd/bwd) instead of bmm + 7 elementwise ATen ops."""
import torch.nn as nn

from ... import functional as HF
from ..backbone import vgg16
from ..backbone.vgg import ConvStack
from ..registry import MODEL
from ..utils import initialize_weights, pooled_classifier


FEATURE_CHANNELS = 512          # VGG-16 conv5_3 width: the bilinear descriptor has 512 x 512 entries


class BilinearPooling(nn.Module):
    """Parameter-free second-order pooling: [B,C,H,W] -> [B,C*C].

    y = normalize( sqrt( X X^T / HW + 1e-5 ) ) per image (X = the map flattened to C x HW).  One column-sum launch
    + one Gram launch whose epilogue writes the final y; backward = one GEMM-shaped launch + one rank-1 fix-up
    (hawkeye_amd/csrc/bcnn_fast.hip, bcnn_pool.hip)."""

    def __init__(self, signed_sqrt=False):
        """signed_sqrt=True selects the normalisation the reference keeps commented out (BCNN.py:23-24):
        sign(G) sqrt(|G| + 1e-10) instead of sqrt(G + 1e-5).  Default = what the reference runs."""
        super().__init__()
        self.signed_sqrt = signed_sqrt

    def forward(self, x):
        return HF.bilinear_pool(x, self.signed_sqrt)


def _frozen(module):
    for weight in module.parameters():
        weight.requires_grad = False
    return module


@MODEL.register
class BCNN(nn.Module):
    """VGG-16 conv trunk -> BilinearPooling -> Linear.  Stage 1 trains the classifier on a frozen trunk (the trunk's
    output is detached so no backward kernel is launched for it), stage 2 (default when `config.stage` is absent)
    fine-tunes everything."""

    def __init__(self, config):
        super().__init__()
        self.stage = config.stage if 'stage' in config else 2
        trunk = vgg16(pretrained=True).features            # the whole `features` stack, final MaxPool included:
        self.backbone = ConvStack(*trunk.children())        # 448x448 input -> [B,512,14,14]; an nn.Sequential (same keys) whose
                                                            # forward fuses the elementwise ops around each convolution
        self.bilinear_pooling = BilinearPooling()
        self.classifier = nn.Linear(FEATURE_CHANNELS * FEATURE_CHANNELS, config.num_classes)
        self.classifier.apply(initialize_weights)
        if self.stage == 1:
            _frozen(self.backbone)

    def forward(self, x):
        feats = self.backbone(x)
        if self.stage == 1:
            feats = feats.detach()
        # pooling + classifier as one autograd node (model/utils.py::pooled_classifier): the classifier's backward knows
        # <y, dy>, so the pooling's backward is one launch; `self.bilinear_pooling` stays the parameter-free module the
        # reference has (and what PeerLearningNet / user code may call on its own)
        return pooled_classifier(self, feats)
