"""Compact bilinear CNN plugin (mirrors model/methods/CBCNN.py:12-164).

`CompactBilinearPooling` keeps the reference's constructor (seeds 1/3/5/7 hashes,
optional explicit hashes) and, like the reference, holds NO parameters or buffers
(the sketches are not in the state_dict).  It evaluates the tensor sketch through
the exact Gram identity on the HIP kernels (hk_cbp_fwd/bwd) - see csrc/cbp.hip."""
import numpy as np
import torch
import torch.nn as nn

from ... import functional as HF
from ..backbone import vgg16
from ..backbone.vgg import ConvStack
from ..registry import MODEL
from ..utils import initialize_weights, wide_linear


class CompactBilinearPooling(nn.Module):
    def __init__(self, input_dim1, input_dim2, output_dim, sum_pool=True,
                 rand_h_1=None, rand_s_1=None, rand_h_2=None, rand_s_2=None):
        super().__init__()
        self.input_dim1, self.input_dim2, self.output_dim, self.sum_pool = input_dim1, input_dim2, output_dim, sum_pool
        h1, s1, h2, s2 = HF.sketch_hashes(input_dim1, input_dim2, output_dim)
        self.rand_h_1 = np.asarray(rand_h_1 if rand_h_1 is not None else h1)
        self.rand_s_1 = np.asarray(rand_s_1 if rand_s_1 is not None else s1)
        self.rand_h_2 = np.asarray(rand_h_2 if rand_h_2 is not None else h2)
        self.rand_s_2 = np.asarray(rand_s_2 if rand_s_2 is not None else s2)
        self._plans = {}          # device -> CbpPlan (lazy, like the reference's lazy .to(device), CBCNN.py:107-110)

    def _plan(self, device):
        key = (device.type, device.index)
        if key not in self._plans:
            # two widths (CBCNN.py:68-94 sizes each sketch matrix by its own input_dim): the plan over the C1 x C2 cross Gram
            make = HF.CbpPlan if self.input_dim1 == self.input_dim2 else HF.CbpRectPlan
            self._plans[key] = make(self.rand_h_1, self.rand_s_1, self.rand_h_2, self.rand_s_2, self.output_dim, device)
        return self._plans[key]

    def __deepcopy__(self, memo):          # plans hold device blobs: rebuild lazily in the copy
        return CompactBilinearPooling(self.input_dim1, self.input_dim2, self.output_dim, self.sum_pool,
                                      self.rand_h_1.copy(), self.rand_s_1.copy(), self.rand_h_2.copy(),
                                      self.rand_s_2.copy())

    def forward(self, bottom1, bottom2=None):
        one_input = bottom2 is None or bottom2 is bottom1
        if bottom2 is None:
            bottom2 = bottom1                                                                # CBCNN.py:101-102 (a clone there)
        assert bottom1.size(1) == self.input_dim1 and bottom2.size(1) == self.input_dim2     # CBCNN.py:104-105
        plan = self._plan(bottom1.device)
        if one_input and self.sum_pool:          # what Hawkeye's CBCNN calls (CBCNN.py:23,33): Gram + binning + finish on the kernels
            return HF.compact_bilinear_pool(bottom1, plan)
        # two different inputs (cross Gram) or sum_pool = False (every location on its own): the sketch on the kernels, then
        # the reference's own two lines (CBCNN.py:132-133; for a [B,H,W,D] tensor F.normalize runs along H - kept as it is)
        cbp = HF.compact_bilinear_sketch(bottom1, bottom2, plan, self.sum_pool)
        cbp = torch.sign(cbp) * torch.sqrt(torch.abs(cbp) + 1e-10)
        return torch.nn.functional.normalize(cbp)


@MODEL.register
class CBCNN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        cin, cout = config.input_channel, config.output_channel
        self.backbone = ConvStack(*list(vgg16(pretrained=True).features.children()))    # an nn.Sequential: same keys
        self.bilinear_pooling = CompactBilinearPooling(cin, cin, cout)
        self.classifier = nn.Linear(cout, config.num_classes)
        self.classifier.apply(initialize_weights)

    def forward(self, x):
        x = self.backbone(x)
        if self.config.stage == 1:
            x = x.detach()
        return wide_linear(self.classifier, self.bilinear_pooling(x))
