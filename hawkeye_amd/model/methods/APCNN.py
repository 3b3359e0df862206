"""AP-CNN plugin (mirrors model/methods/APCNN.py:345-625).

Same module tree as the reference (child order conv1, bn1, relu, maxpool, layer1-4,
fpn, apn, cls5, cls4, cls3, cls_concate - Examples/APCNN.py:38-42 splits optimiser
groups on it - and the same 414 state_dict keys).  Convolutions / BatchNorm / the
small MLP heads stay on PyTorch-ROCm; the attention-pooling hot path runs on the
gfx950 kernels:

  * A = a_s*F + a_c*F is never materialised: hk_att_pool gives GAP(F) and GAP(a_s*F)
    in one pass, pooled = sgap + a_c*gap feeds cls3/4/5 directly, and the same GAP
    feeds ChannelGate and the concat head (the reference reads F five times per level);
  * get_att_roi's per-image python loop + NMS `while` (one host sync per kept box)
    is one kernel launch per level with no host sync;
  * get_roi_crop_feat's per-image loop is two launches (boxes, crop/drop/resize).

Host-side randomness: the reference draws `random.random()` and
`random.randint(0, n-1)` per image (APCNN.py:494-501) after reading ROI counts
back from the device.  By default this module draws two python `random.random()`
per image and derives the index on the device as floor(u*n) (same distribution,
no device->host sync).  `exact_random_stream=True` reproduces the reference's
draw sequence bit for bit at the price of one host sync per forward.
"""
import math
import random

import torch
import torch.nn as nn
import torch.nn.functional as TF

from ... import functional as HF
from ..backbone.resnet import Bottleneck, make_stage
from ..backbone import pretrained as _pre
from ..registry import MODEL
from ..utils import load_state_dict


class BasicConv(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, relu=True, bn=True, bias=False):
        super().__init__()
        self.out_channels = cout
        self.conv = nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5, momentum=0.01, affine=True) if bn else None
        self.relu = nn.ReLU(inplace=True) if relu else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.relu(x) if self.relu is not None else x


class SimpleFPA(nn.Module):
    """master 1x1 branch + global-pool 1x1 branch (APCNN.py:172-199)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.channels_cond = cin
        self.conv_master = BasicConv(cin, cout, kernel_size=1, stride=1)
        self.conv_gpb = BasicConv(cin, cout, kernel_size=1, stride=1)

    def forward(self, x):
        gap, _ = HF.att_pool(x, None)
        return self.conv_master(x) + self.conv_gpb(gap.view(x.shape[0], self.channels_cond, 1, 1))


class PyramidFeatures(nn.Module):
    """Top-down FPN (APCNN.py:202-233)."""

    def __init__(self, b2, b3, b4, b5, feature_size=256):
        super().__init__()
        self.P5_1 = SimpleFPA(b5, feature_size)
        self.P5_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P4_1 = nn.Conv2d(b4, feature_size, kernel_size=1, stride=1, padding=0)
        self.P4_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P3_1 = nn.Conv2d(b3, feature_size, kernel_size=1, stride=1, padding=0)
        self.P3_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)

    def forward(self, inputs):
        b3, b4, b5 = inputs
        p5 = self.P5_1(b5)
        p5_up = TF.interpolate(p5, scale_factor=2)
        p5 = self.P5_2(p5)
        p4 = p5_up + self.P4_1(b4)
        p4_up = TF.interpolate(p4, scale_factor=2)
        p4 = self.P4_2(p4)
        p3 = self.P3_2(self.P3_1(b3) + p4_up)
        return [p3, p4, p5]


class SpatialGate(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.ConvTranspose2d(channels, 1, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return torch.sigmoid(self.conv(x))


class ChannelGate(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels // 16, kernel_size=1, stride=1, padding=0)
        self.conv2 = nn.Conv2d(channels // 16, channels, kernel_size=1, stride=1, padding=0)

    def from_gap(self, gap):
        """gap [B,C] (already pooled by hk_att_pool) -> gate [B,C]"""
        z = TF.relu(self.conv1(gap.view(gap.shape[0], -1, 1, 1)), inplace=True)
        return torch.sigmoid(self.conv2(z)).flatten(1)

    def forward(self, x):
        gap, _ = HF.att_pool(x, None)
        return self.from_gap(gap).view(x.shape[0], -1, 1, 1)


class PyramidAttentions(nn.Module):
    """Bottom-up attention pyramid (APCNN.py:236-268), in pooled form:
    returns pooled_l = GAP(a_s*F + a_c*F) = sgap_l + a_c,l * gap_l, the plain GAPs and the masks."""

    def __init__(self, channel_size=256):
        super().__init__()
        self.A3_1, self.A3_2 = SpatialGate(channel_size), ChannelGate(channel_size)
        self.A4_1, self.A4_2 = SpatialGate(channel_size), ChannelGate(channel_size)
        self.A5_1, self.A5_2 = SpatialGate(channel_size), ChannelGate(channel_size)

    def forward(self, inputs):
        pooled, gaps, ch_prev = [], [], None
        masks = [sg(f) for f, sg in zip(inputs, (self.A3_1, self.A4_1, self.A5_1))]       # ConvTranspose + sigmoid (MIOpen)
        # one pass over every F, the three levels in one launch (only the channel gates below chain across levels)
        gap3, sgap3 = HF.att_pool_levels(inputs, masks)
        for lvl, cg in enumerate((self.A3_2, self.A4_2, self.A5_2)):
            gap, sgap = gap3[lvl], sgap3[lvl]
            a_c = cg.from_gap(gap)
            if ch_prev is not None:
                a_c = (a_c + ch_prev) / 2                     # APCNN.py:260,265
            ch_prev = a_c
            pooled.append(sgap + a_c * gap)
            gaps.append(gap)
        return pooled, gaps, masks


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class LazyRois:
    """The reference returns `[roi_3, roi_4, roi_5]`, each [sum_i k_i, 6] with rows
    [img, x1, y1, x2, y2, score] (APCNN.py:594-599).  Building that ragged layout needs
    the per-image counts on the host, so it is materialised only when somebody looks."""

    def __init__(self, tables):
        self.tables = tables
        self._rows = None

    def _materialise(self):
        if self._rows is None:
            self._rows = [compact_rois(r, n) for r, n in self.tables]
        return self._rows

    def __len__(self):
        return len(self.tables)

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


def compact_rois(rois, cnt):
    """fixed [B,k,5] + counts -> reference layout [sum k_i, 6] (one host sync)."""
    b, k, _ = rois.shape
    valid = torch.arange(k, device=rois.device).unsqueeze(0) < cnt.unsqueeze(1)
    img = torch.arange(b, device=rois.device, dtype=torch.float32).view(b, 1, 1).expand(b, k, 1)
    return torch.cat([img, rois], dim=2)[valid]


def _cls_head(hidden, num_classes, gap_first=True, width=256):
    mods = [nn.AdaptiveAvgPool2d(1)] if gap_first else []
    mods += [Flatten(), nn.BatchNorm1d(width), nn.Linear(width, hidden), nn.BatchNorm1d(hidden),
             nn.ELU(inplace=True), nn.Linear(hidden, num_classes)]
    return nn.Sequential(*mods)


class ResNet(nn.Module):
    """AP-CNN on a ResNet trunk (APCNN.py:345-599)."""

    LEVELS = ((2 ** 3, 64, 5), (2 ** 4, 128, 3), (2 ** 5, 256, 1))   # stride, anchor, top-k  (:567-569)

    def __init__(self, num_classes, block, layers):
        super().__init__()
        self.num_classes = num_classes
        self.exact_random_stream = False
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        c = 64
        self.layer1, c1 = make_stage(block, c, 64, layers[0], 1)
        self.layer2, c2 = make_stage(block, c1, 128, layers[1], 2)
        self.layer3, c3 = make_stage(block, c2, 256, layers[2], 2)
        self.layer4, c4 = make_stage(block, c3, 512, layers[3], 2)
        hidden = 512 if num_classes == 200 else 256                      # :360-363
        self.fpn = PyramidFeatures(c1, c2, c3, c4)
        self.apn = PyramidAttentions(channel_size=256)
        self.cls5 = _cls_head(hidden, num_classes)
        self.cls4 = _cls_head(hidden, num_classes)
        self.cls3 = _cls_head(hidden, num_classes)
        self.cls_concate = _cls_head(hidden, num_classes, gap_first=False, width=256 * 3)
        for m in self.modules():                                         # :418-424
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    # -- stage head: FPN features -> 4 logits + attention masks ---------------------------------
    def _heads(self, feats):
        pooled, gaps, masks = self.apn(feats)
        out_concate = self.cls_concate(torch.cat(gaps, dim=1))           # Concate (:533-538) + cls_concate
        out3 = self.cls3[2:](pooled[0])                                  # [0:2] = AdaptiveAvgPool2d + Flatten: fused
        out4 = self.cls4[2:](pooled[1])
        out5 = self.cls5[2:](pooled[2])
        return out3, out4, out5, out_concate, masks

    def get_att_roi(self, att_mask, feature_stride, anchor_size, img_h, img_w, iou_thred=0.2, topk=1):
        """Same signature as APCNN.py:444; returns the reference's ragged layout (syncs)."""
        return compact_rois(*HF.att_roi_select(att_mask, feature_stride, anchor_size, img_h, img_w,
                                               self.num_classes, iou_thred, topk))

    def _drop_uniforms(self, n, device):
        u = torch.tensor([[random.random(), random.random()] for _ in range(n)], dtype=torch.float32)
        if torch.device(device).type == 'cuda':
            u = u.pin_memory()                       # async H2D: no host sync on the step
        return u.to(device, non_blocking=True)

    def _boxes_exact_stream(self, tables, scale):
        """Reference draw sequence (random.random / random.randint after reading the counts): one host sync."""
        (r3, n3), (r4, n4), (r5, n5) = tables
        box, _ = HF.roi_boxes(tables, None, scale)
        c3, c4 = n3.cpu().tolist(), n4.cpu().tolist()
        drop = torch.tensor([[0., 0., -1., -1.]] * len(c3))
        pick = []
        for i in range(len(c3)):
            pr = random.random()
            if pr < 0.3:
                pick.append((i, 0, random.randint(0, c3[i] - 1)))
            elif pr < 0.6:
                pick.append((i, 1, random.randint(0, c4[i] - 1)))
        drop = drop.to(r3.device)
        for i, lvl, idx in pick:
            drop[i] = (r3 if lvl == 0 else r4)[i, idx, :4] / scale
        return box, drop

    def get_roi_crop_feat(self, x, tables, scale, drops=None):
        """ROI guided zoom-in + drop block on device.  `tables` = three (rois[B,k,5], count[B]);
        `drops` (tests): explicit [B,4] drop boxes overriding the random draws."""
        if isinstance(tables, LazyRois):
            tables = tables.tables
        if drops is not None:
            box, _ = HF.roi_boxes(tables, None, scale)
            drop = drops
        elif not self.training:
            box, drop = HF.roi_boxes(tables, None, scale)
        elif self.exact_random_stream:
            box, drop = self._boxes_exact_stream(tables, scale)
        else:
            box, drop = HF.roi_boxes(tables, self._drop_uniforms(x.shape[0], x.device), scale)
        return HF.roi_crop_resize(x, box, drop, self.training), box

    def forward(self, inputs, targets=None):
        n, _, img_h, img_w = inputs.size()
        x = self.maxpool(self.relu(self.bn1(self.conv1(inputs))))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)

        # stage I
        out3, out4, out5, out_concate, (a3, a4, a5) = self._heads(self.fpn([x2, x3, x4]))

        # ROI pyramid (no gradient, no host sync)
        tables = HF.att_roi_select_levels((a3, a4, a5), self.LEVELS, img_h, img_w, self.num_classes, 0.05)   # one launch

        # stage II on the refined layer-2 map
        x2r, _ = self.get_roi_crop_feat(x2, tables, 2 ** 3)
        x3r = self.layer3(x2r)
        x4r = self.layer4(x3r)
        out3r, out4r, out5r, out_concate_r, _ = self._heads(self.fpn([x2r, x3r, x4r]))

        mask_cat = torch.cat([a3, TF.interpolate(a4, a3.size()[2:]), TF.interpolate(a5, a3.size()[2:])], 1)
        out_list = [out3, out4, out5, out_concate, out3r, out4r, out5r, out_concate_r]
        out_mean = sum(out_list) / len(out_list)
        return out_mean, out_list, mask_cat, LazyRois(tables)


def resnet50(num_classes, **kwargs):
    return ResNet(num_classes, Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(num_classes, **kwargs):
    return ResNet(num_classes, Bottleneck, [3, 4, 23, 3], **kwargs)


@MODEL.register
def APCNN(config):
    model = resnet50(config.num_classes)
    sd = _pre.load('resnet50')
    if sd is not None:
        load_state_dict(model, sd)                                       # lenient (:623-624)
    return model
