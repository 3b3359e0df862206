"""CPU oracle for the Hawkeye high-order pooling / attention-pooling hot path.

TEST INFRASTRUCTURE ONLY.  This file restates, in plain PyTorch-CPU fp32, the
algorithms of the reference's hot-path functions so that the HIP kernels can be
checked against them on a GPU box where ``/root/reference`` does not exist.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``hawkeye_amd/`` does.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the pin is the reference itself: ``oracle/gen_golden.py``
imports ``/root/reference`` in the build container, runs the reference modules on
seeded inputs and stores inputs/outputs/input-gradients under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function below against those
fixtures (bit-exact or <=1e-6, see the test).

Each function cites the reference lines it follows.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# BCNN bilinear pooling  (model/methods/BCNN.py:13-27)
# ----------------------------------------------------------------------------
def bilinear_pool(x):
    """Gram / HW -> sqrt(. + 1e-5) -> l2 normalise.  BCNN.py:13-27."""
    b, c, h, w = x.shape
    m = h * w
    xm = x.reshape(b, c, m)
    g = torch.bmm(xm, xm.transpose(1, 2)) / m          # BCNN.py:18
    z = torch.sqrt(g.reshape(b, -1) + 1e-5)             # BCNN.py:20-21
    return F.normalize(z)                                # BCNN.py:26 (p=2, dim=1, eps=1e-12)


def bilinear_pool_signed_sqrt(x):
    """BilinearPooling.forward with the normalisation the reference keeps commented out (BCNN.py:23-24) in place of
    the `sqrt(x + 1e-5)` of :21 - the same torch ops, just the other line enabled."""
    b, c, h, w = x.shape
    m = h * w
    xm = x.reshape(b, c, m)                                      # :17
    g = torch.bmm(xm, xm.transpose(1, 2)) / m                    # :18
    g = g.reshape(b, -1)                                         # :23
    u = torch.sign(g) * torch.sqrt(torch.abs(g) + 1e-10)         # :24
    return F.normalize(u)                                        # :26


# ----------------------------------------------------------------------------
# Compact bilinear pooling  (model/methods/CBCNN.py:68-164)
# ----------------------------------------------------------------------------
def sketch_hashes(input_dim1, input_dim2, output_dim):
    """Count-sketch hashes with the reference's fixed legacy numpy seeds.
    CBCNN.py:76-91 (seeds 1/3 for h1/s1, 5/7 for h2/s2)."""
    np.random.seed(1)
    h1 = np.random.randint(output_dim, size=input_dim1)
    np.random.seed(3)
    s1 = 2 * np.random.randint(2, size=input_dim1) - 1
    np.random.seed(5)
    h2 = np.random.randint(output_dim, size=input_dim2)
    np.random.seed(7)
    s2 = 2 * np.random.randint(2, size=input_dim2) - 1
    return (h1.astype(np.int64), s1.astype(np.float32),
            h2.astype(np.int64), s2.astype(np.float32))


def _dense_sketch(h, s, d):
    """[C, D] matrix with one +-1 per row.  CBCNN.py:137-164."""
    mat = torch.zeros(len(h), d, dtype=torch.float32)
    mat[torch.arange(len(h)), torch.from_numpy(h)] = torch.from_numpy(s)
    return mat


def compact_bilinear_pool(x, output_dim, hashes=None):
    """Tensor-sketch via FFT, sum-pool, signed sqrt, l2 normalise.
    CBCNN.py:96-135 (literal FFT route: this is what the reference executes)."""
    b, c, h, w = x.shape
    if hashes is None:
        hashes = sketch_hashes(c, c, output_dim)
    h1, s1, h2, s2 = hashes
    m1 = _dense_sketch(h1, s1, output_dim).to(x.dtype)
    m2 = _dense_sketch(h2, s2, output_dim).to(x.dtype)
    flat = x.permute(0, 2, 3, 1).contiguous().view(-1, c)        # :114-115
    sk1 = flat.mm(m1)                                            # :117
    sk2 = flat.clone().mm(m2)                                    # :118
    prod = torch.fft.fft(sk1) * torch.fft.fft(sk2)               # :120-123
    cbp = torch.fft.ifft(prod).real.view(b, h, w, output_dim)    # :125-127
    cbp = cbp.sum(dim=1).sum(dim=1)                              # :129-130
    cbp = torch.sign(cbp) * torch.sqrt(torch.abs(cbp) + 1e-10)   # :132
    return F.normalize(cbp)                                      # :133


def compact_bilinear_pool_gram(x, output_dim, hashes=None):
    """The same function through the count-sketch identity
        c[b,k] = sum_{(i,j): (h1[i]+h2[j]) mod D = k} s1[i] s2[j] (X X^T)[b,i,j]
    (circular convolution of two count sketches = count sketch of the outer product;
    the sum over positions commutes).  Not what the reference executes - it is the
    algebraic form the HIP path uses - kept here so that tests can (a) verify the
    identity against the FFT route on the CPU and (b) pin autograd semantics at bins
    that are exactly 0 (torch: sign'(c) = 0, abs'(0) = 0 -> zero gradient), where the
    FFT route differentiates its own round-off noise."""
    b, c, h, w = x.shape
    if hashes is None:
        hashes = sketch_hashes(c, c, output_dim)
    h1, s1, h2, s2 = hashes
    xm = x.reshape(b, c, h * w)
    gram = torch.bmm(xm, xm.transpose(1, 2))
    bins = ((torch.from_numpy(h1)[:, None] + torch.from_numpy(h2)[None, :]) % output_dim).reshape(-1)
    sign = (torch.from_numpy(s1)[:, None] * torch.from_numpy(s2)[None, :]).to(x.dtype).reshape(-1)
    cbp = torch.zeros(b, output_dim, dtype=x.dtype).index_add(1, bins, gram.reshape(b, -1) * sign)
    cbp = torch.sign(cbp) * torch.sqrt(torch.abs(cbp) + 1e-10)
    return F.normalize(cbp)


# ----------------------------------------------------------------------------
# Fast MPN-COV  (model/methods/MPNCOV.py:105-230)
# ----------------------------------------------------------------------------
class _Covpool(torch.autograd.Function):
    """MPNCOV.py:105-134."""

    @staticmethod
    def forward(ctx, x):
        b, c, h, w = x.shape
        m = h * w
        xm = x.reshape(b, c, m)
        # :115-116  I_hat = (1/M)(I - 11^T/M), built in fp32 then cast
        i_hat = (-1.0 / m / m) * torch.ones(m, m) + (1.0 / m) * torch.eye(m, m)
        i_hat = i_hat.view(1, m, m).repeat(b, 1, 1).type(x.dtype)
        y = xm.bmm(i_hat).bmm(xm.transpose(1, 2))                # :117
        ctx.save_for_backward(x, i_hat)
        return y

    @staticmethod
    def backward(ctx, g):
        x, i_hat = ctx.saved_tensors
        b, c, h, w = x.shape
        xm = x.reshape(b, c, h * w)
        gi = g + g.transpose(1, 2)                               # :131
        gi = gi.bmm(xm).bmm(i_hat)                               # :132
        return gi.reshape(b, c, h, w)


class _Sqrtm(torch.autograd.Function):
    """Newton-Schulz coupled iteration and its hand-derived backward.
    MPNCOV.py:137-202.  Y/Z iterates are kept in python lists instead of the
    reference's [B, iterN, d, d] slabs; the arithmetic is the same."""

    @staticmethod
    def forward(ctx, x, iter_n):
        b, d, _ = x.shape
        eye3 = 3.0 * torch.eye(d, dtype=x.dtype).expand(b, d, d)
        norm_a = (1.0 / 3.0) * x.mul(eye3).sum(dim=1).sum(dim=1)    # :145 (trace)
        a = x.div(norm_a.view(b, 1, 1))                              # :146
        ys, zs = [], []
        zy = 0.5 * (eye3 - a)                                        # :150/:153
        if iter_n < 2:
            yzy = a.bmm(zy)                                          # :151
        else:
            ys.append(a.bmm(zy))                                     # :154
            zs.append(zy)                                            # :155
            for i in range(1, iter_n - 1):                           # :156-159
                zy = 0.5 * (eye3 - zs[i - 1].bmm(ys[i - 1]))
                ys.append(ys[i - 1].bmm(zy))
                zs.append(zy.bmm(zs[i - 1]))
            yzy = 0.5 * ys[-1].bmm(eye3 - zs[-1].bmm(ys[-1]))        # :160
        y = yzy * torch.sqrt(norm_a).view(b, 1, 1)                   # :161
        ctx.iter_n = iter_n
        ctx.ys, ctx.zs = ys, zs
        ctx.save_for_backward(x, a, yzy, norm_a)
        return y

    @staticmethod
    def backward(ctx, g):
        x, a, yzy, norm_a = ctx.saved_tensors
        ys, zs, iter_n = ctx.ys, ctx.zs, ctx.iter_n
        b, d, _ = x.shape
        eye3 = 3.0 * torch.eye(d, dtype=x.dtype).expand(b, d, d)
        sq = torch.sqrt(norm_a)
        dpc = g * sq.view(b, 1, 1)                                   # :174
        aux = (g * yzy).sum(dim=1).sum(dim=1).div(2 * sq)            # :175
        if iter_n < 2:
            der = 0.5 * (dpc.bmm(eye3 - a) - a.bmm(dpc))             # :178
        else:
            yl, zl = ys[-1], zs[-1]
            dldy = 0.5 * (dpc.bmm(eye3 - yl.bmm(zl)) - zl.bmm(yl).bmm(dpc))   # :180-181
            dldz = -0.5 * yl.bmm(dpc).bmm(yl)                                   # :182
            for i in range(iter_n - 3, -1, -1):                                 # :183-193
                yz = eye3 - ys[i].bmm(zs[i])
                zy = zs[i].bmm(ys[i])
                dldy_ = 0.5 * (dldy.bmm(yz) - zs[i].bmm(dldz).bmm(zs[i]) - zy.bmm(dldy))
                dldz_ = 0.5 * (yz.bmm(dldz) - ys[i].bmm(dldy).bmm(ys[i]) - dldz.bmm(zy))
                dldy, dldz = dldy_, dldz_
            der = 0.5 * (dldy.bmm(eye3 - a) - dldz - a.bmm(dldy))    # :194
        der = der.transpose(1, 2)                                    # :195
        grad = der.div(norm_a.view(b, 1, 1))                         # :196
        grad_aux = der.mul(x).sum(dim=1).sum(dim=1)                  # :197
        coef = aux - grad_aux / (norm_a * norm_a)                    # :198-201
        grad = grad + coef.view(b, 1, 1) * torch.eye(d, dtype=x.dtype)
        return grad, None


def triu_index(d):
    """Row-major positions (r, c), c >= r.  MPNCOV.py:213-214."""
    return torch.ones(d, d).triu().reshape(-1).nonzero()


class _Triuvec(torch.autograd.Function):
    """MPNCOV.py:205-230; note the [B, d(d+1)/2, 1] output shape (:216)."""

    @staticmethod
    def forward(ctx, x):
        b, d, _ = x.shape
        idx = triu_index(d)
        ctx.save_for_backward(idx)
        ctx.dims = (b, d)
        return x.reshape(b, d * d)[:, idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        b, d = ctx.dims
        out = torch.zeros(b, d * d, dtype=g.dtype)
        out[:, idx] = g
        return out.reshape(b, d, d)


def covpool(x):
    return _Covpool.apply(x)


def sqrtm(x, iter_n):
    return _Sqrtm.apply(x, iter_n)


def triuvec(x):
    return _Triuvec.apply(x)


def mpncov_pool(x, iter_n=5, is_sqrt=True, is_vec=True):
    """MPNCOV.forward after the 1x1 DR block.  MPNCOV.py:94-102."""
    x = covpool(x)
    if is_sqrt:
        x = sqrtm(x, iter_n)
    if is_vec:
        x = triuvec(x)
    return x


# ----------------------------------------------------------------------------
# AP-CNN attention pooling / ROI selection / ROI refinement (APCNN.py)
# ----------------------------------------------------------------------------
def pyramid_attention(feats, spatial, channel):
    """A_l = a_s*F + a_c*F with bottom-up averaged channel gates.
    APCNN.py:251-268.  `spatial[l]` = sigmoid(ConvT(F_l)) [B,1,h,w],
    `channel[l]` = raw ChannelGate outputs [B,C,1,1]."""
    out, ch_prev = [], None
    for lvl, (f, a_s, a_c) in enumerate(zip(feats, spatial, channel)):
        if lvl > 0:
            a_c = (a_c + ch_prev) / 2                            # :260,:265
        ch_prev = a_c
        out.append(a_s * f + a_c * f)                            # :256,:261,:266
    return out


def attention_pooled(feats, spatial, channel):
    """GAP of the attended maps = first layer of cls3/4/5.  APCNN.py:377-405,561-563."""
    return [F.adaptive_avg_pool2d(a, 1).flatten(1)
            for a in pyramid_attention(feats, spatial, channel)]


def square_anchors(h, w, feature_stride, anchor_size):
    """One square anchor per cell (scales=[size], ratios=[1], anchor_stride=1).
    APCNN.py:306-341 specialised; float64 like numpy, cast by the caller."""
    ys = np.arange(0, h) * feature_stride
    xs = np.arange(0, w) * feature_stride
    cx, cy = np.meshgrid(xs, ys)
    cx = cx.reshape(-1).astype(np.float64)
    cy = cy.reshape(-1).astype(np.float64)
    half = 0.5 * float(anchor_size)
    return torch.from_numpy(np.stack([cx - half, cy - half, cx + half, cy + half], axis=1))


def nms_greedy(p, thresh_iou):
    """Greedy NMS, highest score first, area without +1, strict IoU < thr keeps.
    model/methods/nms.py:4-93.  Ties: the reference uses a non-stable argsort
    (nms.py:30); this oracle fixes the rule "highest index first among equal
    scores" (= stable ascending sort, take from the end)."""
    x1, y1, x2, y2, sc = p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4]
    areas = (x2 - x1) * (y2 - y1)
    order = torch.argsort(sc, stable=True)
    keep = []
    while len(order) > 0:
        idx = order[-1]
        keep.append(p[idx])
        order = order[:-1]
        if len(order) == 0:
            break
        xx1 = torch.max(x1[order], x1[idx])
        yy1 = torch.max(y1[order], y1[idx])
        xx2 = torch.min(x2[order], x2[idx])
        yy2 = torch.min(y2[order], y2[idx])
        iw = torch.clamp(xx2 - xx1, min=0.0)
        ih = torch.clamp(yy2 - yy1, min=0.0)
        inter = iw * ih
        union = (areas[order] - inter) + areas[idx]
        order = order[(inter / union) < thresh_iou]
    return torch.stack(keep, dim=0)


def att_roi(att_mask, feature_stride, anchor_size, img_h, img_w, num_classes,
            iou_thred, topk):
    """APCNN.py:444-476.  Returns [sum_i k_i, 6] rows [img, x1, y1, x2, y2, score]."""
    with torch.no_grad():
        n, _, h, w = att_mask.shape
        keep = torch.zeros_like(att_mask)
        lo, hi = (0.2, 0.8) if num_classes == 200 else (0.1, 0.9)       # :451-455
        keep[:, :, int(lo * h):int(hi * h), int(lo * w):int(hi * w)] = 1
        masked = att_mask * keep
        anchors = square_anchors(h, w, feature_stride, anchor_size)
        rows = []
        for i in range(n):
            boxes = anchors.clone().float()
            scores = masked[i].reshape(-1)
            sel = scores > scores.mean()                                  # :461
            cand = torch.cat([boxes[sel], scores[sel].unsqueeze(1)], dim=1)
            kept = nms_greedy(cand, iou_thred)[:topk]                     # :465
            kept[:, 0] = torch.clamp(kept[:, 0], min=0)                   # :469-472
            kept[:, 1] = torch.clamp(kept[:, 1], min=0)
            kept[:, 2] = torch.clamp(kept[:, 2], max=img_w - 1)
            kept[:, 3] = torch.clamp(kept[:, 3], max=img_h - 1)
            rows.append(torch.cat([torch.full((kept.size(0), 1), float(i)), kept], 1))
        return torch.cat(rows, 0)


def roi_crop_feat(x, roi_list, scale, training, drops=None):
    """ROI-guided zoom-in (+ drop block when training).  APCNN.py:478-531.

    `drops`: optional per-image list of (level, index) / None overriding the
    reference's python `random` draws (:494-504) so both sides of a parity test
    see the same decisions; when None the reference's draw sequence is used."""
    n, c, hh, ww = x.shape
    roi_3, roi_4, roi_5 = roi_list
    roi_all = torch.cat([roi_3, roi_4, roi_5], 0)
    outs = []
    for i in range(n):
        r = roi_all[roi_all[:, 0] == i] / scale
        xx1, yy1 = torch.min(r[:, 1:3], 0)[0]
        xx2, yy2 = torch.max(r[:, 3:5], 0)[0]
        if training:
            r3 = roi_3[roi_3[:, 0] == i] / scale
            r4 = roi_4[roi_4[:, 0] == i] / scale
            mask = torch.ones(c, hh, ww, dtype=x.dtype)
            if drops is None:
                pr = random.random()
                if pr < 0.3:
                    pick = (3, random.randint(0, r3.size(0) - 1))
                elif pr < 0.6:
                    pick = (4, random.randint(0, r4.size(0) - 1))
                else:
                    pick = None
            else:
                pick = drops[i]
            if pick is not None:
                rr = (r3 if pick[0] == 3 else r4)[pick[1]]
                mask[:, rr[2].long():rr[4].long(), rr[1].long():rr[3].long()] = 0
            xd = x[i] * mask
            crop = xd[:, yy1.long():yy2.long(), xx1.long():xx2.long()].contiguous().unsqueeze(0)
            rate = c * (yy2 - yy1) * (xx2 - xx1) / torch.sum(
                mask[:, yy1.long():yy2.long(), xx1.long():xx2.long()])     # :509-511
            crop = crop * rate
        else:
            crop = x[i, :, yy1.long():yy2.long(), xx1.long():xx2.long()].contiguous().unsqueeze(0)
        outs.append(F.interpolate(crop, (hh, ww), mode='bilinear', align_corners=False))
    return torch.cat(outs, 0)


# ----------------------------------------------------------------------------
# OSME  (model/methods/OSME.py:8-44)
# ----------------------------------------------------------------------------
def osme_gate(x, w1, b1, w2, b2):
    """One OSME_block: GAP -> FC-ReLU-FC-sigmoid -> channel scale.  OSME.py:19-24."""
    n, c = x.shape[:2]
    z = F.adaptive_avg_pool2d(x, 1).reshape(n, c)
    m = torch.sigmoid(F.linear(F.relu(F.linear(z, w1, b1)), w2, b2))
    return m.view(n, c, 1, 1) * x


def osme_forward(x, gates, fcs):
    """OSME.forward: per attention gate + FC on the flattened map.  OSME.py:36-44.
    gates: list of (w1,b1,w2,b2); fcs: list of (W,b)."""
    n = x.shape[0]
    feats = [F.linear(osme_gate(x, *g).reshape(n, -1), w, b) for g, (w, b) in zip(gates, fcs)]
    return sum(feats), torch.stack(feats, dim=1)


# ----------------------------------------------------------------------------
# CIN channel interaction  (model/methods/CIN.py:24-60, model/loss/CIN_loss.py:27-50) - SURVEY 8f-2
# ----------------------------------------------------------------------------
def channel_interaction(x, conv_w, conv_b, fc_w, fc_b, training):
    """ChannelInteractionModule.forward, CIN.py:24-60.  x [B,C,W,H]; conv 3x3 C->C; fc: 2*C*WH -> 1.
    eval: Z [B,C,WH]; train: (Z, Z_CCI) with the two batch halves as contrast partners."""
    b, c, w, h = x.shape
    assert b % 2 == 0, 'batch size should not be odd!'                       # :27
    xm = x.view(b, c, w * h)                                                 # :28
    gram = torch.bmm(xm, xm.transpose(1, 2)) / (w * h)                       # :31
    w_sci = F.softmax(-gram, dim=2)                                          # :32
    y = torch.bmm(w_sci, xm)                                                 # :34
    y = F.conv2d(y.view(b, c, w, h), conv_w, conv_b, 1, 1).view(b, c, w * h) # :36-37
    z = y + xm                                                               # :38
    if not training:
        return z                                                             # :40-41
    yf = y.reshape(b, -1)                                                    # :44
    half = b // 2
    y_a = torch.cat((yf[:half], yf[half:]), dim=1)                           # :45
    y_b = torch.cat((yf[half:], yf[:half]), dim=1)                           # :46
    weight = torch.cat((F.linear(y_a, fc_w, fc_b), F.linear(y_b, fc_w, fc_b)), dim=0)   # :47-50
    w_ba = torch.cat((w_sci[half:], w_sci[:half]), dim=0)                    # :51
    w_cci = torch.abs(w_sci - weight.view(-1, 1, 1) * w_ba)                  # :52
    y_cci = torch.bmm(w_cci, xm)                                             # :54
    y_cci = F.conv2d(y_cci.view(b, c, w, h), conv_w, conv_b, 1, 1).view(b, c, w * h)    # :56-57
    return z, y_cci + xm                                                     # :58-60


def cin_loss(output, target, h_w, h_b, alpha=2.0, beta=0.5):
    """CINLoss.__call__, CIN_loss.py:27-50, restated literally - including that `pair_label` compares the first half
    of the labels with the single label target[B//2] (:40) and that `loss_cont_2` is overwritten by
    `loss_cont_1 ** 2` (:44), so the margin term never contributes."""
    if not isinstance(output, tuple):
        return F.cross_entropy(output, target, label_smoothing=0.1)          # :28-29
    z, z_cci = output
    b = z_cci.shape[0]
    loss_ce = F.cross_entropy(z, target, label_smoothing=0.1)                # :35
    z_ab = F.linear(z_cci.reshape(b, -1), h_w, h_b)                          # :38-39
    half = b // 2
    pair = target[:half] == target[half]                                     # :40
    d_pos = F.pairwise_distance(z_ab[:half][pair], z_ab[half:][pair], p=2)
    loss_cont_1 = torch.sum(d_pos ** 2)                                      # :41
    loss_cont_2 = loss_cont_1 ** 2                                           # :42-44 (the margin term is discarded)
    return loss_ce + alpha * (loss_cont_1 + loss_cont_2)                     # :45-48


# ----------------------------------------------------------------------------
# MAMC n-pairs loss  (model/loss/MAMC_loss.py:24-90) - SURVEY 8f-4
# ----------------------------------------------------------------------------
def npairs_masks(targets, p):
    """The four pair sets of MAMC_loss.py:47-55 for the n = b*p anchors (row i = sample i // p, attention i % p):
    same-attention-same-class (contains the anchor itself), sa-different-class, da-same-class, da-dc."""
    cls = torch.repeat_interleave(targets, p)                                # :43
    att = torch.arange(p).repeat(targets.shape[0])                           # :44
    sc = cls[:, None] == cls[None, :]                                        # :49
    sa = att[:, None] == att[None, :]                                        # :50
    return sc & sa, (~sc) & sa, sc & (~sa), (~sc) & (~sa)                    # :52-55


def npairs_loss(inputs, targets):
    """NPairsLoss.forward (MAMC_loss.py:34-90) without the python loop over anchors:
        L = 1/n sum_i [ sum_{j in SASC_i} log(1 + sum_{k in SADC_i u DASC_i u DADC_i} e^{s_ik - s_ij})
                      + sum_{j in SADC_i} log(1 + sum_{k in DADC_i} e^{s_ik - s_ij})
                      + sum_{j in DASC_i} log(1 + sum_{k in DADC_i} e^{s_ik - s_ij}) ],   s = x_hat x_hat^T.
    Empty positive or negative sets contribute 0 exactly as the reference's empty `repeat`/`sum` do (:64-88)."""
    b, p, _ = inputs.shape
    n = b * p
    x = F.normalize(inputs.contiguous().view(n, -1), p=2, dim=1)             # :40-42
    s = x @ x.t()                                                            # :45
    sasc, sadc, dasc, dadc = npairs_masks(targets, p)
    diff = s[:, None, :] - s[:, :, None]                                     # [i, j, k] = s_ik - s_ij   (:68,:78,:88)

    def term(pos, neg):
        e = torch.exp(diff) * neg[:, None, :].to(s.dtype)
        return (torch.log(1 + e.sum(dim=2)) * pos.to(s.dtype)).sum()

    return (term(sasc, sadc | dasc | dadc) + term(sadc, dadc) + term(dasc, dadc)) / n   # :90


def mamc_loss(pred, parts, targets, lambda_a=0.5, use_mamc=True):
    """MAMCLoss.forward, MAMC_loss.py:14-21: CE(label_smoothing=0.1) + lambda_a * n-pairs."""
    ce = F.cross_entropy(pred, targets, label_smoothing=0.1)
    return ce + lambda_a * npairs_loss(parts, targets) if use_mamc else ce


# ----------------------------------------------------------------------------
# Whole-model restatements used by bench.py's cpu_baseline leg and smoke()
# ----------------------------------------------------------------------------
_VGG16 = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg16_features():
    """All 31 layers of VGG-16 `features` (last MaxPool kept).
    model/backbone/vgg.py:56-76, BCNN.py:38-39."""
    layers, cin = [], 3
    for v in _VGG16:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def _init(m):
    """model/utils.py:5-16."""
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, val=0)


class BCNNOracle(nn.Module):
    """Reference BCNN on the CPU path (BCNN.py:30-55) with random-init VGG-16."""

    def __init__(self, num_classes=200, stage=2):
        super().__init__()
        self.stage = stage
        self.backbone = vgg16_features()
        self.backbone.apply(_init)                                # vgg.py:45-46
        self.classifier = nn.Linear(512 ** 2, num_classes)
        self.classifier.apply(_init)                              # BCNN.py:43
        if stage == 1:
            for p in self.backbone.parameters():
                p.requires_grad = False                           # BCNN.py:45-47

    def forward(self, x):
        x = self.backbone(x)
        if self.stage == 1:
            x = x.detach()
        return self.classifier(bilinear_pool(x))
