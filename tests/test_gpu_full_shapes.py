"""GPU parity at the shapes the BASELINE configs actually dispatch (per-GPU batch 64 / 16), against fixtures the
REFERENCE produced at those batches (tests/golden/full_shapes.npz, oracle/gen_golden.py::gen_full) and against the
oracle on the picked samples.

Why a separate tier: at these batches the entry points take other code paths than at the B = 2 pins of
test_gpu_parity.py - the Newton-Schulz chain runs its two batch halves on two HIP queues, the covariance takes the
eight-wave 64-row kernel, the CBP Gram pairs row blocks (B = 64) or walks them cyclically (B = 16), the ROI kernels get
16 - 32 maps per workgroup - and the round-2 tests compared those paths with each other, not with the reference.
Everything goes through the default dispatch: no tuning knob is touched here.
"""
import os
import random

import numpy as np
import pytest
import torch

import hawkeye_oracle as O
from inputs import rs_randn, rs_relu_randn, sub

pytestmark = pytest.mark.gpu
DEV = 'cuda'
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(G, 'full_shapes.npz'))


@pytest.fixture(scope='module')
def F():
    import hawkeye_amd.functional as F_
    from hawkeye_amd import _lib
    assert b'gfx950' in _lib.load().hk_version()
    return F_


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a = (a.detach() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))).double().cpu().reshape(-1)
    b = (b.detach() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def per(a):
    return a.detach().double().reshape(a.shape[0], -1).cpu()


@pytest.mark.parametrize('symmetric', [True, False])
def test_mpn_head_at_batch_64_vs_reference(F, g, symmetric):
    """symmetric = True is what the MPN head runs (hk_ns_sqrtm_fwd_sym: tiles below the diagonal blocks mirrored instead
    of computed), False the full products of hk_ns_sqrtm_fwd - both against the same reference goldens.
    Covariance (C = 256: the eight-wave 64-row backward, the cyclic forward walk), Newton-Schulz at d = 256 with 5
    iterations on TWO HIP queues of 32 samples each, triuvec, and the whole backward - MPNCOV.py:105-230 at the batch
    configs[2] is benchmarked at.  Every sample is pinned through its sums, the first / middle / last sample of each
    queue's half through a strided subsample of every tensor, and the same samples against the oracle in full."""
    xn, wn = rs_relu_randn(3101, (64, 256, 14, 14)), rs_randn(3102, (64, 32896, 1))
    xg = t(xn).to(DEV).requires_grad_(True)
    cov = F.covpool(xg)
    cov.retain_grad()
    sq = F.sqrtm(cov, 5, symmetric=symmetric)
    sq.retain_grad()
    tv = F.triuvec(sq)
    (tv * t(wn).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    # every one of the 64 samples
    np.testing.assert_allclose(per(cov).sum(1), g['mpn_cov_sum'], rtol=2e-5)
    np.testing.assert_allclose(per(cov).abs().sum(1), g['mpn_cov_abs'], rtol=1e-5)
    np.testing.assert_allclose(per(sq).abs().sum(1), g['mpn_sq_abs'], rtol=1e-5)
    np.testing.assert_allclose(per(tv).sum(1), g['mpn_tv_sum'], rtol=1e-5)
    np.testing.assert_allclose(per(cov.grad).abs().sum(1), g['mpn_dcov_abs'], rtol=1e-4)
    np.testing.assert_allclose(per(xg.grad).abs().sum(1), g['mpn_dx_abs'], rtol=1e-4)
    picks = g['pick'].tolist()
    for s in picks:
        assert rel(sub(cov[s].cpu(), 61), g[f'mpn_cov_{s}']) < 2e-6, s
        assert rel(sub(sq[s].cpu(), 61), g[f'mpn_sq_{s}']) < 1e-5, s
        assert rel(sub(cov.grad[s].cpu(), 61), g[f'mpn_dcov_{s}']) < 1e-4, s
        assert rel(sub(xg.grad[s].cpu(), 61), g[f'mpn_dx_{s}']) < 1e-4, s
    # the picked samples, every element, against the oracle (per-sample independent: run on the picks alone)
    xo = t(xn[picks]).requires_grad_(True)
    co = O.covpool(xo)
    co.retain_grad()
    so = O.sqrtm(co, 5)
    (O.triuvec(so) * t(wn[picks])).sum().backward()
    assert rel(cov[picks], co) < 2e-6 and rel(sq[picks], so) < 1e-5
    assert rel(cov.grad[picks], co.grad) < 1e-4 and rel(xg.grad[picks], xo.grad) < 1e-4
    assert torch.equal(cov.detach(), cov.detach().transpose(1, 2))
    if symmetric:      # the 128 x 128 block right of the diagonal is written twice
        assert torch.equal(sq.detach()[:, :128, 128:], sq.detach()[:, 128:, :128].transpose(1, 2))


@pytest.mark.parametrize('b', [64, 16])
def test_cbp_at_config_batches_vs_reference(F, g, b):
    """Compact bilinear pooling at D = 6000, C = 512 at the benchmark batch (64) and at configs/CBCNN_S2.yaml's (16 =
    the first 16 samples of the same inputs): this path's Gram route against what the reference's FFT route returned
    (CBCNN.py:96-135) - y of the picked samples in full, every sample through its sums, dX subsampled."""
    xn, wn = rs_relu_randn(3201, (64, 512, 14, 14))[:b], rs_randn(3202, (64, 6000))[:b]
    dev = torch.device('cuda', torch.cuda.current_device()) if DEV == 'cuda' else torch.device(DEV)
    plan = F.CbpPlan(*F.sketch_hashes(512, 512, 6000), 6000, dev)
    xg = t(xn).to(DEV).requires_grad_(True)
    y = F.compact_bilinear_pool(xg, plan)
    (y * t(wn).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(per(y).abs().sum(1), g['cbp_y_abs'][:b], rtol=1e-5)
    np.testing.assert_allclose(per(y).sum(1), g['cbp_y_sum'][:b], rtol=1e-4, atol=1e-5)
    # dX: the gradient goes through dc = du / (2 sqrt|c|), which amplifies the float32 error of a bin that cancels: ANY
    # float32 route is cond[s] (stored per sample: first-order estimate, gen_full) away from the exact gradient - median
    # 5e-4, 1e-2 .. 4e-1 for the four samples with a cancelling bin, where the reference's own float32 run is 1e-2 .. 6e-1
    # off too.  The pin is the reference run in float64; the bound per sample is 1e-4 + 2 cond[s].
    cond = g['cbp_cond'][:b]
    got = per(xg.grad).abs().sum(1).numpy()
    assert (np.abs(got / g['cbp_dx64_abs'][:b] - 1) < 1e-4 + 2 * cond).all()
    assert np.median(np.abs(got / g['cbp_dx64_abs'][:b] - 1)) < 1e-4
    picks = [s for s in g['pick'].tolist() if s < b]
    assert len(picks) >= 3
    for s in picks:
        assert rel(y[s], g[f'cbp_y_{s}']) < 1e-5, s
        np.testing.assert_allclose(y[s].detach().cpu().numpy(), g[f'cbp_y_{s}'], rtol=2e-3, atol=2e-6)     # every bin, sign included
        assert rel(sub(xg.grad[s].cpu(), 61), g[f'cbp_dx64_{s}']) < 1e-4 + 2 * cond[s], s
        assert rel(sub(xg.grad[s].cpu(), 61), g[f'cbp_dx_{s}']) < 2e-4 + 4 * cond[s], s       # vs the reference's float32 run
    # two of them against the oracle's restatement of the FFT route
    two = picks[:1] + picks[-1:]
    xo = t(xn[two]).requires_grad_(True)
    yo = O.compact_bilinear_pool(xo, 6000)
    (yo * t(wn[two])).sum().backward()
    assert rel(y[two], yo) < 1e-5 and rel(xg.grad[two], xo.grad) < 2e-4 + 4 * float(cond[two].max())
    assert torch.allclose(y.norm(dim=1), torch.ones(b, device=DEV), atol=1e-5)


def test_apcnn_attention_pool_at_batch_16_vs_reference(g):
    """PyramidAttentions on 256-channel pyramids at configs/APCNN.yaml's batch (16 x 256 x {56, 28, 14}^2) with the
    reference module's own weights: pooled vectors, masks and the gradient at all three levels (APCNN.py:236-268,
    :561-563) - hk_att_pool_fwd / bwd at the sizes the 8142-class config launches them."""
    from hawkeye_amd.model.methods.APCNN import PyramidAttentions
    apn = PyramidAttentions(channel_size=256)
    apn.load_state_dict({k[6:].replace('__', '.'): t(g[k]) for k in g.files if k.startswith('apn_w_')}, strict=True)
    apn = apn.to(DEV)
    feats = [t(rs_randn(3310 + i, (16, 256, s, s))).to(DEV).requires_grad_(True) for i, s in enumerate((56, 28, 14))]
    pooled, gaps, masks = apn(feats)
    sum((p * t(rs_randn(3320 + i, tuple(p.shape))).to(DEV)).sum() for i, p in enumerate(pooled)).backward()
    torch.cuda.synchronize()
    for lvl, (p, m, f) in enumerate(zip(pooled, masks, feats)):
        k = lvl + 3
        assert rel(p, g[f'apn_pooled{k}']) < 1e-5, k
        np.testing.assert_allclose(per(m).sum(1), g[f'apn_mask{k}_sum'], rtol=1e-5)
        np.testing.assert_allclose(per(f.grad).abs().sum(1), g[f'apn_df{k}_abs'], rtol=1e-5)
        for s in g['pick16'].tolist():
            assert rel(sub(f.grad[s].cpu(), 211), g[f'apn_df{k}_{s}']) < 1e-5, (k, s)


LEVELS = ((8, 64, 5), (16, 128, 3), (32, 256, 1))


def _compact(rois, cnt):
    rows = []
    for i in range(rois.shape[0]):
        k = int(cnt[i])
        rows.append(torch.cat([torch.full((k, 1), float(i)), rois[i, :k].cpu()], 1))
    return torch.cat(rows, 0)


def test_apcnn_roi_pipeline_at_batch_16_8142_classes_vs_reference(F, g):
    """get_att_roi with the iNat2018 border band (8142 classes: rows / columns 0.1 - 0.9, APCNN.py:451-455) on 16
    images, then get_roi_crop_feat on the 16 x 512 x 56 x 56 map in train mode (the reference's python-random drop
    sequence replayed) and in eval mode: ROI tables bit-exact, crops and their gradient against the reference."""
    masks = [t(1.0 / (1.0 + np.exp(-2.0 * rs_randn(3330 + l, (16, 1, hw, hw))))).float() for l, hw in enumerate((56, 28, 14))]
    tabs, rois = [], []
    for m, (s, a, k), key in zip(masks, LEVELS, ('ap_roi3', 'ap_roi4', 'ap_roi5')):
        r, c = F.att_roi_select(m.to(DEV), s, a, 448, 448, 8142, 0.05, k)
        np.testing.assert_array_equal(_compact(r, c).numpy(), g[key])                 # boxes and scores, bit-exact
        tabs.append((r, c))
        rois.append(t(g[key]))
    drops = [None if l == 0 else (int(l), int(i)) for l, i in g['ap_drops']]
    assert any(d is not None and d[0] == 3 for d in drops) and any(d is not None and d[0] == 4 for d in drops)
    xn, wn = rs_randn(3340, (16, 512, 56, 56)), rs_randn(3341, (16, 512, 56, 56))
    allr = torch.cat(rois, 0)
    for mode in ('train', 'eval'):
        box = torch.zeros(16, 4)
        drop = torch.tensor([[0., 0., -1., -1.]] * 16)
        for i in range(16):
            r = allr[allr[:, 0] == i] / 8
            box[i] = torch.cat([r[:, 1:3].min(0)[0], r[:, 3:5].max(0)[0]])
            if mode == 'train' and drops[i] is not None:
                src = rois[0] if drops[i][0] == 3 else rois[1]
                drop[i] = (src[src[:, 0] == i] / 8)[drops[i][1], 1:5]
        # the device-side union boxes are the same boxes
        dbox, _ = F.roi_boxes(tabs, None, 8.0)
        assert torch.equal(dbox.cpu(), box)
        xg = t(xn).to(DEV).requires_grad_(True)
        y = F.roi_crop_resize(xg, box.to(DEV), drop.to(DEV), mode == 'train')
        (y * t(wn).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        np.testing.assert_allclose(per(y).abs().sum(1), g[f'ap_y_abs_{mode}'], rtol=1e-5)
        np.testing.assert_allclose(per(xg.grad).abs().sum(1), g[f'ap_dx_abs_{mode}'], rtol=1e-5)
        for s in g['pick16'].tolist():
            assert rel(sub(y[s].cpu(), 211), g[f'ap_y_{mode}_{s}']) < 1e-6, (mode, s)
            assert rel(sub(xg.grad[s].cpu(), 211), g[f'ap_dx_{mode}_{s}']) < 1e-6, (mode, s)
        # one image in full against the oracle (per-image independent)
        i = 7
        ri = [r[r[:, 0] == i].clone() for r in rois]
        for r in ri:
            r[:, 0] = 0
        xo = t(xn[i:i + 1]).requires_grad_(True)
        yo = O.roi_crop_feat(xo, ri, 8, training=(mode == 'train'), drops=[drops[i]])
        (yo * t(wn[i:i + 1])).sum().backward()
        assert rel(y[i:i + 1], yo) < 1e-6 and rel(xg.grad[i:i + 1], xo.grad) < 1e-6


def test_classifier_backward_full_shape_is_reproducible_under_memory_pressure():
    """linear_bwd64_kernel at the metric's shape (64 x 262144 -> 200): the barrier that ends a pipeline unit waits with a
    COUNTED s_waitcnt vmcnt(n) (everything but the pieces and stores issued since) - if a count were off by one, or if
    stores did not retire in issue order, a unit would occasionally be consumed before it has landed.  Forty launches, half
    of them while a side stream hammers HBM with copies (memory latency several times the quiet one): dy, dW and db must be
    bit-identical every time and agree with fp64."""
    from hawkeye_amd import _lib
    from hawkeye_amd._lib import ptr, stream
    lib = _lib.load()
    B, J, K = 64, 262144, 200
    gen = torch.Generator().manual_seed(5)
    y = torch.randn(B, J, generator=gen).to(DEV)
    w = (torch.randn(K, J, generator=gen) / 512).to(DEV)
    g = torch.randn(B, K, generator=gen).to(DEV)
    ref = None
    junk_a, junk_b = torch.empty(64 * 1024 * 1024, device=DEV), torch.empty(64 * 1024 * 1024, device=DEV)
    side = torch.cuda.Stream()
    for it in range(40):
        dy = torch.full((B, J), float('nan'), device=DEV)
        dw = torch.full((K, J), float('nan'), device=DEV)
        db = torch.full((K,), float('nan'), device=DEV)
        if it % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(6):
                    junk_b.copy_(junk_a)
        assert lib.hk_linear_bwd(ptr(y), ptr(w), ptr(g), ptr(dy), ptr(dw), ptr(db), B, J, K, stream()) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = (dy, dw, db)
            assert rel(dy, g.double().cpu() @ w.double().cpu()) < 2e-6
            assert rel(dw, g.double().cpu().t() @ y.double().cpu()) < 2e-6
            assert rel(db, g.double().cpu().sum(0)) < 2e-6
        else:
            assert torch.equal(dy, ref[0]) and torch.equal(dw, ref[1]) and torch.equal(db, ref[2]), it


def test_trunk_epilogues_at_the_metric_shape(F):
    """hk_bias_relu_* at the first VGG stage of the metric's configuration - a 64 x 64 x 448 x 448 channels_last map, 3.29 GB: byte
    offsets beyond 2^32, 205 M float4 per pass - against the framework's own ops ON THE DEVICE (the same fp32 arithmetic: the
    values and the input gradients must be bit-identical, here too), the bias gradients against the device's float64 sum; and
    run to run the same bits.  (tests/test_gpu_kernels.py pins the same kernels on small maps against torch on the CPU.)"""
    n, c, h, w = 64, 64, 448, 448
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.empty(n, c, h, w, device=DEV, memory_format=torch.channels_last).normal_(-0.3, 1.0, generator=gen)
    b = torch.randn(c, device=DEV, generator=gen) * 0.3
    # plain form
    dy = torch.empty_like(x).normal_(0.0, 1.0, generator=gen)
    xr = x.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = torch.relu(xr + br.view(1, -1, 1, 1))
    yr.backward(dy)
    xg, bg = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.bias_relu(xg * 1.0, bg)
    assert torch.equal(y, yr.detach())
    y.backward(dy)
    assert torch.equal(xg.grad, xr.grad)
    db64 = xr.grad.double().sum((0, 2, 3))
    assert rel(bg.grad, db64) < 1e-6 and rel(br.grad, db64) < 1e-4           # (ours: a fixed-order sum; ATen's own float32 reduction)
    del yr, y, xr, xg, dy
    torch.cuda.empty_cache()
    # pooled form
    dp = torch.empty(n, c, h // 2, w // 2, device=DEV, memory_format=torch.channels_last).normal_(0.0, 1.0, generator=gen)
    xr = x.clone().requires_grad_(True)
    pr = torch.nn.functional.max_pool2d(torch.relu(xr + b.view(1, -1, 1, 1)), 2, 2)
    pr.backward(dp)
    grads = []
    for _ in range(2):
        xg, bg = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        p = F.bias_relu_pool(xg * 1.0, bg)
        assert torch.equal(p, pr.detach())
        p.backward(dp)
        assert torch.equal(xg.grad, xr.grad)
        grads.append(bg.grad.clone())
        del p, xg
    assert torch.equal(grads[0], grads[1]) and rel(grads[0], xr.grad.double().sum((0, 2, 3))) < 1e-6


def test_first_convolution_kernels_at_the_metric_shape(F):
    """hk_conv1_bias_relu_fwd / bwd at the metric's shape - 64 x 3 x 448 x 448 in, a 3.29 GB map out (byte offsets beyond 2^32):
    the output against the library's convolution + ReLU on the device; the weight / bias gradients (a) of the first two images
    against a float64 evaluation on the CPU, (b) of the whole batch against the SUM of the gradients of its four quarters
    computed by the same kernels on their own - additivity ties the large-offset rows to the small-offset ones - and (c)
    against the library's own float32 backward (loosely: its weight gradient is an atomic sum over 12.8 M pixels)."""
    n, h, w = 64, 448, 448
    gen = torch.Generator(device=DEV).manual_seed(23)
    x = torch.empty(n, 3, h, w, device=DEV, memory_format=torch.channels_last).normal_(0.0, 1.0, generator=gen)
    wt = (torch.randn(64, 3, 3, 3, device=DEV, generator=gen) * 0.3)
    b = torch.randn(64, device=DEV, generator=gen) * 0.2
    dy = torch.empty(n, 64, h, w, device=DEV, memory_format=torch.channels_last).normal_(0.0, 1.0, generator=gen)

    def ours(xs, dys):
        wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = F.conv1_bias_relu(xs, wg, bg)
        y.backward(dys)
        return y.detach(), wg.grad, bg.grad
    y, dw, db = ours(x, dy)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.relu(torch.nn.functional.conv2d(x, wr, br, padding=1))
    assert rel(y, yr) < 1e-6
    yr.backward(dy)
    assert rel(dw, wr.grad) < 2e-3 and rel(db, br.grad) < 2e-3                     # (c)
    del yr
    parts = [ours(x[i:i + 16], dy[i:i + 16]) for i in range(0, n, 16)]
    assert all(torch.equal(p[0], y[i:i + 16]) for p, i in zip(parts, range(0, n, 16)))
    assert rel(dw, sum(p[1].double() for p in parts)) < 1e-5 and rel(db, sum(p[2].double() for p in parts)) < 1e-5    # (b)
    y2, dw2, db2 = ours(x[:2], dy[:2])
    w64, b64 = wt.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    # (the float64 evaluation takes the SIGN pattern of the float32 output: of 25.7 M outputs a dozen lie within rounding of the
    #  ReLU's kink, and each one that flips moves the gradient by 1e-4 of its norm - that is the kink, not the kernel)
    g64 = dy[:2].cpu().double() * (y2.cpu() > 0)
    torch.nn.functional.conv2d(x[:2].cpu().double(), w64, b64, padding=1).backward(g64)
    assert rel(dw2, w64.grad) < 1e-5 and rel(db2, b64.grad) < 1e-5                 # (a)
    assert torch.equal(ours(x, dy)[1], dw)                                          # run to run: the same bits
