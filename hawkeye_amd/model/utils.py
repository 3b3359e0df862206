"""Initialisers and lenient checkpoint loading.

The initial statistics matter for synthetic-data runs and for the seeded whole-model parity tests, so they follow
the reference (model/utils.py:5-28): convolutions Kaiming-normal on fan-out for ReLU with zero bias, BatchNorm2d to
(1, 0), Linear layers Kaiming-normal (default fan-in) with zero bias.  `load_state_dict` is the reference's
"take what matches by name and shape" loader used for ImageNet checkpoints.
"""

import torch.nn as nn
from torch.nn import init


def _zero_bias(layer):
    if getattr(layer, 'bias', None) is not None:
        init.zeros_(layer.bias)


def initialize_weights(m):
    """`module.apply(initialize_weights)`-style visitor."""
    if isinstance(m, nn.Conv2d):
        init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        _zero_bias(m)
        return
    if isinstance(m, nn.BatchNorm2d):
        init.ones_(m.weight)
        init.zeros_(m.bias)
        return
    if isinstance(m, nn.Linear):
        init.kaiming_normal_(m.weight.data)
        _zero_bias(m)


def load_state_dict(model, state_dict):
    """Copy the entries of `state_dict` whose key exists in `model` with the same shape; keep the rest as initialised."""
    target = model.state_dict()
    usable = {}
    for key, value in state_dict.items():
        if key in target and tuple(value.shape) == tuple(target[key].shape):
            usable[key] = value
    target.update(usable)
    model.load_state_dict(target)
    return sorted(usable)


def wide_linear(layer, x):
    """`layer(x)` for the classifier / part FCs that sit directly on a pooled vector (tens of thousands of features,
    a few hundred outputs) on the hand-written kernels: hk_linear_fwd (linear_skinny_kernel: W and the features streamed
    once through LDS-DMA) and hk_linear_bwd (linear_bwd64_kernel: dy = g W and dW = g^T y in one launch; OSME's part FCs:
    linear_bwd16_kernel) - SURVEY 8f-1, replaces `self.classifier(x)` at model/methods/BCNN.py:54, CBCNN.py:34,
    MPNCOV.py:37 and `self.fcs[p](...)` at OSME.py:43.  The `nn.Linear` stays the parameter holder, so `state_dict` keys,
    initialisers and optimiser groups are untouched.  There is no switch back to the library GEMMs."""
    from .. import functional as F        # hawkeye_amd.functional
    return F.linear(x, layer.weight, layer.bias)


def pooled_classifier(model, feats):
    """`model.classifier(model.bilinear_pooling(feats))` (model/methods/BCNN.py:53-54) as ONE autograd node of two kernels
    each way: the default sqrt(G + 1e-5) pooling with the classifier behind it - whose backward hands <y, dy> to the Gram
    backward in closed form (hk_bcnn_pool_bwd_tdot: one launch, no second pass over dX) - or, for
    BilinearPooling(signed_sqrt=True), the reference's commented alternative (BCNN.py:23-24) with the l2 scale folded
    into the classifier.  The modules stay what the reference has - `bilinear_pooling` parameter-free and
    callable on its own, `classifier` an nn.Linear holding the parameters."""
    from .. import functional as F
    layer = model.classifier
    if _hooked(model.bilinear_pooling) or _hooked(layer):
        # somebody registered forward hooks on one of the two modules (feature extraction, CAM tooling): the fused node
        # never goes through their __call__, so take the two-node composition - the same kernels, one launch more in the
        # backward - with both modules called the way the reference calls them (BCNN.py:53-54)
        return _call_on_kernels(layer, model.bilinear_pooling(feats))
    if model.bilinear_pooling.signed_sqrt:
        return F.ssqrt_pool_linear(feats, layer.weight, layer.bias)
    return F.bilinear_pool_linear(feats, layer.weight, layer.bias)


def _hooked(module):
    import torch.nn.modules.module as M
    return bool(module._forward_hooks or module._forward_pre_hooks or M._global_forward_hooks or M._global_forward_pre_hooks)


def _call_on_kernels(layer, x):
    """`layer(x)` through nn.Module.__call__ (hooks fire) with the product computed by hk_linear_fwd / bwd instead of the
    library GEMM: the instance's `forward` is pointed at wide_linear for the duration of the call."""
    had = layer.__dict__.get('forward')
    layer.forward = lambda inp: wide_linear(layer, inp)
    try:
        return layer(x)
    finally:
        if had is None:
            del layer.forward
        else:
            layer.forward = had
