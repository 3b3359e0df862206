"""Deterministic inputs shared by oracle/gen_golden.py (which stores the
reference's outputs) and the tests (which regenerate the same inputs).
numpy's legacy RandomState (MT19937) is stable across numpy versions."""
import numpy as np
import torch


def rs_randn(seed, shape):
    return np.random.RandomState(seed).randn(*shape).astype(np.float32)


def rs_relu_randn(seed, shape):
    return np.maximum(rs_randn(seed, shape), 0.0)


def rs_signed_channels(seed, shape):
    """|N(0,1)| + 0.1 with a random sign per CHANNEL ([B,C,H,W]): every Gram entry X_i . X_j is bounded away from zero
    with mixed signs - a well-conditioned input for sign(G) sqrt(|G| + eps), whose slope 1 / (2 sqrt(|G| + eps)) blows up
    at G = 0 (with zero-mean features a few of the C*C entries land within fp32 rounding of 0 and dominate any fp32
    vs fp32 comparison of the gradient)."""
    a = np.abs(rs_randn(seed, shape)) + np.float32(0.1)
    sign = np.random.RandomState(seed + 1000).choice(np.array([-1.0, 1.0], dtype=np.float32), size=shape[1])
    return (a * sign[None, :, None, None]).astype(np.float32)


def sub(a, stride=1009):
    """Strided subsample of a flattened tensor (stride prime, offset 0)."""
    if torch.is_tensor(a):
        return a.detach().reshape(-1)[::stride].clone()
    return np.asarray(a).reshape(-1)[::stride].copy()


def seeded_init(module, seed):
    """Overwrite every parameter/buffer with small seeded values, in state_dict
    order, so that two independently constructed copies of a model agree."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            a = rs.randn(*v.shape).astype(np.float32) if v.dim() else np.float32(rs.randn())
            if k.endswith('running_var'):
                a = np.abs(a) + 0.5
            elif v.dim() >= 2:
                fan_in = int(np.prod(v.shape[1:]))
                a = a * np.float32(np.sqrt(2.0 / fan_in))
                if '_1.conv.weight' in k and v.shape[1] == 1:
                    # AP-CNN SpatialGate ConvTranspose2d [C,1,3,3]: true fan-in is C*9; keep sigmoid unsaturated
                    # (the reference itself crashes in nms.py:93 when a saturated map leaves no score > mean)
                    a = a * np.float32(0.1 / v.shape[0] ** 0.5)
            elif k.endswith('weight'):
                a = 1.0 + 0.1 * a
                if k.endswith('bn3.weight'):
                    a = 0.25 * a          # damp the residual branches so deep trunks keep O(1) activations
            else:
                a = 0.1 * a
            v.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)))


# MAMC n-pairs loss cases (oracle/gen_golden.py:gen_mamc): name -> (batch, parts, dim, labels); inputs rs_randn(300 + i)
MAMC_CASES = {
    'balanced': (10, 2, 1024, [3, 3, 7, 7, 1, 1, 9, 9, 4, 4]),
    'mixed': (6, 3, 16, [2, 5, 2, 2, 0, 5]),
    'all_distinct': (4, 1, 8, [0, 1, 2, 3]),
    'one_class': (3, 2, 8, [6, 6, 6]),
}
