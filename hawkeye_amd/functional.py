"""torch.autograd.Function wrappers over the C ABI (include/hawkeye_hip.h).

Same pattern the reference already uses for MPN-COV (model/methods/MPNCOV.py:105-230:
hand-written forward/backward `Function`s), applied to every op on the hot path.
Outputs and workspaces are allocated through torch's caching allocator; kernels
are enqueued on torch's current HIP stream; nothing synchronises the host.
fp32 only, HIP tensors only - there is no CPU fallback (see hawkeye_amd/_lib.py).
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.HawkeyeHipError(f'hawkeye_amd ops are fp32 (reference parity target); got {t.dtype}')
    return t.contiguous()


def _on(device):
    """Make `device` the current HIP device for a host-side ABI call (memcpy + stream of that device)."""
    return torch.cuda.device(device)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------- BCNN
class _BilinearPool(torch.autograd.Function):
    """replaces BilinearPooling.forward, model/methods/BCNN.py:13-27."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        b, c, h, w = x.shape
        hw = h * w
        y = torch.empty(b, c * c, dtype=torch.float32, device=x.device)
        inv_norm = torch.empty(b, dtype=torch.float32, device=x.device)
        colsum = torch.empty(b, hw, dtype=torch.float32, device=x.device)
        nws = lib.hk_bcnn_pool_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        check(lib.hk_bcnn_pool_fwd(ptr(x), ptr(y), ptr(inv_norm), ptr(colsum), b, c, hw, ptr(ws), nws, stream()),
              'hk_bcnn_pool_fwd')
        ctx.save_for_backward(x, y, inv_norm, colsum)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, inv_norm, colsum = ctx.saved_tensors
        b, c, h, w = x.shape
        hw = h * w
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        nws = lib.hk_bcnn_pool_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        check(lib.hk_bcnn_pool_bwd(ptr(x), ptr(y), ptr(dy), ptr(inv_norm), ptr(colsum), ptr(dx), b, c, hw,
                                   ptr(ws), nws, stream()), 'hk_bcnn_pool_bwd')
        return dx


class _BilinearPoolSignedSqrt(torch.autograd.Function):
    """The reference's alternative normalisation (commented out at model/methods/BCNN.py:23-24):
    sign(G) sqrt(|G| + 1e-10), l2-normalised."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        b, c, h, w = x.shape
        hw = h * w
        y = torch.empty(b, c * c, dtype=torch.float32, device=x.device)
        inv_norm = torch.empty(b, dtype=torch.float32, device=x.device)
        nws = lib.hk_bcnn_ssqrt_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        check(lib.hk_bcnn_ssqrt_pool_fwd(ptr(x), ptr(y), ptr(inv_norm), b, c, hw, ptr(ws), nws, stream()),
              'hk_bcnn_ssqrt_pool_fwd')
        ctx.save_for_backward(x, y, inv_norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, inv_norm = ctx.saved_tensors
        b, c, h, w = x.shape
        hw = h * w
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        nws = lib.hk_bcnn_ssqrt_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        check(lib.hk_bcnn_ssqrt_pool_bwd(ptr(x), ptr(y), ptr(dy), ptr(inv_norm), ptr(dx), b, c, hw, ptr(ws), nws,
                                         stream()), 'hk_bcnn_ssqrt_pool_bwd')
        return dx


def bilinear_pool(x, signed_sqrt=False):
    """[B,C,H,W] -> [B,C*C]:  sqrt(X X^T / HW + 1e-5), l2-normalised (what the reference runs, BCNN.py:21);
    signed_sqrt=True: sign(G) sqrt(|G| + 1e-10) instead (its commented alternative, BCNN.py:23-24)."""
    return _BilinearPoolSignedSqrt.apply(x) if signed_sqrt else _BilinearPool.apply(x)


# --------------------------------------------------------------------- MPN-COV
class _Covpool(torch.autograd.Function):
    """replaces Covpool, model/methods/MPNCOV.py:105-134."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        b, c, h, w = x.shape
        m = h * w
        cov = torch.empty(b, c, c, dtype=torch.float32, device=x.device)
        mu = torch.empty(b, c, dtype=torch.float32, device=x.device)
        check(lib.hk_cov_pool_fwd(ptr(x), ptr(cov), ptr(mu), b, c, m, stream()), 'hk_cov_pool_fwd')
        ctx.save_for_backward(x, mu)
        return cov

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, mu = ctx.saved_tensors
        b, c, h, w = x.shape
        g = _f32c(g)
        dx = torch.empty_like(x)
        check(lib.hk_cov_pool_bwd(ptr(x), ptr(mu), ptr(g), ptr(dx), b, c, h * w, stream()), 'hk_cov_pool_bwd')
        return dx


class _Sqrtm(torch.autograd.Function):
    """replaces Sqrtm, model/methods/MPNCOV.py:137-202."""

    @staticmethod
    def forward(ctx, a, iter_n, symmetric=False, literal_backward=False):
        lib = _lib.load()
        a = _f32c(a)
        b, d, _ = a.shape
        out = torch.empty_like(a)
        norm_a = torch.empty(b, dtype=torch.float32, device=a.device)
        slots = max(iter_n - 1, 1)
        ysave = torch.empty(b, slots, d, d, dtype=torch.float32, device=a.device)
        zsave = torch.empty(b, slots, d, d, dtype=torch.float32, device=a.device)
        nws = lib.hk_ns_sqrtm_ws_bytes(b, d, iter_n, 0)
        ws = _ws(nws, a.device)
        fwd = lib.hk_ns_sqrtm_fwd_sym if symmetric else lib.hk_ns_sqrtm_fwd
        check(fwd(ptr(a), ptr(out), ptr(norm_a), ptr(ysave), ptr(zsave), b, d, iter_n, ptr(ws), nws, stream()),
              'hk_ns_sqrtm_fwd')
        ctx.iter_n = iter_n
        ctx.literal_backward = bool(literal_backward)
        ctx.save_for_backward(a, out, norm_a, ysave, zsave)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, out, norm_a, ysave, zsave = ctx.saved_tensors
        b, d, _ = a.shape
        g = _f32c(g)
        da = torch.empty_like(a)
        nws = lib.hk_ns_sqrtm_ws_bytes(b, d, ctx.iter_n, 1)
        ws = _ws(nws, a.device)
        # 34 products (exact for any input) or, on request, the reference's 38 literally (include/hawkeye_hip.h)
        fn = lib.hk_ns_sqrtm_bwd_general if ctx.literal_backward else lib.hk_ns_sqrtm_bwd
        check(fn(ptr(a), ptr(out), ptr(norm_a), ptr(ysave), ptr(zsave), ptr(g), ptr(da),
                 b, d, ctx.iter_n, ptr(ws), nws, stream()), 'hk_ns_sqrtm_bwd')
        return da, None, None, None


class _SqrtmTriuvec(torch.autograd.Function):
    """Sqrtm followed by Triuvec (MPNCOV.py:88-92) with the packed upper triangle written by the chain's last product
    (hk_ns_sqrtm_triu_fwd); backward = Triuvec's scatter, then Sqrtm.backward."""

    @staticmethod
    def forward(ctx, a, iter_n, symmetric=False):
        lib = _lib.load()
        a = _f32c(a)
        b, d, _ = a.shape
        out = torch.empty_like(a)
        tv = torch.empty(b, d * (d + 1) // 2, 1, dtype=torch.float32, device=a.device)
        norm_a = torch.empty(b, dtype=torch.float32, device=a.device)
        slots = max(iter_n - 1, 1)
        ysave = torch.empty(b, slots, d, d, dtype=torch.float32, device=a.device)
        zsave = torch.empty(b, slots, d, d, dtype=torch.float32, device=a.device)
        nws = lib.hk_ns_sqrtm_ws_bytes(b, d, iter_n, 0)
        ws = _ws(nws, a.device)
        check(lib.hk_ns_sqrtm_triu_fwd(ptr(a), ptr(out), ptr(tv), ptr(norm_a), ptr(ysave), ptr(zsave), b, d, iter_n,
                                       1 if symmetric else 0, ptr(ws), nws, stream()), 'hk_ns_sqrtm_triu_fwd')
        ctx.iter_n = iter_n
        ctx.save_for_backward(a, out, norm_a, ysave, zsave)
        return tv

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, out, norm_a, ysave, zsave = ctx.saved_tensors
        b, d, _ = a.shape
        g = _f32c(g)
        dsq = torch.empty_like(a)
        check(lib.hk_triu_vec_bwd(ptr(g), ptr(dsq), b, d, stream()), 'hk_triu_vec_bwd')
        da = torch.empty_like(a)
        nws = lib.hk_ns_sqrtm_ws_bytes(b, d, ctx.iter_n, 1)
        ws = _ws(nws, a.device)
        check(lib.hk_ns_sqrtm_bwd(ptr(a), ptr(out), ptr(norm_a), ptr(ysave), ptr(zsave), ptr(dsq), ptr(da),
                                  b, d, ctx.iter_n, ptr(ws), nws, stream()), 'hk_ns_sqrtm_bwd')
        return da, None, None


class _Triuvec(torch.autograd.Function):
    """replaces Triuvec, model/methods/MPNCOV.py:205-230 (keeps its [B, L, 1] output shape)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        b, d, _ = x.shape
        y = torch.empty(b, d * (d + 1) // 2, 1, dtype=torch.float32, device=x.device)
        check(lib.hk_triu_vec_fwd(ptr(x), ptr(y), b, d, stream()), 'hk_triu_vec_fwd')
        ctx.dims = (b, d)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        b, d = ctx.dims
        g = _f32c(g)
        dx = torch.empty(b, d, d, dtype=torch.float32, device=g.device)
        check(lib.hk_triu_vec_bwd(ptr(g), ptr(dx), b, d, stream()), 'hk_triu_vec_bwd')
        return dx


def covpool(x):
    return _Covpool.apply(x)


def sqrtm(x, iter_n, symmetric=False, literal_backward=False):
    """Newton-Schulz square root of x [B,d,d] (MPNCOV.py:137-202; any square matrices).
    symmetric=True: the caller guarantees x == x^T (a covariance: what MPNCOV feeds it) - the forward then computes only
    the tiles on or above the diagonal blocks of every product (hk_ns_sqrtm_fwd_sym; wrong for other inputs).
    The backward shares one product between Y_i Z_i and Z_i Y_i (the iterates are polynomials in one matrix: they commute
    for every input); literal_backward=True runs the reference's 38 products literally (A/B checks)."""
    return _Sqrtm.apply(x, int(iter_n), bool(symmetric), bool(literal_backward))


def triuvec(x):
    return _Triuvec.apply(x)


def sqrtm_triuvec(x, iter_n, symmetric=False):
    """triuvec(sqrtm(x, iter_n, symmetric)) in one chain of launches: the last Newton-Schulz product writes the packed
    upper triangle itself (same bits as the two calls)."""
    return _SqrtmTriuvec.apply(x, int(iter_n), bool(symmetric))


# --------------------------------------------------------------------- CBP
def sketch_hashes(input_dim1, input_dim2, output_dim):
    """The reference's fixed count-sketch hashes (CBCNN.py:76-91): legacy numpy
    RandomState seeds 1/3 (h1/s1) and 5/7 (h2/s2).  Saves and restores the
    global numpy RNG state so constructing a model does not disturb user code."""
    state = np.random.get_state()
    try:
        np.random.seed(1)
        h1 = np.random.randint(output_dim, size=input_dim1)
        np.random.seed(3)
        s1 = 2 * np.random.randint(2, size=input_dim1) - 1
        np.random.seed(5)
        h2 = np.random.randint(output_dim, size=input_dim2)
        np.random.seed(7)
        s2 = 2 * np.random.randint(2, size=input_dim2) - 1
    finally:
        np.random.set_state(state)
    return h1.astype(np.int32), s1.astype(np.float32), h2.astype(np.int32), s2.astype(np.float32)


class CbpPlan:
    """Device-side CSR plan (bin -> signed Gram entries) built once per device."""

    def __init__(self, h1, s1, h2, s2, output_dim, device):
        lib = _lib.load()
        assert h1.ndim == 1 and s1.ndim == 1 and len(h1) == len(s1)            # CBCNN.py:151-152
        assert len(h1) == len(h2), 'the Gram route needs input_dim1 == input_dim2'
        assert np.all(h1 >= 0) and np.all(h1 < output_dim)                      # CBCNN.py:153
        assert np.all(h2 >= 0) and np.all(h2 < output_dim)
        self.C, self.D, self.device = len(h1), int(output_dim), device
        h1 = np.ascontiguousarray(h1, dtype=np.int32)
        h2 = np.ascontiguousarray(h2, dtype=np.int32)
        s1 = np.ascontiguousarray(s1, dtype=np.float32)
        s2 = np.ascontiguousarray(s2, dtype=np.float32)
        self.blob = torch.empty(lib.hk_cbp_plan_bytes(self.C, self.D), dtype=torch.uint8, device=device)
        with _on(device):
            check(lib.hk_cbp_plan_build(h1.ctypes.data, s1.ctypes.data, h2.ctypes.data, s2.ctypes.data,
                                        self.C, self.D, ptr(self.blob), stream()), 'hk_cbp_plan_build')

    def __del__(self):
        # the library keeps a host-side note per plan ADDRESS: forget it before torch can hand the memory to someone else
        try:
            if getattr(self, 'blob', None) is not None:
                _lib.load().hk_cbp_plan_destroy(ptr(self.blob))
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class _CompactBilinearPool(torch.autograd.Function):
    """replaces CompactBilinearPooling.forward, model/methods/CBCNN.py:96-135."""

    @staticmethod
    def forward(ctx, x, plan):
        lib = _lib.load()
        x = _f32c(x)
        b, c, h, w = x.shape
        hw, d = h * w, plan.D
        if plan.C != c:
            raise _lib.HawkeyeHipError(f'compact_bilinear_pool: the plan was built for {plan.C} channels, input has {c}')
        y = torch.empty(b, d, dtype=torch.float32, device=x.device)
        c_raw = torch.empty(b, d, dtype=torch.float32, device=x.device)
        inv_norm = torch.empty(b, dtype=torch.float32, device=x.device)
        nws = lib.hk_cbp_ws_bytes(b, c, hw, d)
        ws = _ws(nws, x.device)
        check(lib.hk_cbp_fwd(ptr(x), ptr(plan.blob), ptr(y), ptr(c_raw), ptr(inv_norm), b, c, hw, d,
                             ptr(ws), nws, stream()), 'hk_cbp_fwd')
        ctx.plan = plan
        ctx.save_for_backward(x, y, c_raw, inv_norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, c_raw, inv_norm = ctx.saved_tensors
        plan = ctx.plan
        b, c, h, w = x.shape
        hw, d = h * w, plan.D
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        nws = lib.hk_cbp_ws_bytes(b, c, hw, d)
        ws = _ws(nws, x.device)
        check(lib.hk_cbp_bwd(ptr(x), ptr(plan.blob), ptr(y), ptr(c_raw), ptr(inv_norm), ptr(dy), ptr(dx),
                             b, c, hw, d, ptr(ws), nws, stream()), 'hk_cbp_bwd')
        return dx, None


def compact_bilinear_pool(x, plan):
    return _CompactBilinearPool.apply(x, plan)


def _bgemm_raw(lib, a, b, out, trans_a, trans_b, m, n, k, nb):
    check(lib.hk_bgemm_f32(ptr(a), a.shape[2], a.shape[1] * a.shape[2], int(trans_a), ptr(b), b.shape[2], b.shape[1] * b.shape[2],
                           int(trans_b), ptr(out), n, m * n, m, n, k, nb, 1.0, 0.0, 0.0, stream()), 'hk_bgemm_f32')


class _CbpCrossSum(torch.autograd.Function):
    """Count sketch of the summed outer product of TWO different maps (CompactBilinearPooling.forward with bottom2 given,
    CBCNN.py:96-130, before its signed sqrt): c[b,k] = sum_{(i,j) -> k} s1_i s2_j (X1 X2^T)[b,i,j] - the cross Gram on the
    generic MFMA tiles (hk_bgemm_f32), the plan's CSR gather over it (hk_cbp_bin_matrix); backward: dG from dc
    (hk_cbp_unbin_matrix), dX1 = dG X2, dX2 = dG^T X1."""

    @staticmethod
    def forward(ctx, x1, x2, plan):
        lib = _lib.load()
        x1, x2 = _f32c(x1), _f32c(x2)
        b, c, h, w = x1.shape
        if tuple(x2.shape) != (b, c, h, w) or plan.C != c:
            raise _lib.HawkeyeHipError(f'compact_bilinear_pool: inputs {tuple(x1.shape)} / {tuple(x2.shape)}, plan for {plan.C} channels')
        hw, d = h * w, plan.D
        g = torch.empty(b, c, c, dtype=torch.float32, device=x1.device)
        _bgemm_raw(lib, x1.view(b, c, hw), x2.view(b, c, hw), g, False, True, c, c, hw, b)
        c_raw = torch.empty(b, d, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_bin_matrix(ptr(g), ptr(plan.blob), ptr(c_raw), b, c, d, stream()), 'hk_cbp_bin_matrix')
        ctx.plan = plan
        ctx.save_for_backward(x1, x2)
        return c_raw

    @staticmethod
    def backward(ctx, dc):
        lib = _lib.load()
        x1, x2 = ctx.saved_tensors
        plan = ctx.plan
        b, c, h, w = x1.shape
        hw, d = h * w, plan.D
        dc = _f32c(dc)
        dg = torch.empty(b, c, c, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_unbin_matrix(ptr(dc), ptr(plan.blob), ptr(dg), b, c, d, stream()), 'hk_cbp_unbin_matrix')
        dx1 = dx2 = None
        if ctx.needs_input_grad[0]:
            dx1 = torch.empty_like(x1)
            _bgemm_raw(lib, dg, x2.view(b, c, hw), dx1.view(b, c, hw), False, False, c, hw, c, b)
        if ctx.needs_input_grad[1]:
            dx2 = torch.empty_like(x2)
            _bgemm_raw(lib, dg, x1.view(b, c, hw), dx2.view(b, c, hw), True, False, c, hw, c, b)
        return dx1, dx2, None


def _loc_bwd_fits(ctx, c1, c2, d):
    """hk_cbp_loc_bwd keeps dc[b,p,:], both channel columns and both hash tables of a location in LDS (144 KB of the CU's 160):
    say so in the FORWARD of a pass that will need it, not in the middle of backward()."""
    if any(ctx.needs_input_grad[:2]) and (d + 2 * (c1 + c2)) * 4 > 144 * 1024:
        raise _lib.HawkeyeHipError(f'compact_bilinear_pool(sum_pool=False): the backward holds D + 2 (C1 + C2) = {d + 2 * (c1 + c2)} '
                                   f'floats per location in LDS (limit {144 * 256}); run this shape under torch.no_grad() or reduce D')


class _CbpPerLocation(torch.autograd.Function):
    """The tensor sketch of every location on its own (CompactBilinearPooling.forward with sum_pool = False, CBCNN.py:117-128
    before its signed sqrt): c[b,h,w,k] = sum_{(i,j) -> k} s1_i s2_j x1[b,i,h,w] x2[b,j,h,w]  ->  [B,H,W,D]."""

    @staticmethod
    def forward(ctx, x1, x2, plan):
        lib = _lib.load()
        x1, x2 = _f32c(x1), _f32c(x2)
        b, c, h, w = x1.shape
        if tuple(x2.shape) != (b, c, h, w) or plan.C != c:
            raise _lib.HawkeyeHipError(f'compact_bilinear_pool: inputs {tuple(x1.shape)} / {tuple(x2.shape)}, plan for {plan.C} channels')
        _loc_bwd_fits(ctx, c, c, plan.D)
        out = torch.empty(b, h, w, plan.D, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_loc_fwd(ptr(x1), ptr(x2), ptr(plan.blob), ptr(out), b, c, h * w, plan.D, stream()), 'hk_cbp_loc_fwd')
        ctx.plan = plan
        ctx.save_for_backward(x1, x2)
        return out

    @staticmethod
    def backward(ctx, dc):
        lib = _lib.load()
        x1, x2 = ctx.saved_tensors
        plan = ctx.plan
        b, c, h, w = x1.shape
        dc = _f32c(dc)
        dx1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        check(lib.hk_cbp_loc_bwd(ptr(x1), ptr(x2), ptr(dc), ptr(plan.blob), ptr(dx1), ptr(dx2), b, c, h * w, plan.D, stream()),
              'hk_cbp_loc_bwd')
        return dx1, dx2, None


class CbpRectPlan:
    """Device-side plan for input_dim1 != input_dim2 (CompactBilinearPooling(C1, C2, D), CBCNN.py:68-94): hashes, signs and
    the CSR table bin -> signed entries of the C1 x C2 cross Gram.  No host-side state in the library."""

    def __init__(self, h1, s1, h2, s2, output_dim, device):
        lib = _lib.load()
        assert h1.ndim == 1 and s1.ndim == 1 and len(h1) == len(s1)            # CBCNN.py:151-152
        assert h2.ndim == 1 and s2.ndim == 1 and len(h2) == len(s2)
        assert np.all(h1 >= 0) and np.all(h1 < output_dim)                      # CBCNN.py:153
        assert np.all(h2 >= 0) and np.all(h2 < output_dim)
        self.C1, self.C2, self.D, self.device = len(h1), len(h2), int(output_dim), device
        h1 = np.ascontiguousarray(h1, dtype=np.int32)
        h2 = np.ascontiguousarray(h2, dtype=np.int32)
        s1 = np.ascontiguousarray(s1, dtype=np.float32)
        s2 = np.ascontiguousarray(s2, dtype=np.float32)
        self.blob = torch.empty(lib.hk_cbp_rect_plan_bytes(self.C1, self.C2, self.D), dtype=torch.uint8, device=device)
        with _on(device):
            check(lib.hk_cbp_rect_plan_build(h1.ctypes.data, s1.ctypes.data, self.C1, h2.ctypes.data, s2.ctypes.data, self.C2,
                                             self.D, ptr(self.blob), stream()), 'hk_cbp_rect_plan_build')


def _rect_shapes(x1, x2, plan, what):
    b, c1, h, w = x1.shape
    if x2.shape[0] != b or tuple(x2.shape[2:]) != (h, w) or (c1, x2.shape[1]) != (plan.C1, plan.C2):
        raise _lib.HawkeyeHipError(f'{what}: inputs {tuple(x1.shape)} / {tuple(x2.shape)}, plan for {plan.C1} x {plan.C2} channels')
    return b, c1, x2.shape[1], h, w


class _CbpRectSum(torch.autograd.Function):
    """_CbpCrossSum for two inputs of DIFFERENT widths: the C1 x C2 cross Gram on the generic MFMA tiles, the rect plan's
    gather over it; backward dG from dc, dX1 = dG X2, dX2 = dG^T X1."""

    @staticmethod
    def forward(ctx, x1, x2, plan):
        lib = _lib.load()
        x1, x2 = _f32c(x1), _f32c(x2)
        b, c1, c2, h, w = _rect_shapes(x1, x2, plan, 'compact_bilinear_pool')
        hw, d = h * w, plan.D
        g = torch.empty(b, c1, c2, dtype=torch.float32, device=x1.device)
        _bgemm_raw(lib, x1.view(b, c1, hw), x2.view(b, c2, hw), g, False, True, c1, c2, hw, b)
        c_raw = torch.empty(b, d, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_rect_bin_matrix(ptr(g), ptr(plan.blob), ptr(c_raw), b, c1, c2, d, stream()), 'hk_cbp_rect_bin_matrix')
        ctx.plan = plan
        ctx.save_for_backward(x1, x2)
        return c_raw

    @staticmethod
    def backward(ctx, dc):
        lib = _lib.load()
        x1, x2 = ctx.saved_tensors
        plan = ctx.plan
        b, c1, h, w = x1.shape
        c2, hw, d = x2.shape[1], h * w, plan.D
        dc = _f32c(dc)
        dg = torch.empty(b, c1, c2, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_rect_unbin_matrix(ptr(dc), ptr(plan.blob), ptr(dg), b, c1, c2, d, stream()), 'hk_cbp_rect_unbin_matrix')
        dx1 = dx2 = None
        if ctx.needs_input_grad[0]:
            dx1 = torch.empty_like(x1)
            _bgemm_raw(lib, dg, x2.view(b, c2, hw), dx1.view(b, c1, hw), False, False, c1, hw, c2, b)
        if ctx.needs_input_grad[1]:
            dx2 = torch.empty_like(x2)
            _bgemm_raw(lib, dg, x1.view(b, c1, hw), dx2.view(b, c2, hw), True, False, c2, hw, c1, b)
        return dx1, dx2, None


class _CbpRectPerLocation(torch.autograd.Function):
    """_CbpPerLocation for two inputs of different widths (sum_pool = False)."""

    @staticmethod
    def forward(ctx, x1, x2, plan):
        lib = _lib.load()
        x1, x2 = _f32c(x1), _f32c(x2)
        b, c1, c2, h, w = _rect_shapes(x1, x2, plan, 'compact_bilinear_pool')
        _loc_bwd_fits(ctx, c1, c2, plan.D)
        out = torch.empty(b, h, w, plan.D, dtype=torch.float32, device=x1.device)
        check(lib.hk_cbp_rect_loc_fwd(ptr(x1), ptr(x2), ptr(plan.blob), ptr(out), b, c1, c2, h * w, plan.D, stream()),
              'hk_cbp_rect_loc_fwd')
        ctx.plan = plan
        ctx.save_for_backward(x1, x2)
        return out

    @staticmethod
    def backward(ctx, dc):
        lib = _lib.load()
        x1, x2 = ctx.saved_tensors
        plan = ctx.plan
        b, c1, h, w = x1.shape
        dc = _f32c(dc)
        dx1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        check(lib.hk_cbp_rect_loc_bwd(ptr(x1), ptr(x2), ptr(dc), ptr(plan.blob), ptr(dx1), ptr(dx2), b, c1, x2.shape[1], h * w,
                                      plan.D, stream()), 'hk_cbp_rect_loc_bwd')
        return dx1, dx2, None


def compact_bilinear_sketch(x1, x2, plan, sum_pool=True):
    """The count sketch BEFORE the signed square root, for the forms Hawkeye's own CBCNN does not take: two different inputs
    ([B,D]) or no sum over the map ([B,H,W,D]); x2 may be x1.  A CbpRectPlan (input_dim1 != input_dim2) takes the C1 x C2 forms."""
    if isinstance(plan, CbpRectPlan):
        return _CbpRectSum.apply(x1, x2, plan) if sum_pool else _CbpRectPerLocation.apply(x1, x2, plan)
    return _CbpCrossSum.apply(x1, x2, plan) if sum_pool else _CbpPerLocation.apply(x1, x2, plan)


# --------------------------------------------------------------------- AP-CNN
class _AttPool(torch.autograd.Function):
    """gap = mean_hw F ; sgap = mean_hw a_s F  (one pass over F).
    replaces APCNN.py:256-266 + the AdaptiveAvgPool2d(1) heads :377-405,:533-538."""

    @staticmethod
    def forward(ctx, f, a_s):
        lib = _lib.load()
        f = _f32c(f)
        b, c, h, w = f.shape
        gap = torch.empty(b, c, dtype=torch.float32, device=f.device)
        if a_s is None:
            check(lib.hk_att_pool_fwd(ptr(f), None, ptr(gap), None, b, c, h * w, stream()), 'hk_att_pool_fwd')
            ctx.save_for_backward(f)
            ctx.has_att = False
            return gap, None
        a_s = _f32c(a_s)
        if a_s.numel() != b * h * w:
            raise _lib.HawkeyeHipError(f'att_pool: spatial attention must have {b} x {h} x {w} elements, got {tuple(a_s.shape)}')
        sgap = torch.empty(b, c, dtype=torch.float32, device=f.device)
        check(lib.hk_att_pool_fwd(ptr(f), ptr(a_s), ptr(gap), ptr(sgap), b, c, h * w, stream()), 'hk_att_pool_fwd')
        ctx.save_for_backward(f, a_s)
        ctx.has_att = True
        return gap, sgap

    @staticmethod
    def backward(ctx, dgap, dsgap):
        lib = _lib.load()
        f = ctx.saved_tensors[0]
        b, c, h, w = f.shape
        df = torch.empty_like(f)
        if dgap is None:
            dgap = torch.zeros(b, c, dtype=torch.float32, device=f.device)
        dgap = _f32c(dgap)
        if not ctx.has_att:
            check(lib.hk_att_pool_bwd(ptr(f), None, ptr(dgap), None, ptr(df), None, b, c, h * w, stream()),
                  'hk_att_pool_bwd')
            return df, None
        a_s = ctx.saved_tensors[1]
        if dsgap is None:
            dsgap = torch.zeros(b, c, dtype=torch.float32, device=f.device)
        dsgap = _f32c(dsgap)
        da = torch.empty_like(a_s)
        check(lib.hk_att_pool_bwd(ptr(f), ptr(a_s), ptr(dgap), ptr(dsgap), ptr(df), ptr(da), b, c, h * w,
                                  stream()), 'hk_att_pool_bwd')
        return df, da


def att_pool(f, a_s=None):
    """-> (gap [B,C], sgap [B,C] or None)"""
    return _AttPool.apply(f, a_s)


class _AttPool3(torch.autograd.Function):
    """The three pyramid levels of PyramidAttentions (APCNN.py:256-266) in one launch per direction:
    (f3, f4, f5, a3, a4, a5) -> gap [3,B,C], sgap [3,B,C]."""

    @staticmethod
    def forward(ctx, f0, f1, f2, a0, a1, a2):
        lib = _lib.load()
        fs = [_f32c(f) for f in (f0, f1, f2)]
        as_ = [_f32c(a) for a in (a0, a1, a2)]
        b, c = fs[0].shape[:2]
        hws = [f.shape[2] * f.shape[3] for f in fs]
        for f, a, hw in zip(fs, as_, hws):
            if f.shape[:2] != (b, c) or a.numel() != b * hw:
                raise _lib.HawkeyeHipError(f'att_pool_levels: level shapes {tuple(f.shape)} / {tuple(a.shape)} do not match [{b},{c},h,w] / [{b},1,h,w]')
        gap = torch.empty(3, b, c, dtype=torch.float32, device=fs[0].device)
        sgap = torch.empty_like(gap)
        check(lib.hk_att_pool3_fwd(ptr(fs[0]), ptr(fs[1]), ptr(fs[2]), ptr(as_[0]), ptr(as_[1]), ptr(as_[2]), ptr(gap), ptr(sgap),
                                   b, c, hws[0], hws[1], hws[2], stream()), 'hk_att_pool3_fwd')
        ctx.save_for_backward(*fs, *as_)
        return gap, sgap

    @staticmethod
    def backward(ctx, dgap, dsgap):
        lib = _lib.load()
        fs, as_ = ctx.saved_tensors[:3], ctx.saved_tensors[3:]
        b, c = fs[0].shape[:2]
        hws = [f.shape[2] * f.shape[3] for f in fs]
        dgap = _f32c(dgap) if dgap is not None else torch.zeros(3, b, c, dtype=torch.float32, device=fs[0].device)
        dsgap = _f32c(dsgap) if dsgap is not None else torch.zeros(3, b, c, dtype=torch.float32, device=fs[0].device)
        dfs = [torch.empty_like(f) for f in fs]
        das = [torch.empty_like(a) for a in as_]
        check(lib.hk_att_pool3_bwd(ptr(fs[0]), ptr(fs[1]), ptr(fs[2]), ptr(as_[0]), ptr(as_[1]), ptr(as_[2]), ptr(dgap), ptr(dsgap),
                                   ptr(dfs[0]), ptr(dfs[1]), ptr(dfs[2]), ptr(das[0]), ptr(das[1]), ptr(das[2]),
                                   b, c, hws[0], hws[1], hws[2], stream()), 'hk_att_pool3_bwd')
        return (*dfs, *das)


def att_pool_levels(feats, atts):
    """feats: three [B,C,h,w] pyramid levels, atts: their spatial attentions [B,1,h,w] -> (gap [3,B,C], sgap [3,B,C])."""
    return _AttPool3.apply(*feats, *atts)


def att_roi_select(att_mask, feature_stride, anchor_size, img_h, img_w, num_classes, iou_thred, topk):
    """Device-side get_att_roi (APCNN.py:444-476).  -> (rois [B,topk,5], count [B] int32),
    rows beyond count are zero.  No gradient (reference: torch.no_grad, :447)."""
    lib = _lib.load()
    with torch.no_grad():
        a = _f32c(att_mask.detach())
        n, _, h, w = a.shape
        lo, hi = (0.2, 0.8) if num_classes == 200 else (0.1, 0.9)     # APCNN.py:451-455
        r0, r1, c0, c1 = int(lo * h), int(hi * h), int(lo * w), int(hi * w)
        rois = torch.empty(n, topk, 5, dtype=torch.float32, device=a.device)
        cnt = torch.empty(n, dtype=torch.int32, device=a.device)
        check(lib.hk_att_roi_select(ptr(a), ptr(rois), ptr(cnt), n, h, w, int(feature_stride), float(anchor_size),
                                    int(img_h), int(img_w), r0, r1, c0, c1, float(iou_thred), int(topk), stream()),
              'hk_att_roi_select')
    return rois, cnt


def att_roi_select_levels(att_masks, levels, img_h, img_w, num_classes, iou_thred):
    """The three pyramid levels of one forward in one launch (hk_att_roi_select3): att_masks = (a3, a4, a5),
    levels = ((feature_stride, anchor_size, topk), ...).  -> [(rois, count)] * 3, the results of three att_roi_select
    calls bit for bit."""
    import ctypes
    lib = _lib.load()
    assert len(att_masks) == 3 and len(levels) == 3
    with torch.no_grad():
        a = [_f32c(m.detach()) for m in att_masks]
        n = a[0].shape[0]
        lo, hi = (0.2, 0.8) if num_classes == 200 else (0.1, 0.9)     # APCNN.py:451-455
        hs, ws = [m.shape[2] for m in a], [m.shape[3] for m in a]
        keep = []
        for h, w in zip(hs, ws):
            keep += [int(lo * h), int(hi * h), int(lo * w), int(hi * w)]
        rois = [torch.empty(n, int(k), 5, dtype=torch.float32, device=a[0].device) for _, _, k in levels]
        cnt = [torch.empty(n, dtype=torch.int32, device=a[0].device) for _ in levels]
        P3, I3, F3, I12 = ctypes.c_void_p * 3, ctypes.c_int * 3, ctypes.c_float * 3, ctypes.c_int * 12
        arrs = (P3(*[ptr(m) for m in a]), P3(*[ptr(r) for r in rois]), P3(*[ptr(c) for c in cnt]), I3(*hs), I3(*ws),
                I3(*[int(s) for s, _, _ in levels]), F3(*[float(z) for _, z, _ in levels]), I12(*keep),
                I3(*[int(k) for _, _, k in levels]))               # host arrays, read by the entry point before it returns
        ad = [ctypes.addressof(x) for x in arrs]
        check(lib.hk_att_roi_select3(ad[0], ad[1], ad[2], n, ad[3], ad[4], ad[5], ad[6], int(img_h), int(img_w), ad[7],
                                     float(iou_thred), ad[8], stream()), 'hk_att_roi_select3')
    return list(zip(rois, cnt))


def roi_boxes(tables, u01, scale):
    """Union / drop boxes on device from three (rois, count) tables.  -> box [B,4], drop [B,4]"""
    lib = _lib.load()
    (r3, n3), (r4, n4), (r5, n5) = tables
    b = r3.shape[0]
    box = torch.empty(b, 4, dtype=torch.float32, device=r3.device)
    drop = torch.empty(b, 4, dtype=torch.float32, device=r3.device)
    check(lib.hk_roi_boxes(ptr(r3), ptr(n3), r3.shape[1], ptr(r4), ptr(n4), r4.shape[1], ptr(r5), ptr(n5),
                           r5.shape[1], ptr(u01) if u01 is not None else None, float(scale), ptr(box), ptr(drop),
                           b, stream()), 'hk_roi_boxes')
    return box, drop


class _RoiCropResize(torch.autograd.Function):
    """replaces get_roi_crop_feat's per-image python loop, APCNN.py:478-531."""

    @staticmethod
    def forward(ctx, x, box, drop, training):
        lib = _lib.load()
        x = _f32c(x)
        b, c, h, w = x.shape
        box, drop = _f32c(box), _f32c(drop)
        if tuple(box.shape) != (b, 4) or tuple(drop.shape) != (b, 4):
            raise _lib.HawkeyeHipError(f'roi_crop_resize: box and drop must be [{b}, 4], got {tuple(box.shape)} and {tuple(drop.shape)}')
        y = torch.empty_like(x)
        check(lib.hk_roi_crop_resize_fwd(ptr(x), ptr(box), ptr(drop), ptr(y), b, c, h, w, int(training), stream()),
              'hk_roi_crop_resize_fwd')
        ctx.save_for_backward(box, drop)
        ctx.training = int(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        box, drop = ctx.saved_tensors
        dy = _f32c(dy)
        b, c, h, w = dy.shape
        dx = torch.empty_like(dy)
        check(lib.hk_roi_crop_resize_bwd(ptr(dy), ptr(box), ptr(drop), ptr(dx), b, c, h, w, ctx.training, stream()),
              'hk_roi_crop_resize_bwd')
        return dx, None, None, None


def roi_crop_resize(x, box, drop, training):
    return _RoiCropResize.apply(x, box, drop, training)


# --------------------------------------------------------------------- OSME
class _OsmeGap(torch.autograd.Function):
    """z = GAP(x).  replaces OSME.py:21 (avg_pool + squeeze)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        n, c, h, w = x.shape
        z = torch.empty(n, c, dtype=torch.float32, device=x.device)
        check(lib.hk_osme_gap(ptr(x), ptr(z), n, c, h * w, stream()), 'hk_osme_gap')
        ctx.shape = x.shape
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        n, c, h, w = ctx.shape
        dz = _f32c(dz)
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dz.device)
        # dx[b,c,:] = dz[b,c] / HW: the row-parallel GAP backward of the attention pooling (one launch; without a gate
        # the entry point takes no F)
        check(lib.hk_att_pool_bwd(None, None, ptr(dz), None, ptr(dx), None, n, c, h * w, stream()), 'hk_att_pool_bwd')
        return dx


class _OsmeScale(torch.autograd.Function):
    """s[p] = m[p] (.) x for all P gates in one pass.  replaces OSME.py:23."""

    @staticmethod
    def forward(ctx, x, m):
        lib = _lib.load()
        x, m = _f32c(x), _f32c(m)
        n, c, h, w = x.shape
        if m.dim() != 3 or tuple(m.shape[1:]) != (n, c):
            raise _lib.HawkeyeHipError(f'osme_scale: gates must be [P, N, C] = [P, {n}, {c}], got {tuple(m.shape)}')
        p = m.shape[0]
        s = torch.empty(p, n, c, h, w, dtype=torch.float32, device=x.device)
        check(lib.hk_osme_scale_fwd(ptr(x), ptr(m), ptr(s), p, n, c, h * w, stream()), 'hk_osme_scale_fwd')
        ctx.save_for_backward(x, m)
        return s

    @staticmethod
    def backward(ctx, ds):
        lib = _lib.load()
        x, m = ctx.saved_tensors
        n, c, h, w = x.shape
        p = m.shape[0]
        ds = _f32c(ds)
        dx = torch.empty_like(x)
        dm = torch.empty_like(m)
        check(lib.hk_osme_scale_bwd(ptr(x), ptr(m), ptr(ds), None, ptr(dx), ptr(dm), p, n, c, h * w, stream()),
              'hk_osme_scale_bwd')
        return dx, dm


def osme_gap(x):
    return _OsmeGap.apply(x)


def osme_scale(x, m):
    """x [N,C,H,W], m [P,N,C] -> s [P,N,C,H,W]"""
    return _OsmeScale.apply(x, m)


# --------------------------------------------------------------------- generic
# --------------------------------------------------------------------- CIN channel interaction
# hk_cin_sci_fwd: for C % 64 == 0 and 7x7 / 8x8 / 6x6 maps ONE kernel (cin.hip: two passes over the column blocks, softmax
# statistics first, then W written once and consumed from registers by the second product) - 297 us at the plugin's shape
# (B = 20, C = 2048, HW = 49) against 486 us for rocBLAS bmm + softmax + bmm.  14x14 / 12x12 / 10x10 maps (a 448^2 input):
# the scores are materialised by the Gram panel kernel, then row statistics, then ONE kernel that applies the softmax on the
# way into the second product and writes W once, in place (631 us at B = 20, C = 2048, HW = 196; library 937).  Other shapes
# run the three-kernel chain on the generic MFMA tile inside the same entry point - there is no library branch.


class _CinSci(torch.autograd.Function):
    """W = softmax_rows(-X X^T / HW), Y = W X.  replaces model/methods/CIN.py:31-34.  W is an output too: the
    contrastive branch (cin_cci) consumes it and sends a gradient back into it."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x)
        b, c, hw = x.shape
        w = torch.empty(b, c, c, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check(lib.hk_cin_sci_fwd(ptr(x), ptr(w), ptr(y), b, c, hw, stream()), 'hk_cin_sci_fwd')
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)
        return y, w

    @staticmethod
    def backward(ctx, dy, dw):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        b, c, hw = x.shape
        dy = _f32c(dy) if dy is not None else torch.zeros_like(x)
        dwbuf = _f32c(dw).clone() if dw is not None else torch.empty_like(w)     # overwritten by the kernel chain
        dx = torch.empty_like(x)
        check(lib.hk_cin_sci_bwd(ptr(x), ptr(w), ptr(dy), ptr(dwbuf), int(dw is not None), ptr(dx), b, c, hw, stream()),
              'hk_cin_sci_bwd')
        return dx


class _CinCci(torch.autograd.Function):
    """Yc[b] = |W[b] - w_b W[(b + B/2) % B]| X[b].  replaces model/methods/CIN.py:51-54."""

    @staticmethod
    def forward(ctx, w, x, wt):
        lib = _lib.load()
        w, x, wt = _f32c(w), _f32c(x), _f32c(wt)
        b, c, hw = x.shape
        if b % 2 or wt.shape != (b,):
            raise _lib.HawkeyeHipError(f'cin_cci: batch {b} must be even and weights [B], got {tuple(wt.shape)}')
        y = torch.empty_like(x)
        check(lib.hk_cin_cci_fwd(ptr(x), ptr(w), ptr(wt), ptr(y), b, c, hw, stream()), 'hk_cin_cci_fwd')
        ctx.save_for_backward(w, x, wt)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        w, x, wt = ctx.saved_tensors
        b, c, hw = x.shape
        dy = _f32c(dy)
        dx, dw, dwt = torch.empty_like(x), torch.empty_like(w), torch.empty_like(wt)
        nws = lib.hk_cin_cci_ws_bytes(b, c)
        ws = _ws(nws, x.device)
        check(lib.hk_cin_cci_bwd(ptr(x), ptr(w), ptr(wt), ptr(dy), ptr(dx), ptr(dw), ptr(dwt), b, c, hw, ptr(ws), nws,
                                 stream()), 'hk_cin_cci_bwd')
        return dw, dx, dwt


def cin_sci(x):
    """x [B,C,HW] -> (Y [B,C,HW], W_SCI [B,C,C])."""
    return _CinSci.apply(x)


def cin_cci(w_sci, x, weight):
    """W_SCI [B,C,C], x [B,C,HW], weight [B] (eta for the first half of the batch, gamma for the second) -> Y_CCI."""
    return _CinCci.apply(w_sci, x, weight)


# --------------------------------------------------------------------- MAMC n-pairs loss
class _NPairsLoss(torch.autograd.Function):
    """replaces NPairsLoss.forward, model/loss/MAMC_loss.py:34-90.  The kernel returns the loss and its gradient
    with respect to the parts in one pass; backward only scales it."""

    @staticmethod
    def forward(ctx, parts, targets):
        lib = _lib.load()
        parts = _f32c(parts)
        b, p, d = parts.shape
        labels = targets.to(device=parts.device, dtype=torch.int32).contiguous()
        if labels.shape != (b,):
            raise _lib.HawkeyeHipError(f'npairs_loss: {b} samples but targets of shape {tuple(targets.shape)}')
        loss = torch.empty(1, dtype=torch.float32, device=parts.device)
        dx = torch.empty_like(parts)
        nws = lib.hk_npairs_ws_bytes(b * p, d)
        ws = _ws(nws, parts.device)
        check(lib.hk_npairs_loss(ptr(parts), ptr(labels), ptr(loss), ptr(dx), b, p, d, ptr(ws), nws, stream()),
              'hk_npairs_loss')
        ctx.save_for_backward(dx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def npairs_loss(parts, targets):
    """parts [B,P,D] (OSMENet's second output), targets [B] -> scalar n-pairs loss (MAMC eq. 11)."""
    return _NPairsLoss.apply(parts, targets)


# --------------------------------------------------------------------- classifier
class _Linear(torch.autograd.Function):
    """replaces nn.Linear on the pooled vector (model/methods/BCNN.py:42,54 and the other heads' classifiers)."""

    @staticmethod
    def forward(ctx, y, weight, bias):
        lib = _lib.load()
        y, weight = _f32c(y), _f32c(weight)
        b, j = y.shape
        k = weight.shape[0]
        if weight.shape[1] != j:
            raise _lib.HawkeyeHipError(f'linear: weight {tuple(weight.shape)} does not match input {tuple(y.shape)}')
        bias_c = _f32c(bias) if bias is not None else None
        out = torch.empty(b, k, dtype=torch.float32, device=y.device)
        nws = lib.hk_linear_ws_bytes(b, j, k)
        ws = _ws(nws, y.device)
        check(lib.hk_linear_fwd(ptr(y), ptr(weight), ptr(bias_c), ptr(out), b, j, k, ptr(ws), nws, stream()),
              'hk_linear_fwd')
        ctx.save_for_backward(y, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, weight = ctx.saved_tensors
        g = _f32c(g)
        b, j = y.shape
        k = weight.shape[0]
        dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(k, dtype=torch.float32, device=y.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        check(lib.hk_linear_bwd(ptr(y), ptr(weight), ptr(g), ptr(dy), ptr(dw), ptr(db), b, j, k, stream()),
              'hk_linear_bwd')
        return dy, dw, db


class _SsqrtPoolLinear(torch.autograd.Function):
    """Signed-sqrt bilinear pooling (the reference's commented alternative, BCNN.py:23-24) + the classifier on it with the
    l2 scale folded into the classifier: the pooled vector is handed over as u = sign(G) sqrt(|G| + 1e-10),
    unnormalised - the scale pass over B x C^2 floats is not launched - and logits = inv_norm[b] (u W^T) + bias."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, weight = _f32c(x), _f32c(weight)
        b, c, h, w = x.shape
        hw, j, k = h * w, c * c, weight.shape[0]
        if weight.shape[1] != j:
            raise _lib.HawkeyeHipError(f'ssqrt_pool_linear: weight {tuple(weight.shape)} does not match {c} x {c} pooled features')
        u = torch.empty(b, j, dtype=torch.float32, device=x.device)
        inv_norm = torch.empty(b, dtype=torch.float32, device=x.device)
        nws = lib.hk_bcnn_ssqrt_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        # the kernels write the PRE-BIAS product `pre` = inv_norm (u W^T); it stays private to this node (the backward's
        # <y, dy> = sum_k g_k pre_k needs it exact, whatever the bias is and whatever the caller does to the logits in place)
        pre = torch.empty(b, k, dtype=torch.float32, device=x.device)
        nwl = lib.hk_linear_ws_bytes(b, j, k)
        wsl = _ws(nwl, x.device)
        # two launches + the classifier's reduce: the Gram kernel leaves partial sums of u^2 and the reduce launch forms 1 / |u|
        import ctypes
        nparts = ctypes.c_int(0)
        rc = lib.hk_bcnn_ssqrt_pool_fwd_parts(ptr(x), ptr(u), ptr(ws), ctypes.byref(nparts), b, c, hw, stream())
        if rc == _lib.HK_OK:
            check(lib.hk_linear_fwd_ssq(ptr(u), ptr(weight), None, ptr(ws), nparts.value, ptr(inv_norm), ptr(pre), b, j, k,
                                        ptr(wsl), nwl, stream()), 'hk_linear_fwd_ssq')
        else:
            if rc != _lib.HK_ERR_UNSUPPORTED:
                check(rc, 'hk_bcnn_ssqrt_pool_fwd_parts')
            check(lib.hk_bcnn_ssqrt_pool_fwd_unscaled(ptr(x), ptr(u), ptr(inv_norm), b, c, hw, ptr(ws), nws, stream()),
                  'hk_bcnn_ssqrt_pool_fwd_unscaled')
            check(lib.hk_linear_fwd_scaled(ptr(u), ptr(weight), None, ptr(inv_norm), ptr(pre), b, j, k, ptr(wsl), nwl,
                                           stream()), 'hk_linear_fwd_scaled')
        ctx.save_for_backward(x, u, inv_norm, weight, pre)
        ctx.has_bias = bias is not None
        return pre + _f32c(bias) if bias is not None else pre.clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, u, inv_norm, weight, pre = ctx.saved_tensors
        g = _f32c(g)
        b, c, h, w = x.shape
        hw, j, k = h * w, c * c, weight.shape[0]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dy = torch.empty_like(u) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(k, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        rc = lib.hk_linear_bwd_scaled(ptr(u), ptr(weight), ptr(g), ptr(inv_norm), ptr(dy), ptr(dw), ptr(db), b, j, k, stream())
        if rc == _lib.HK_ERR_UNSUPPORTED:      # shapes the one-launch kernel does not serve: dW from the scaled g explicitly
            gs = g * inv_norm[:, None]
            check(lib.hk_linear_bwd(ptr(u), ptr(weight), ptr(g), ptr(dy), None, ptr(db), b, j, k, stream()), 'hk_linear_bwd')
            if need_w:
                check(lib.hk_linear_bwd(ptr(u), ptr(weight), ptr(gs), None, ptr(dw), None, b, j, k, stream()), 'hk_linear_bwd')
        else:
            check(rc, 'hk_linear_bwd_scaled')
        dx = None
        if need_x:
            dx = torch.empty_like(x)
            nws = lib.hk_bcnn_ssqrt_ws_bytes(b, c, hw)
            ws = _ws(nws, x.device)
            # <y, dy> = sum_k g_k pre_k (the pre-bias product): the classifier's operands instead of a pass over u and dy
            check(lib.hk_bcnn_ssqrt_pool_bwd_tdot(ptr(x), ptr(u), ptr(dy), ptr(inv_norm), ptr(g), ptr(pre), None, k, 1, ptr(dx),
                                                  b, c, hw, ptr(ws), nws, stream()), 'hk_bcnn_ssqrt_pool_bwd_tdot')
        return dx, dw, db


class _BilinearPoolLinear(torch.autograd.Function):
    """BilinearPooling + the classifier on it (model/methods/BCNN.py:13-27 + :54) as ONE autograd node.  Forward: the two
    entry points back to back (the l2 scale is already free in the Gram epilogue: nothing to fold).  Backward: because
    dy = g W is produced here, the inner product <y, dy> that F.normalize's backward needs is known in closed form,
        t[b] = sum_j y[b,j] sum_k g[b,k] W[k,j] = sum_k g[b,k] (y W^T)[b,k]      (the pre-bias product, kept by this node),
    so hk_bcnn_pool_bwd_tdot runs the Gram backward as one launch with the rank-1 term applied while dX is written -
    no partial sums of y * dy, no second pass over dX."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, weight = _f32c(x), _f32c(weight)
        b, c, h, w = x.shape
        hw, j, k = h * w, c * c, weight.shape[0]
        if weight.shape[1] != j:
            raise _lib.HawkeyeHipError(f'bilinear_pool_linear: weight {tuple(weight.shape)} does not match {c} x {c} pooled features')
        y = torch.empty(b, j, dtype=torch.float32, device=x.device)
        inv_norm = torch.empty(b, dtype=torch.float32, device=x.device)
        colsum = torch.empty(b, hw, dtype=torch.float32, device=x.device)
        nws = lib.hk_bcnn_pool_ws_bytes(b, c, hw)
        ws = _ws(nws, x.device)
        check(lib.hk_bcnn_pool_fwd(ptr(x), ptr(y), ptr(inv_norm), ptr(colsum), b, c, hw, ptr(ws), nws, stream()),
              'hk_bcnn_pool_fwd')
        # the kernel writes the PRE-BIAS product y W^T, which stays private to this node: the backward's closed form
        # t = sum_k g_k (y W^T)_k then neither cancels against a large bias nor breaks when the caller modifies the logits in
        # place (nn.Linear does not save its output either); the bias is added by one [B,K] elementwise op - the same bits
        # as the kernel's own `sum + bias`
        pre = torch.empty(b, k, dtype=torch.float32, device=x.device)
        nwl = lib.hk_linear_ws_bytes(b, j, k)
        wsl = _ws(nwl, x.device)
        check(lib.hk_linear_fwd(ptr(y), ptr(weight), None, ptr(pre), b, j, k, ptr(wsl), nwl, stream()), 'hk_linear_fwd')
        ctx.save_for_backward(x, y, inv_norm, colsum, weight, pre)
        ctx.has_bias = bias is not None
        return pre + _f32c(bias) if bias is not None else pre.clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, y, inv_norm, colsum, weight, pre = ctx.saved_tensors
        g = _f32c(g)
        b, c, h, w = x.shape
        hw, j, k = h * w, c * c, weight.shape[0]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dy = torch.empty_like(y) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(k, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if dy is not None or dw is not None or db is not None:
            check(lib.hk_linear_bwd(ptr(y), ptr(weight), ptr(g), ptr(dy), ptr(dw), ptr(db), b, j, k, stream()), 'hk_linear_bwd')
        dx = None
        if need_x:
            dx = torch.empty_like(x)
            nws = lib.hk_bcnn_pool_ws_bytes(b, c, hw)
            ws = _ws(nws, x.device)
            check(lib.hk_bcnn_pool_bwd_tdot(ptr(x), ptr(y), ptr(dy), ptr(inv_norm), ptr(colsum), ptr(g), ptr(pre), None, k,
                                            ptr(dx), b, c, hw, ptr(ws), nws, stream()), 'hk_bcnn_pool_bwd_tdot')
        return dx, dw, db


def bilinear_pool_linear(x, weight, bias=None):
    """x [B,C,h,w] -> logits [B,K] = Linear(BilinearPooling(x)) (BCNN.py:13-27,54) as one node: see _BilinearPoolLinear."""
    return _BilinearPoolLinear.apply(x, weight, bias)


def ssqrt_pool_linear(x, weight, bias=None):
    """x [B,C,h,w] -> logits [B,K] = Linear(normalize(sign(G) sqrt(|G| + 1e-10))) with the normalisation folded into the
    classifier's epilogue (the pooled vector is never rescaled in memory)."""
    return _SsqrtPoolLinear.apply(x, weight, bias)


def linear(y, weight, bias=None):
    """y [B,J] @ weight[K,J]^T + bias[K] -> [B,K] on the split-K f32-MFMA path."""
    return _Linear.apply(y, weight, bias)


# --------------------------------------------------------------------- VGG trunk epilogues
def _nhwc(t, what):
    """A [N,C,H,W] fp32 tensor whose memory is NHWC (channels_last) - the layout the trunk runs in."""
    if t.dtype != torch.float32:
        raise _lib.HawkeyeHipError(f'{what}: fp32 only, got {t.dtype}')
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise _lib.HawkeyeHipError(f'{what}: needs a channels_last [N,C,H,W] tensor, got shape {tuple(t.shape)} strides {t.stride()}')
    return t


def trunk_epilogue_ok(x, pool=False):
    """Whether hk_bias_relu_* serve this convolution output: an fp32 channels_last map on a HIP device, C / 4 a divisor of 256
    (VGG: 64 .. 512), even H and W for the pooled form.  Anything else stays on the framework's own ops (model/backbone/vgg.py)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    n, c, h, w = x.shape
    ok = n > 0 and c % 4 == 0 and c // 4 <= 256 and 256 % (c // 4) == 0 and x.is_contiguous(memory_format=torch.channels_last)
    return ok and (not pool or (h % 2 == 0 and w % 2 == 0 and h > 0 and w > 0)) and x.data_ptr() % 16 == 0


class _BiasReLU(torch.autograd.Function):
    """y = relu(conv_out + bias) IN PLACE on the convolution's output; backward: dx = dy where y > 0, dbias = sum dx in one
    pass.  replaces nn.Conv2d's bias add + nn.ReLU(inplace=True) of the VGG trunk (model/backbone/vgg.py:24-57)."""

    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        _nhwc(x, 'bias_relu')
        n, c, h, w = x.shape
        b = _f32c(bias)
        # a pass that will run backward keeps the SIGN of the output as one byte per channel quad (1/16 of the map): the backward
        # then reads that instead of the map (which stays alive anyway - it is the next convolution's input)
        mask = torch.empty(n, h, w, c // 4, dtype=torch.uint8, device=x.device) if any(ctx.needs_input_grad) else None
        check(lib.hk_bias_relu_fwd(ptr(x), ptr(b), ptr(mask), n * h * w, c, stream()), 'hk_bias_relu_fwd')
        ctx.mark_dirty(x)
        ctx.save_for_backward(mask)
        ctx.in_shape = (n, c, h, w)
        return x

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        n, c, h, w = ctx.in_shape
        if dy.dtype != torch.float32:
            raise _lib.HawkeyeHipError(f'bias_relu backward: fp32 only, got {dy.dtype}')
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(dy, memory_format=torch.channels_last)
        db = torch.empty(c, dtype=torch.float32, device=dy.device)
        nws = lib.hk_trunk_ws_bytes(c)
        ws = _ws(nws, dy.device)
        check(lib.hk_bias_relu_bwd(ptr(dy), None, ptr(mask), ptr(dx), ptr(db), n * h * w, c, ptr(ws), nws, stream()), 'hk_bias_relu_bwd')
        return dx, db


class _BiasReLUPool(torch.autograd.Function):
    """p = maxpool2x2(relu(conv_out + bias)); the full-resolution activation is never written - a 2-bit argmax per element is
    kept instead - and the backward routes dp through pool, ReLU and the bias sum in one pass.  replaces the bias add +
    nn.ReLU + nn.MaxPool2d(2, 2) behind the last convolution of each VGG stage (model/backbone/vgg.py:24-57)."""

    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        _nhwc(x, 'bias_relu_pool')
        n, c, h, w = x.shape
        b = _f32c(bias)
        p = torch.empty(n, c, h // 2, w // 2, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        am = torch.empty(n, h // 2, w // 2, c // 4, dtype=torch.uint8, device=x.device)
        check(lib.hk_bias_relu_pool_fwd(ptr(x), ptr(b), ptr(p), ptr(am), n, h, w, c, stream()), 'hk_bias_relu_pool_fwd')
        ctx.save_for_backward(p, am)
        ctx.in_shape = (n, c, h, w)
        return p

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        p, am = ctx.saved_tensors
        n, c, h, w = ctx.in_shape
        if dp.dtype != torch.float32:
            raise _lib.HawkeyeHipError(f'bias_relu_pool backward: fp32 only, got {dp.dtype}')
        dp = dp.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=p.device, memory_format=torch.channels_last)
        db = torch.empty(c, dtype=torch.float32, device=p.device)
        nws = lib.hk_trunk_ws_bytes(c)
        ws = _ws(nws, p.device)
        check(lib.hk_bias_relu_pool_bwd(ptr(dp), ptr(p), ptr(am), ptr(dx), ptr(db), n, h, w, c, ptr(ws), nws, stream()),
              'hk_bias_relu_pool_bwd')
        return dx, db


def _same_dense_layout(a, b):
    return (a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype == torch.float32 and a.numel() % 4 == 0 and a.numel() > 0
            and (a.is_contiguous() or (a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last))))


def add_relu_ok(a, b):
    """Whether hk_add_relu_fwd serves `relu_(a.add_(b))`: two fp32 HIP tensors of one dense layout."""
    return torch.is_tensor(a) and torch.is_tensor(b) and a.is_cuda and b.is_cuda and _same_dense_layout(a, b) and \
        a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0


class _AddReLU(torch.autograd.Function):
    """a = relu(a + b) in place on a (the residual sum that ends a ResNet bottleneck, model/backbone/resnet.py:89-136):
    one pass instead of add_ + relu_; backward: one masked copy that is the gradient of both operands."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        if not _same_dense_layout(a, b):
            raise _lib.HawkeyeHipError(f'add_relu: operands must share one dense fp32 layout, got {tuple(a.shape)} {a.stride()} / '
                                       f'{tuple(b.shape)} {b.stride()}')
        check(lib.hk_add_relu_fwd(ptr(a), ptr(b), a.numel(), stream()), 'hk_add_relu_fwd')
        ctx.mark_dirty(a)
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        if dy.dtype != torch.float32:
            raise _lib.HawkeyeHipError(f'add_relu backward: fp32 only, got {dy.dtype}')
        if dy.stride() != y.stride():
            dy = dy.contiguous(memory_format=torch.channels_last) if (y.dim() == 4 and not y.is_contiguous()) else dy.contiguous()
        g = torch.empty_like(y)
        check(lib.hk_relu_mask_bwd(ptr(dy), ptr(y), ptr(g), y.numel(), stream()), 'hk_relu_mask_bwd')
        return g, g


def add_relu(a, b):
    """relu(a + b) in place on a; see _AddReLU."""
    return _AddReLU.apply(a, b)


def conv1_ok(x, conv):
    """Whether hk_conv1_bias_relu_* serve this convolution: the trunk's first layer - Conv2d(Cin <= 3, 64, 3, stride 1, padding 1)
    with a bias on an fp32 channels_last HIP batch that needs no gradient itself."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not x.requires_grad):
        return False
    two = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    return (x.shape[1] <= 3 and conv.in_channels == x.shape[1] and conv.out_channels == 64 and conv.bias is not None
            and two(conv.kernel_size) == (3, 3) and two(conv.stride) == (1, 1) and two(conv.padding) == (1, 1)
            and two(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0)


class _Conv1BiasReLU(torch.autograd.Function):
    """relu(conv2d(x, weight, bias, padding=1)) for the trunk's first layer (Cin <= 3 -> 64 channels) in one kernel per
    direction (Cin <= 3): the output map is written once (with its sign mask), and the backward forms dW and dbias straight from the
    output's gradient without writing the masked gradient map.  replaces the library convolution + its bias / ReLU passes
    (model/backbone/vgg.py:24-57, `features[0:2]`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _nhwc(x, 'conv1_bias_relu')
        n, cin, h, w = x.shape
        if tuple(weight.shape) != (64, cin, 3, 3):
            raise _lib.HawkeyeHipError(f'conv1_bias_relu: weight {tuple(weight.shape)} does not match [64, {cin}, 3, 3]')
        wt = weight.detach().permute(2, 3, 1, 0).reshape(9 * cin, 64).contiguous()          # tap-major: (kh, kw, c) x out
        b = _f32c(bias.detach())
        y = torch.empty(n, 64, h, w, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        mask = torch.empty(n, h, w, 16, dtype=torch.uint8, device=x.device) if any(ctx.needs_input_grad[1:]) else None
        check(lib.hk_conv1_bias_relu_fwd(ptr(x), ptr(wt), ptr(b), ptr(y), ptr(mask), n, h, w, cin, 64, stream()), 'hk_conv1_bias_relu_fwd')
        ctx.save_for_backward(x, mask)
        ctx.wfmt = weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous()
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mask = ctx.saved_tensors
        n, cin, h, w = x.shape
        if dy.dtype != torch.float32:
            raise _lib.HawkeyeHipError(f'conv1_bias_relu backward: fp32 only, got {dy.dtype}')
        dy = dy.contiguous(memory_format=torch.channels_last)
        dwt = torch.empty(9 * cin, 64, dtype=torch.float32, device=x.device)
        db = torch.empty(64, dtype=torch.float32, device=x.device)
        nws = lib.hk_conv1_ws_bytes(cin)
        ws = _ws(nws, x.device)
        check(lib.hk_conv1_bias_relu_bwd(ptr(dy), ptr(mask), ptr(x), ptr(dwt), ptr(db), n, h, w, cin, 64, ptr(ws), nws, stream()),
              'hk_conv1_bias_relu_bwd')
        dw = dwt.view(3, 3, cin, 64).permute(3, 2, 0, 1)                                     # [64, Cin, 3, 3]
        dw = dw.contiguous(memory_format=torch.channels_last) if ctx.wfmt else dw.contiguous()
        return None, dw, db


def conv1_bias_relu(x, weight, bias):
    """relu(conv2d(x, weight, bias, stride=1, padding=1)) for Cin <= 3 -> 64 channels; see _Conv1BiasReLU."""
    return _Conv1BiasReLU.apply(x, weight, bias)


def bias_relu(x, bias):
    """relu(x + bias[None,:,None,None]) in place on x (a channels_last convolution output); see _BiasReLU."""
    return _BiasReLU.apply(x, bias)


def bias_relu_pool(x, bias):
    """max_pool2d(relu(x + bias[None,:,None,None]), 2, 2) for a channels_last convolution output; see _BiasReLUPool."""
    return _BiasReLUPool.apply(x, bias)


# --------------------------------------------------------------------- input finalisation
def image_finalize(u8, erase=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), channels_last=False):
    """uint8 [B,H,W,3] on the device (+ int32 erase boxes [B,4] = top, left, h, w) -> normalised fp32 [B,3,H,W]
    (channels_last storage if asked).  No gradient: it is the end of the data pipeline."""
    import ctypes
    lib = _lib.load()
    if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3:
        raise _lib.HawkeyeHipError(f'image_finalize wants uint8 [B,H,W,3], got {u8.dtype} {tuple(u8.shape)}')
    u8 = u8.contiguous()
    b, h, w, _ = u8.shape
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out = torch.empty(b, 3, h, w, dtype=torch.float32, device=u8.device).contiguous(memory_format=fmt)
    if erase is not None:
        erase = erase.to(device=u8.device, dtype=torch.int32).contiguous()
        if erase.shape != (b, 4):
            raise _lib.HawkeyeHipError(f'image_finalize: erase boxes must be [B,4], got {tuple(erase.shape)}')
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib.hk_image_finalize(ptr(u8), ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), ptr(erase),
                                ptr(out.permute(0, 2, 3, 1) if channels_last else out), b, h, w, int(channels_last),
                                stream()), 'hk_image_finalize')
    return out


def bgemm(a, b, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, diag=0.0, out=None):
    """Batched fp32 MFMA GEMM (test / composition helper).  a [B,M,K] (or [B,K,M]), b [B,K,N] (or [B,N,K])."""
    lib = _lib.load()
    a, b = _f32c(a), _f32c(b)
    nb = a.shape[0]
    m, k = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    n = b.shape[1] if trans_b else b.shape[2]
    if out is None:
        out = torch.zeros(nb, m, n, dtype=torch.float32, device=a.device)
    check(lib.hk_bgemm_f32(ptr(a), a.shape[2], a.shape[1] * a.shape[2], int(trans_a),
                           ptr(b), b.shape[2], b.shape[1] * b.shape[2], int(trans_b),
                           ptr(out), n, m * n, m, n, k, nb, float(alpha), float(beta), float(diag), stream()),
          'hk_bgemm_f32')
    return out
