// Fast MPN-COV head (replaces model/methods/MPNCOV.py:105-230):
//   covariance pooling, Newton-Schulz matrix square root (forward + the
//   hand-derived backward), upper-triangle vectorisation.
// All contractions run on the fp32 MFMA path through hk::bgemm_kernel with fused
// epilogues (alpha * s_b * acc + beta * C + diag * I); the elementwise glue the
// reference spends ~25 small kernels on per direction is folded into those
// epilogues or into four small kernels below.  No host loop over the batch
// (reference: MPNCOV.py:198-201), no CPU-side index rebuild (:213-214).
#include <cstdlib>
#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// bcnn_fast.hip (panel-resident kernels; HK_ERR_UNSUPPORTED when the shape is not covered)
int gram_fast_raw(const float* x, const float* mu, float alpha, float* g, int B, int C, int HW, hipStream_t st);
int cov_fast_bwd(const float* x, const float* mu, const float* g, float* dx, int B, int C, int HW, hipStream_t st);
static inline bool force_generic() {
    const char* e = getenv("HK_BCNN_GENERIC");
    return e && e[0] == '1';
}

// ----------------------------------------------------------------- covariance
// mu[b,c] = mean_m x[b,c,m] : one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ x, float* __restrict__ mu,
                                                       long long rows, int M) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * M;
    float s = 0.f;
    for (int i = lane; i < M; i += 64) s += p[i];
    s = wave_sum(s);
    if (lane == 0) mu[row] = s / (float)M;
}

// ----------------------------------------------------------------- Newton-Schulz glue
// norm_a[b] = trace(a[b]) ; sq[b] = sqrt(norm_a[b])
__global__ __launch_bounds__(256) void ns_trace_kernel(const float* __restrict__ a, float* __restrict__ norm_a,
                                                       float* __restrict__ sq, int d) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* p = a + (long long)b * d * d;
    float s = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) s += p[(long long)i * d + i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        if (norm_a) norm_a[b] = s;
        sq[b] = sqrtf(s);
    }
}

// A = a / norm_a ; Z0 = 0.5 (3I - A)      (MPNCOV.py:146,150/153)
__global__ __launch_bounds__(256) void ns_scale_kernel(const float* __restrict__ a, const float* __restrict__ norm_a,
                                                       float* __restrict__ A, float* __restrict__ z0,
                                                       long long z0_bs, int d) {
    const int b = blockIdx.y;
    const long long n = (long long)d * d;
    const float na = norm_a[b];
    const float* p = a + b * n;
    float* q = A + b * n;
    float* z = z0 ? z0 + b * z0_bs : nullptr;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const float v = p[e] / na;
        q[e] = v;
        if (z) {
            const int i = (int)(e / d), j = (int)(e % d);
            z[e] = 0.5f * ((i == j ? 3.0f : 0.0f) - v);
        }
    }
}

// out = ca * x * (sb ? sb[b] : 1) + cb * y      (elementwise, per batch scale)
__global__ __launch_bounds__(256) void ns_axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ sb, float ca, float cb,
                                                       float* __restrict__ out, long long n) {
    const int b = blockIdx.y;
    const float s = sb ? ca * sb[b] : ca;
    const float* xp = x + b * n;
    const float* yp = y ? y + b * n : nullptr;
    float* o = out + b * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
        o[e] = s * xp[e] + (yp ? cb * yp[e] : 0.f);
}

// red0[b] = sum(g o out) ; red1[b] = sum(D^T o a)   (MPNCOV.py:175,197) - one workgroup per sample
__global__ __launch_bounds__(1024) void ns_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                             const float* __restrict__ D, const float* __restrict__ a,
                                                             float* __restrict__ red0, float* __restrict__ red1, int d) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const long long n = (long long)d * d;
    const float *gp = g + b * n, *op = out + b * n, *Dp = D + b * n, *ap = a + b * n;
    float s0 = 0.f, s1 = 0.f;
    for (long long e = threadIdx.x; e < n; e += 1024) {
        s0 += gp[e] * op[e];
        const int i = (int)(e / d), j = (int)(e % d);
        s1 += Dp[(long long)j * d + i] * ap[e];
    }
    s0 = block_sum<16>(s0, red);
    s1 = block_sum<16>(s1, red);
    if (threadIdx.x == 0) { red0[b] = s0; red1[b] = s1; }
}

// da = D^T / norm_a + (red0/(2 norm_a) - red1/norm_a^2) I      (MPNCOV.py:195-201)
__global__ __launch_bounds__(256) void ns_bwd_final_kernel(const float* __restrict__ D, const float* __restrict__ norm_a,
                                                           const float* __restrict__ red0, const float* __restrict__ red1,
                                                           float* __restrict__ da, int d) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long long n = (long long)d * d;
    const float na = norm_a[b];
    const float coef = red0[b] / (2.0f * na) - red1[b] / (na * na);
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // read D[j0.., i0..] (rows j, cols i) coalesced, write da[i][j]
    for (int r = ty; r < 32; r += 8) {
        const int jj = j0 + r, ii = i0 + tx;
        tile[r][tx] = (jj < d && ii < d) ? D[b * n + (long long)jj * d + ii] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ii = i0 + r, jj = j0 + tx;
        if (ii < d && jj < d) da[b * n + (long long)ii * d + jj] = tile[tx][r] / na + (ii == jj ? coef : 0.f);
    }
}

// ----------------------------------------------------------------- triu vec
__device__ __forceinline__ long long triu_off(int r, int d) { return (long long)r * d - (long long)r * (r - 1) / 2; }

__global__ __launch_bounds__(256) void triu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int d) {
    const int b = blockIdx.y, r = blockIdx.x;
    const long long L = (long long)d * (d + 1) / 2;
    const float* xp = x + ((long long)b * d + r) * d;
    float* yp = y + b * L + triu_off(r, d) - r;
    for (int c = r + threadIdx.x; c < d; c += 256) yp[c] = xp[c];
}

__global__ __launch_bounds__(256) void triu_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int d) {
    const int b = blockIdx.y, r = blockIdx.x;
    const long long L = (long long)d * (d + 1) / 2;
    float* xp = dx + ((long long)b * d + r) * d;
    const float* yp = dy + b * L + triu_off(r, d) - r;
    for (int c = threadIdx.x; c < d; c += 256) xp[c] = (c >= r) ? yp[c] : 0.f;
}

// HK_NS_SYM=1 (opt-in until measured on the GPU): every product of the forward chain and the Y Z products of the
// backward are products of commuting symmetric matrices (polynomials in A), so their results are symmetric: only
// the 10 of 16 tiles on or above the diagonal are computed and mirrored (hk_bgemm.h, SYM), and Z Y is not computed
// at all (it is the transpose of Y Z).  12 -> 7.5 and 38 -> 32.5 GEMM-equivalents at iterN = 5.  Requires a
// symmetric input (a covariance); CPU experiment (fp32 torch, d = 256): 7e-7 from the reference's full products,
// the same distance the reference itself is from fp64.
static inline bool ns_sym() {
    const char* e = getenv("HK_NS_SYM");
    return e && e[0] == '1';
}

// C = alpha * s_b * A B + beta C + diag I for d x d row-major batches with explicit batch strides
static inline int mm(const float* A, long long sa, const float* Bm, long long sb, float* C, long long sc, int d, int nb,
                     float alpha, const float* bscale, float beta, float diag, hipStream_t st, bool sym_result = false) {
    const LdPlain la = make_plain(A, sa, d, d, d);
    const LdPlain lb = make_plain(Bm, sb, d, d, d);
    const EpAffine ep = make_affine(C, sc, d, alpha, bscale, beta, diag);
    if (sym_result && ns_sym()) return bgemm_launch_sym<true, false>(la, lb, ep, d, d, nb, st);
    // A/B switch: 0 = 64x64x32 (default), 1 = 128x128x32 / 4 waves, 2 = 64x64x64, 3 = 64x64x16,
    //             4 = 128x128x32 / 8 waves / two-chunk prefetch, 5 = 64x64x32 / two-chunk prefetch,
    //             6 / 7 = fp32 product from six / three bf16 piece products on the bf16 matrix pipe (hk_bgemm.h),
    //             8 / 9 = the same on the 128x128 / 8-wave / two-chunk-prefetch tile
    const char* e = getenv("HK_NS_GEMM");
    const int variant = e ? atoi(e) : 0;
    if (variant == 4 && d >= 128) return bgemm128_launch<true, false>(la, lb, ep, d, d, d, nb, st);
    if (variant == 5) return bgemm64p2_launch<true, false>(la, lb, ep, d, d, d, nb, st);
    if (variant == 6) return bgemm_bf16split_launch<6>(la, lb, ep, d, d, d, nb, st);
    if (variant == 7) return bgemm_bf16split_launch<3>(la, lb, ep, d, d, d, nb, st);
    if (variant == 8 && d >= 128) return bgemm_bf16split128_launch<6>(la, lb, ep, d, d, d, nb, st);
    if (variant == 9 && d >= 128) return bgemm_bf16split128_launch<3>(la, lb, ep, d, d, d, nb, st);
    return bgemm_launch<true, false>(la, lb, ep, d, d, d, nb, st, variant);
}

// two results from one symmetric product P = A B:  C1 = 3 I - P  and  C2 = P   (backward: "YZ" and "ZY" = P^T = P)
struct EpDualNs {
    float *c1, *c2;
    long long bs;
    int ld;
    __device__ __forceinline__ void operator()(int b, int i, int j, float v) const {
        const long long o = (long long)b * bs + (long long)i * ld + j;
        c1[o] = (i == j ? 3.0f : 0.0f) - v;
        c2[o] = v;
    }
};

static inline int mm_yz_pair(const float* Y, const float* Z, long long sbs, float* W1, float* W2, long long n, int d,
                             int nb, hipStream_t st) {
    const LdPlain la = make_plain(Y, sbs, d, d, d);
    const LdPlain lb = make_plain(Z, sbs, d, d, d);
    EpDualNs ep;
    ep.c1 = W1; ep.c2 = W2; ep.bs = n; ep.ld = d;
    return bgemm_launch_sym<true, false>(la, lb, ep, d, d, nb, st);
}

static inline dim3 ew_grid(long long n, int B) {
    long long g = (n + 255) / 256;
    if (g > 256) g = 256;
    return dim3((unsigned)g, (unsigned)B);
}

}  // namespace hk

using namespace hk;

#define HK_TRY(x)                 \
    do {                          \
        int rc__ = (x);           \
        if (rc__ != HK_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------ cov pool
extern "C" int hk_cov_pool_fwd(const float* x, float* cov, float* mu, int B, int C, int M, hk_stream_t stream) {
    if (!x || !cov || !mu || B <= 0 || C <= 0 || M <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * C;
    hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, mu, rows, M);
    HK_LAUNCH_CHECK();
    if (!force_generic()) {
        const int rc = gram_fast_raw(x, mu, 1.0f / (float)M, cov, B, C, M, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    LdRowSub l;
    l.base = make_plain(x, (long long)C * M, M, C, M);
    l.mu = mu; l.mubs = C;
    const EpAffine ep = make_affine(cov, (long long)C * C, C, 1.0f / (float)M, nullptr, 0.f, 0.f);
    return bgemm_launch<true, true>(l, l, ep, C, C, M, B, st);
}

extern "C" int hk_cov_pool_bwd(const float* x, const float* mu, const float* dcov, float* dx, int B, int C, int M,
                               hk_stream_t stream) {
    if (!x || !mu || !dcov || !dx || B <= 0 || C <= 0 || M <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!force_generic()) {
        const int rc = cov_fast_bwd(x, mu, dcov, dx, B, C, M, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    LdSym ls;
    ls.p = dcov; ls.bs = (long long)C * C; ls.d = C;
    LdRowSub lx;  // B operand: K x N = C x M, rows are channels
    lx.base = make_plain(x, (long long)C * M, M, C, M);
    lx.mu = mu; lx.mubs = C;
    const EpAffine ep = make_affine(dx, (long long)C * M, M, 1.0f / (float)M, nullptr, 0.f, 0.f);
    return bgemm_launch<true, false>(ls, lx, ep, C, M, C, B, st);
}

// ------------------------------------------------------------------ NS sqrtm
extern "C" size_t hk_ns_sqrtm_ws_bytes(int B, int d, int iter_n, int backward) {
    (void)iter_n;
    const size_t mat = (size_t)B * d * d * sizeof(float);
    const size_t small = (size_t)4 * B * sizeof(float) + 256;
    return (backward ? 9 * mat : 2 * mat) + small;
}

extern "C" int hk_ns_sqrtm_fwd(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                               int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!a || !out || !norm_a || B <= 0 || d <= 0 || iter_n < 1) return HK_ERR_BAD_ARG;
    if (iter_n >= 2 && (!ysave || !zsave)) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_ns_sqrtm_ws_bytes(B, d, iter_n, 0)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)d * d;
    float* A = (float*)ws;
    float* T = A + (long long)B * n;
    float* sq = T + (long long)B * n;
    const int S = iter_n >= 2 ? iter_n - 1 : 1;   // slots in ysave / zsave
    const long long sbs = (long long)S * n;

    hipLaunchKernelGGL(ns_trace_kernel, dim3(B), dim3(256), 0, st, a, norm_a, sq, d);
    HK_LAUNCH_CHECK();
    if (iter_n < 2) {
        hipLaunchKernelGGL(ns_scale_kernel, ew_grid(n, B), dim3(256), 0, st, a, (const float*)norm_a, A, T, n, d);
        HK_LAUNCH_CHECK();
        return mm(A, n, T, n, out, n, d, B, 1.0f, sq, 0.f, 0.f, st, true);         // :151,:161
    }
    hipLaunchKernelGGL(ns_scale_kernel, ew_grid(n, B), dim3(256), 0, st, a, (const float*)norm_a, A, zsave, sbs, d);
    HK_LAUNCH_CHECK();
    HK_TRY(mm(A, n, zsave, sbs, ysave, sbs, d, B, 1.f, nullptr, 0.f, 0.f, st, true));   // Y0 = A ZY   :154
    for (int i = 1; i < iter_n - 1; ++i) {                                           // :156-159
        const float* Yp = ysave + (long long)(i - 1) * n;
        const float* Zp = zsave + (long long)(i - 1) * n;
        HK_TRY(mm(Zp, sbs, Yp, sbs, T, n, d, B, -0.5f, nullptr, 0.f, 1.5f, st, true));    // ZY = .5(3I - Z Y)
        HK_TRY(mm(Yp, sbs, T, n, ysave + (long long)i * n, sbs, d, B, 1.f, nullptr, 0.f, 0.f, st, true));
        HK_TRY(mm(T, n, Zp, sbs, zsave + (long long)i * n, sbs, d, B, 1.f, nullptr, 0.f, 0.f, st, true));
    }
    const float* Yl = ysave + (long long)(iter_n - 2) * n;
    const float* Zl = zsave + (long long)(iter_n - 2) * n;
    HK_TRY(mm(Zl, sbs, Yl, sbs, T, n, d, B, -1.f, nullptr, 0.f, 3.f, st, true));    // 3I - Z Y      :160
    return mm(Yl, sbs, T, n, out, n, d, B, 0.5f, sq, 0.f, 0.f, st, true);           // .5 Y (.) sqrt(normA)  :160-161
}

extern "C" int hk_ns_sqrtm_bwd(const float* a, const float* out, const float* norm_a, const float* ysave,
                               const float* zsave, const float* dout, float* da, int B, int d, int iter_n, void* ws,
                               size_t ws_bytes, hk_stream_t stream) {
    if (!a || !out || !norm_a || !dout || !da || B <= 0 || d <= 0 || iter_n < 1) return HK_ERR_BAD_ARG;
    if (iter_n >= 2 && (!ysave || !zsave)) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_ns_sqrtm_ws_bytes(B, d, iter_n, 1)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)d * d, bn = (long long)B * n;
    float* A = (float*)ws;
    float *W1 = A + bn, *W2 = W1 + bn, *W3 = W2 + bn;
    float *dY = W3 + bn, *dZ = dY + bn, *dYn = dZ + bn, *dZn = dYn + bn, *D = dZn + bn;
    float* sq = D + bn;
    float *red0 = sq + B, *red1 = red0 + B;
    const int S = iter_n >= 2 ? iter_n - 1 : 1;
    const long long sbs = (long long)S * n;
    const float* g = dout;

    hipLaunchKernelGGL(ns_trace_kernel, dim3(B), dim3(256), 0, st, a, (float*)nullptr, sq, d);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(ns_scale_kernel, ew_grid(n, B), dim3(256), 0, st, a, norm_a, A, (float*)nullptr, n, d);
    HK_LAUNCH_CHECK();

    if (iter_n < 2) {
        // der = .5 (dpc (3I - A) - A dpc),  dpc = sq g                                   :178
        hipLaunchKernelGGL(ns_axpby_kernel, ew_grid(n, B), dim3(256), 0, st, g, (const float*)nullptr,
                           (const float*)sq, 1.5f, 0.f, D, n);
        HK_LAUNCH_CHECK();
        HK_TRY(mm(g, n, A, n, D, n, d, B, -0.5f, sq, 1.f, 0.f, st));
        HK_TRY(mm(A, n, g, n, D, n, d, B, -0.5f, sq, 1.f, 0.f, st));
    } else {
        const float* Yl = ysave + (long long)(iter_n - 2) * n;
        const float* Zl = zsave + (long long)(iter_n - 2) * n;
        // dldY = .5 (dpc (3I - Yl Zl) - Zl Yl dpc)                                         :180-181
        if (ns_sym()) {
            HK_TRY(mm_yz_pair(Yl, Zl, sbs, W1, W2, n, d, B, st));                            // W1 = 3I - Yl Zl, W2 = Zl Yl
        } else {
            HK_TRY(mm(Yl, sbs, Zl, sbs, W1, n, d, B, -1.f, nullptr, 0.f, 3.f, st));
            HK_TRY(mm(Zl, sbs, Yl, sbs, W2, n, d, B, 1.f, nullptr, 0.f, 0.f, st));
        }
        HK_TRY(mm(g, n, W1, n, dY, n, d, B, 0.5f, sq, 0.f, 0.f, st));
        HK_TRY(mm(W2, n, g, n, dY, n, d, B, -0.5f, sq, 1.f, 0.f, st));
        // dldZ = -.5 Yl dpc Yl                                                             :182
        HK_TRY(mm(Yl, sbs, g, n, W3, n, d, B, 1.f, nullptr, 0.f, 0.f, st));
        HK_TRY(mm(W3, n, Yl, sbs, dZ, n, d, B, -0.5f, sq, 0.f, 0.f, st));
        for (int i = iter_n - 3; i >= 0; --i) {                                             // :183-193
            const float* Yi = ysave + (long long)i * n;
            const float* Zi = zsave + (long long)i * n;
            if (ns_sym()) {
                HK_TRY(mm_yz_pair(Yi, Zi, sbs, W1, W2, n, d, B, st));                      // both from one product
            } else {
                HK_TRY(mm(Yi, sbs, Zi, sbs, W1, n, d, B, -1.f, nullptr, 0.f, 3.f, st));    // YZ = 3I - Y Z
                HK_TRY(mm(Zi, sbs, Yi, sbs, W2, n, d, B, 1.f, nullptr, 0.f, 0.f, st));     // ZY = Z Y
            }
            HK_TRY(mm(dY, n, W1, n, dYn, n, d, B, 0.5f, nullptr, 0.f, 0.f, st));           // .5 dldY YZ
            HK_TRY(mm(Zi, sbs, dZ, n, W3, n, d, B, 1.f, nullptr, 0.f, 0.f, st));           // Z dldZ
            HK_TRY(mm(W3, n, Zi, sbs, dYn, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));        //  - .5 (Z dldZ) Z
            HK_TRY(mm(W2, n, dY, n, dYn, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));          //  - .5 ZY dldY
            HK_TRY(mm(W1, n, dZ, n, dZn, n, d, B, 0.5f, nullptr, 0.f, 0.f, st));           // .5 YZ dldZ
            HK_TRY(mm(Yi, sbs, dY, n, W3, n, d, B, 1.f, nullptr, 0.f, 0.f, st));           // Y dldY
            HK_TRY(mm(W3, n, Yi, sbs, dZn, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));        //  - .5 (Y dldY) Y
            HK_TRY(mm(dZ, n, W2, n, dZn, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));          //  - .5 dldZ ZY
            float* t = dY; dY = dYn; dYn = t;
            t = dZ; dZ = dZn; dZn = t;
        }
        // der = .5 (dldY (3I - A) - dldZ - A dldY) = 1.5 dldY - .5 dldZ - .5 dldY A - .5 A dldY   :194
        hipLaunchKernelGGL(ns_axpby_kernel, ew_grid(n, B), dim3(256), 0, st, (const float*)dY, (const float*)dZ,
                           (const float*)nullptr, 1.5f, -0.5f, D, n);
        HK_LAUNCH_CHECK();
        HK_TRY(mm(dY, n, A, n, D, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));
        HK_TRY(mm(A, n, dY, n, D, n, d, B, -0.5f, nullptr, 1.f, 0.f, st));
    }
    hipLaunchKernelGGL(ns_bwd_reduce_kernel, dim3(B), dim3(1024), 0, st, g, out, (const float*)D, a, red0, red1, d);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(ns_bwd_final_kernel, dim3((d + 31) / 32, (d + 31) / 32, B), dim3(256), 0, st, (const float*)D,
                       norm_a, (const float*)red0, (const float*)red1, da, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// ------------------------------------------------------------------ triu vec
extern "C" int hk_triu_vec_fwd(const float* x, float* y, int B, int d, hk_stream_t stream) {
    if (!x || !y || B <= 0 || d <= 0) return HK_ERR_BAD_ARG;
    hipLaunchKernelGGL(triu_fwd_kernel, dim3(d, B), dim3(256), 0, (hipStream_t)stream, x, y, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_triu_vec_bwd(const float* dy, float* dx, int B, int d, hk_stream_t stream) {
    if (!dy || !dx || B <= 0 || d <= 0) return HK_ERR_BAD_ARG;
    hipLaunchKernelGGL(triu_bwd_kernel, dim3(d, B), dim3(256), 0, (hipStream_t)stream, dy, dx, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
