// Fast MPN-COV head (replaces model/methods/MPNCOV.py:105-230):
//   covariance pooling, Newton-Schulz matrix square root (forward + the
//   hand-derived backward), upper-triangle vectorisation.
// All contractions run on the fp32 MFMA path: the covariance on the panel kernels of
// bcnn_fast.hip (generic fallback hk::bgemm_kernel), the Newton-Schulz chain on the grouped
// kernel of hk_nsmm.h (9 launches per direction); the elementwise glue the reference spends
// ~25 small kernels on per direction is folded into GEMM epilogues or the four small kernels below.  No host loop over the batch
// (reference: MPNCOV.py:198-201), no CPU-side index rebuild (:213-214).
#include <cstdlib>
#include "hk_bgemm.h"
#include <cstring>
#include "hk_nsmm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// bcnn_fast.hip (panel-resident kernels; HK_ERR_UNSUPPORTED when the shape is not covered)
int gram_fast_raw(const float* x, float* mu, float alpha, float* g, int B, int C, int HW, hipStream_t st);
int cov_fast_bwd(const float* x, const float* mu, const float* g, float* dx, int B, int C, int HW, hipStream_t st);
static inline bool force_generic() { return tuning().bcnn_generic == 1; }   // A/B lever (hk_tuning_set)

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
// One pass over a:  norm_a[b] = trace(a[b]) (computed here when TRACE, else read), sq[b] = sqrt(norm_a[b]),
// A = a / norm_a, Z0 = 0.5 (3I - A)      (MPNCOV.py:144-146,150/153).  Every workgroup of a sample recomputes the
// trace itself (d diagonal elements, fixed-order block sum: identical in all of them) instead of waiting for a
// separate trace kernel.
template <bool TRACE>
__global__ __launch_bounds__(256) void ns_scale_kernel(const float* __restrict__ a, float* __restrict__ norm_a,
                                                       float* __restrict__ sq, float* __restrict__ A,
                                                       float* __restrict__ z0, long long z0_bs, int d) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long long n = (long long)d * d;
    const float* p = a + b * n;
    float na;
    if (TRACE) {
        float s = 0.f;
        for (int i = threadIdx.x; i < d; i += 256) s += p[(long long)i * d + i];
        na = block_sum<4>(s, red);
        if (blockIdx.x == 0 && threadIdx.x == 0) norm_a[b] = na;
    } else {
        na = norm_a[b];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sq[b] = sqrtf(na);
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

// Partial sums of  red0 = sum(g o out)  and  red1 = sum(D^T o a)   (MPNCOV.py:175,197): one workgroup per 32x32
// tile, the transposed operand read coalesced and turned through LDS; part[b][tile][0/1].  (Round 1 ran this as ONE
// workgroup per sample with a strided read of D: 114 us of the 927 us backward, profiles/r2_ns_kernel_trace.csv.)
__global__ __launch_bounds__(256) void ns_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                            const float* __restrict__ D, const float* __restrict__ a,
                                                            float* __restrict__ part, int d, int nt) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int b = blockIdx.y, ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const long long n = (long long)d * d;
    const int i0 = ti * 32, j0 = tj * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {                        // D[j0 + r][i0 + tx]
        const int jj = j0 + r, ii = i0 + tx;
        tile[r][tx] = (jj < d && ii < d) ? D[b * n + (long long)jj * d + ii] : 0.f;
    }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    for (int r = ty; r < 32; r += 8) {                        // element (i0 + r, j0 + tx)
        const int ii = i0 + r, jj = j0 + tx;
        if (ii < d && jj < d) {
            const long long e = b * n + (long long)ii * d + jj;
            s0 += g[e] * out[e];
            s1 += tile[tx][r] * a[e];
        }
    }
    s0 = block_sum<4>(s0, red);
    s1 = block_sum<4>(s1, red);
    if (threadIdx.x == 0) {
        part[((long long)b * nt * nt + blockIdx.x) * 2] = s0;
        part[((long long)b * nt * nt + blockIdx.x) * 2 + 1] = s1;
    }
}

// da = D^T / norm_a + (red0/(2 norm_a) - red1/norm_a^2) I      (MPNCOV.py:195-201); red0 / red1 = the tile partials
// above - or, for iterN >= 2, the np tile partials written by the chain's last product (hk_nsmm.h, LAST) - added in
// tile order (every workgroup does it for itself: np <= 64 values at d = 256)
__global__ __launch_bounds__(256) void ns_bwd_final_kernel(const float* __restrict__ D, const float* __restrict__ norm_a,
                                                           const float* __restrict__ part, int np,
                                                           float* __restrict__ da, int d) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int b = blockIdx.z;
    const long long n = (long long)d * d;
    const float na = norm_a[b];
    // the tile partials, one per thread, then the fixed-order block sum (a serial loop over them in every one of the
    // 4096 workgroups doubled this kernel's time)
    float r0 = 0.f, r1 = 0.f;
    for (int t = threadIdx.x; t < np; t += 256) {
        r0 += part[((long long)b * np + t) * 2];
        r1 += part[((long long)b * np + t) * 2 + 1];
    }
    r0 = block_sum<4>(r0, red);
    r1 = block_sum<4>(r1, red);
    const float coef = r0 / (2.0f * na) - r1 / (na * na);
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

// The same when the chain's last product has already written D^T / norm_a into da (hk_nsmm.h, LAST on the aligned path):
// only the diagonal term is left - one workgroup per sample.
__global__ __launch_bounds__(256) void ns_bwd_diag_kernel(const float* __restrict__ norm_a, const float* __restrict__ part,
                                                          int np, float* __restrict__ da, int d) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float na = norm_a[b];
    float r0 = 0.f, r1 = 0.f;
    for (int t = threadIdx.x; t < np; t += 256) {
        r0 += part[((long long)b * np + t) * 2];
        r1 += part[((long long)b * np + t) * 2 + 1];
    }
    r0 = block_sum<4>(r0, red);
    r1 = block_sum<4>(r1, red);
    const float coef = r0 / (2.0f * na) - r1 / (na * na);
    float* p = da + (long long)b * d * d;
    for (int i = threadIdx.x; i < d; i += 256) p[(long long)i * d + i] += coef;
}

// ----------------------------------------------------------------- triu vec
__device__ __forceinline__ long long triu_off(int r, int d) { return (long long)r * d - (long long)r * (r - 1) / 2; }

// A workgroup owns TRIU_RPB consecutive rows of one matrix, a wave one row at a time, a lane the columns r + lane + 64 k:
// both sides are coalesced, and a thread has TRIU_RPB / 4 x 4 independent loads in flight (one workgroup of 256 threads
// per row - a kilobyte each - was launch-bound: 8.0 / 9.6 us for 17 / 25 MB).
constexpr int TRIU_RPB = 16;
__global__ __launch_bounds__(256) void triu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int d) {
    const int b = blockIdx.y, r0 = blockIdx.x * TRIU_RPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long L = (long long)d * (d + 1) / 2;
#pragma unroll
    for (int rr = wave; rr < TRIU_RPB; rr += 4) {
        const int r = r0 + rr;
        if (r >= d) break;
        const float* xp = x + ((long long)b * d + r) * d;
        float* yp = y + b * L + triu_off(r, d) - r;
        for (int c0 = r + lane; c0 < d; c0 += 256) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = c0 + 64 * k < d ? xp[c0 + 64 * k] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + 64 * k < d) yp[c0 + 64 * k] = v[k];
        }
    }
}

__global__ __launch_bounds__(256) void triu_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int d) {
    const int b = blockIdx.y, r0 = blockIdx.x * TRIU_RPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long L = (long long)d * (d + 1) / 2;
#pragma unroll
    for (int rr = wave; rr < TRIU_RPB; rr += 4) {
        const int r = r0 + rr;
        if (r >= d) break;
        float* xp = dx + ((long long)b * d + r) * d;
        const float* yp = dy + b * L + triu_off(r, d) - r;
        for (int c0 = lane; c0 < d; c0 += 256) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 64 * k;
                v[k] = (c < d && c >= r) ? yp[c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + 64 * k < d) xp[c0 + 64 * k] = v[k];
        }
    }
}

// ----------------------------------------------------------------- Newton-Schulz products (hk_nsmm.h)
// Every product of the chain is a launch of hk::nsmm_kernel over a GROUP of independent problems.  The iterates
// Y_i, Z_i are polynomials in the normalised input A (Y_0 = A (3I - A) / 2, Z_0 = (3I - A) / 2, and every step multiplies
// polynomials in A), so they commute and Z_i Y_i = Y_i Z_i for ANY input, symmetric or not: the backward takes "YZ"
// and "ZY" (MPNCOV.py:184-185) from ONE product (two results of the same accumulator).  Measured distance from the
// reference's separate products: 7e-7, the distance of the reference itself from fp64, on covariances and on
// non-symmetric inputs alike (tests/test_gpu_kernels.py::test_ns_general_input_backward).
static inline NsGroup ns_group(const NsProb& p0) {
    NsGroup g;
    g.p[0] = p0; g.p[1] = p0; g.p[2] = p0; g.p[3] = p0;
    g.np = 1;
    return g;
}
static inline NsGroup& operator+=(NsGroup& g, const NsProb& p) {
    g.p[g.np++] = p;
    return g;
}
// single product C = alpha * s_b * A B + diag I
static inline NsProb ns_single(const float* A, long long sa, const float* Bm, long long sb, float* C, long long sc,
                               float alpha, float diag, const float* bscale = nullptr) {
    NsProb p = ns_prob(C, sc, alpha, diag, bscale);
    p += ns_term(A, sa, Bm, sb);
    return p;
}
// P = Y Z  ->  W1 = 3 I - P ,  W2 = P      (MPNCOV.py:180,184-185)
static inline NsProb ns_yz_pair(const float* Y, const float* Z, long long sbs, float* W1, float* W2, long long n) {
    NsProb p = ns_single(Y, sbs, Z, sbs, W1, n, -1.f, 3.f);
    if (W2) { p.C2 = W2; p.sc2 = n; p.alpha2 = 1.f; p.diag2 = 0.f; }
    return p;
}

// Dispatch of one group launch of the chain.  With tuning().ns_streams == 1 and a batch of >= 16 samples the two
// halves of the batch run the SAME chain on two HIP queues (the caller's stream and a library-owned one, joined before
// the entry point returns).  Why: every workgroup of a launch is resident at once (512 workgroups = 2 per CU), so all
// of them are in their prologue (first operand chunk: nothing for the matrix pipe to do) and in their epilogue (the
// store burst) at the same time - 8-12 us per launch of a 30-120 us launch (profiles/r2_ns_*).  Two independent chains
// on two queues drift apart, and one's fill / drain is covered by the other's main loop.
struct NsDispatch {
    int d, B;
    hipStream_t st;
    AuxScope aux;
    bool sym;                // every product of the chain has a symmetric result (hk_ns_sqrtm_fwd_sym)
    NsDispatch(int d_, int B_, hipStream_t st_, bool sym_ = false)
        : d(d_), B(B_), st(st_), aux(st_, B_ >= 16 ? (tuning().ns_streams < B_ / 8 ? tuning().ns_streams : B_ / 8 - 1) : 0),
          sym(sym_) {}
    int operator()(const NsGroup& g, bool first = false, bool last = false) const {
        const int parts = aux.count() + 1;
        if (parts == 1) return nsmm_launch(g, d, B, st, 0, 0, sym, first, last);
        for (int i = 0; i < parts; ++i) {                    // samples [b0, b1) on queue i (0: the caller's stream)
            const int b0 = (int)((long long)B * i / parts), b1 = (int)((long long)B * (i + 1) / parts);
            const int rc = nsmm_launch(g, d, b1 - b0, i == 0 ? st : aux.aux(i - 1), 0, b0, sym, first, last);
            if (rc != HK_OK) return rc;
        }
        return HK_OK;
    }
    // the helper queues' work happens before whatever is enqueued on `st` next (also done by the destructor: an error
    // return between fork and join leaves nothing running unordered on a helper queue)
    int join() { return aux.join(); }
};

// grid of ns_scale_kernel: a few workgroups per sample (each one recomputes the trace before it starts: with the
// 256-per-sample grid of an elementwise kernel that prologue costs more than the scaling, 26 us instead of ~8)
static inline dim3 ns_scale_grid(long long n, int B) {
    long long g = n / 4096;
    if (g < 1) g = 1;
    if (g > 16) g = 16;
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
    if (!force_generic()) {      // the panel kernel computes the channel means itself (and writes mu for the backward)
        const int rc = gram_fast_raw(x, mu, 1.0f / (float)M, cov, B, C, M, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    const long long rows = (long long)B * C;
    hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, mu, rows, M);
    HK_LAUNCH_CHECK();
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
    const size_t mat = (size_t)B * d * d * sizeof(float);
    const size_t nt = (size_t)(d + 31) / 32;
    const size_t small = ((size_t)B + (backward ? 2 * (size_t)B * nt * nt : 0)) * sizeof(float) + 256;
    if (backward) return 10 * mat + small;
    (void)iter_n;
    return 2 * mat + small + 256;                  // forward: A + T
}

// Forward schedule at iterN = 5 (MPNCOV.py:137-164): 12 products in 9 launches -
//   Y0 = A ZY0 | for i = 1..3: ZY = .5 (3I - Z Y) | {Y' = Y ZY, Z' = ZY Z} in one launch | 3I - Z Y | .5 sqrt(tr) Y (.)
// sym = true: `a` is symmetric, so is every product of the chain - the launches skip the tiles below the diagonal blocks
// (3 of 4 tiles at d = 256) and mirror the rest; the results differ from the full products by the rounding asymmetry
// of a product of commuting symmetric matrices (~1e-7 relative: tests/test_gpu_full_shapes.py).
static int ns_sqrtm_fwd_impl(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                             int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream, bool sym, float* tv = nullptr) {
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

    if (iter_n < 2) {
        hipLaunchKernelGGL(ns_scale_kernel<true>, ns_scale_grid(n, B), dim3(256), 0, st, a, norm_a, sq, A, T, n, d);
        HK_LAUNCH_CHECK();
        NsProb p1 = ns_single(A, n, T, n, out, n, 1.0f, 0.f, sq);                                  // :151,:161
        p1.tv = tv;
        return nsmm_launch(ns_group(p1), d, B, st, 0, 0, sym);
    }
    // the first launch works on a itself: trace, normalisation, Z0 = ZY and Y0 = A ZY in one kernel         :144-154
    NsDispatch L(d, B, st, sym);
    {
        NsProb p0 = ns_single(a, n, a, n, ysave, sbs, 1.f, 0.f);
        p0.E1 = a; p0.se1 = n;
        p0.C2 = zsave; p0.sc2 = sbs;
        p0.norm_out = norm_a;
        HK_TRY(L(ns_group(p0), true));
    }
    for (int i = 1; i < iter_n - 1; ++i) {                                                          // :156-159
        const float* Yp = ysave + (long long)(i - 1) * n;
        const float* Zp = zsave + (long long)(i - 1) * n;
        HK_TRY(L(ns_group(ns_single(Zp, sbs, Yp, sbs, T, n, -0.5f, 1.5f))));                       // ZY = .5(3I - Z Y)
        NsGroup g = ns_group(ns_single(Yp, sbs, T, n, ysave + (long long)i * n, sbs, 1.f, 0.f));   // Y' = Y ZY
        g += ns_single(T, n, Zp, sbs, zsave + (long long)i * n, sbs, 1.f, 0.f);                    // Z' = ZY Z
        HK_TRY(L(g));
    }
    const float* Yl = ysave + (long long)(iter_n - 2) * n;
    const float* Zl = zsave + (long long)(iter_n - 2) * n;
    HK_TRY(L(ns_group(ns_single(Zl, sbs, Yl, sbs, T, n, -1.f, 3.f))));                             // 3I - Z Y      :160
    {
        NsProb pl = ns_single(Yl, sbs, T, n, out, n, 0.5f, 0.f, norm_a);                           // .5 Y (.) sqrt(normA)
        pl.bscale_fn = 1;
        pl.tv = tv;
        HK_TRY(L(ns_group(pl)));
    }
    return L.join();
}

extern "C" int hk_ns_sqrtm_fwd(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                               int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream) {
    return ns_sqrtm_fwd_impl(a, out, norm_a, ysave, zsave, B, d, iter_n, ws, ws_bytes, stream, false);
}

// Sqrtm + Triuvec (MPNCOV.py:88-92 applies them back to back): the chain's last product also writes the packed upper
// triangle tv [B, d (d + 1) / 2] - no separate pass over `out`
extern "C" int hk_ns_sqrtm_triu_fwd(const float* a, float* out, float* tv, float* norm_a, float* ysave, float* zsave, int B,
                                    int d, int iter_n, int symmetric, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!tv) return HK_ERR_BAD_ARG;
    return ns_sqrtm_fwd_impl(a, out, norm_a, ysave, zsave, B, d, iter_n, ws, ws_bytes, stream,
                             symmetric != 0 && tuning().ns_sym != 0, tv);
}

extern "C" int hk_ns_sqrtm_fwd_sym(const float* a, float* out, float* norm_a, float* ysave, float* zsave, int B, int d,
                                   int iter_n, void* ws, size_t ws_bytes, hk_stream_t stream) {
    return ns_sqrtm_fwd_impl(a, out, norm_a, ysave, zsave, B, d, iter_n, ws, ws_bytes, stream, tuning().ns_sym != 0);
}

// Backward schedule at iterN = 5 (MPNCOV.py:166-202): the reference's 38 products as 34 in 9 launches -
//   {YZ pair, Yl g} | {dldY, dldZ} | 3 x ( {YZ pair, Z dldZ, Y dldY} | {dldY', dldZ'} as two K = 3d sums ) | der
// general = false: Z_i Y_i is taken from the Y_i Z_i accumulator - 34 products; exact for any `a` (the iterates are
//                  polynomials in A = a / tr(a) and commute).
// general = true : Z_i Y_i is its own product in the same launch - the reference's 38, literally (MPNCOV.py:180-185).
static int ns_sqrtm_bwd_impl(const float* a, const float* out, const float* norm_a, const float* ysave,
                             const float* zsave, const float* dout, float* da, int B, int d, int iter_n, void* ws,
                             size_t ws_bytes, hk_stream_t stream, bool general) {
    if (!a || !out || !norm_a || !dout || !da || B <= 0 || d <= 0 || iter_n < 1) return HK_ERR_BAD_ARG;
    if (iter_n >= 2 && (!ysave || !zsave)) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_ns_sqrtm_ws_bytes(B, d, iter_n, 1)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)d * d, bn = (long long)B * n;
    float* A = (float*)ws;
    float *W1 = A + bn, *W2 = W1 + bn, *W3 = W2 + bn, *W4 = W3 + bn;
    float *dY = W4 + bn, *dZ = dY + bn, *dYn = dZ + bn, *dZn = dYn + bn, *D = dZn + bn;
    float* sq = D + bn;
    float* part = sq + B;                            // [B][nt * nt][2] partial sums of the two trace terms
    const int nt = (d + 31) / 32;
    const int S = iter_n >= 2 ? iter_n - 1 : 1;
    const long long sbs = (long long)S * n;
    const float* g = dout;

    if (iter_n < 2) {      // one iteration: the plain schedule (scale pass, product, reduction pass)
        hipLaunchKernelGGL(ns_scale_kernel<false>, ns_scale_grid(n, B), dim3(256), 0, st, a, const_cast<float*>(norm_a), sq, A,
                           (float*)nullptr, n, d);    // sq = sqrt(norm_a): the trace the forward saved
        HK_LAUNCH_CHECK();
    }
    int npart = nt * nt;                             // partial sums per sample handed to ns_bwd_final_kernel
    bool transposed = false;                         // the chain's last launch wrote D^T / norm_a into da

    NsDispatch L(d, B, st);
    if (iter_n < 2) {
        // der = .5 (dpc (3I - A) - A dpc) = sq (1.5 g - .5 (g A + A g)),  dpc = sq g                 :178
        NsProb p = ns_prob(D, n, -0.5f, 0.f, sq);
        p += ns_term(g, n, A, n);
        p += ns_term(A, n, g, n);
        p.E1 = g; p.se1 = n; p.e1 = 1.5f; p.e1_scaled = 1;
        HK_TRY(L(ns_group(p)));
    } else {
        const float* Yl = ysave + (long long)(iter_n - 2) * n;
        const float* Zl = zsave + (long long)(iter_n - 2) * n;
        {   // W1 = 3I - Yl Zl, W2 = Zl Yl (= Yl Zl for a symmetric input), W3 = Yl g
            NsGroup gr = ns_group(ns_yz_pair(Yl, Zl, sbs, W1, general ? nullptr : W2, n));
            if (general) gr += ns_single(Zl, sbs, Yl, sbs, W2, n, 1.f, 0.f);
            gr += ns_single(Yl, sbs, g, n, W3, n, 1.f, 0.f);
            HK_TRY(L(gr));
        }
        {   // dldY = .5 sq (g (3I - Yl Zl) - Zl Yl g)   :180-181 ;  dldZ = -.5 sq (Yl g) Yl   :182
            NsProb py = ns_prob(dY, n, 0.5f, 0.f, norm_a);
            py.bscale_fn = 1;                                                                       // sq = sqrt(norm_a)
            py += ns_term(g, n, W1, n);
            py += ns_term(W2, n, g, n, -1.f);
            NsGroup gr = ns_group(py);
            NsProb pz = ns_single(W3, n, Yl, sbs, dZ, n, -0.5f, 0.f, norm_a);
            pz.bscale_fn = 1;
            gr += pz;
            HK_TRY(L(gr));
        }
        for (int i = iter_n - 3; i >= 0; --i) {                                                      // :183-193
            const float* Yi = ysave + (long long)i * n;
            const float* Zi = zsave + (long long)i * n;
            {   // W1 = YZ = 3I - Y Z, W2 = ZY = Z Y, W3 = Z dldZ, W4 = Y dldY
                NsGroup gr = ns_group(ns_yz_pair(Yi, Zi, sbs, W1, general ? nullptr : W2, n));
                if (general) gr += ns_single(Zi, sbs, Yi, sbs, W2, n, 1.f, 0.f);
                gr += ns_single(Zi, sbs, dZ, n, W3, n, 1.f, 0.f);
                gr += ns_single(Yi, sbs, dY, n, W4, n, 1.f, 0.f);
                HK_TRY(L(gr));
            }
            {   // dldY' = .5 (dldY YZ - (Z dldZ) Z - ZY dldY) ;  dldZ' = .5 (YZ dldZ - (Y dldY) Y - dldZ ZY)
                NsProb py = ns_prob(dYn, n, 0.5f, 0.f);
                py += ns_term(dY, n, W1, n);
                py += ns_term(W3, n, Zi, sbs, -1.f);
                py += ns_term(W2, n, dY, n, -1.f);
                NsProb pz = ns_prob(dZn, n, 0.5f, 0.f);
                pz += ns_term(W1, n, dZ, n);
                pz += ns_term(W4, n, Yi, sbs, -1.f);
                pz += ns_term(dZ, n, W2, n, -1.f);
                NsGroup gr = ns_group(py);
                gr += pz;
                HK_TRY(L(gr));
            }
            float* t = dY; dY = dYn; dYn = t;
            t = dZ; dZ = dZn; dZn = t;
        }
        // der = .5 (dldY (3I - A) - dldZ - A dldY) = 1.5 dldY - .5 dldZ - .5 (dldY A + A dldY)   :194
        // with A = a / norm_a taken from a itself (the product's scale is per sample) and the two trace terms of
        // :175,197 reduced in this launch's epilogue: no scale pass in front of the chain, no reduction pass behind it
        NsProb p = ns_prob(D, n, -0.5f, 0.f, norm_a);
        p.bscale_fn = 2;
        p += ns_term(dY, n, a, n);
        p += ns_term(a, n, dY, n);
        p.E1 = dY; p.se1 = n; p.e1 = 1.5f;
        p.E2 = dZ; p.se2 = n; p.e2 = -0.5f;
        p.rg = g; p.rout = out; p.ra = a; p.rpart = part;
        p.C = da;                                    // aligned path: the launch writes D^T / norm_a into da itself
        transposed = nsmm_last_transposes(p, d);
        if (!transposed) p.C = D;
        HK_TRY(L(ns_group(p), false, true));
        npart = nsmm_last_tiles(d);
    }
    HK_TRY(L.join());
    if (transposed) {                                // da += (red0 / (2 norm_a) - red1 / norm_a^2) I
        hipLaunchKernelGGL(ns_bwd_diag_kernel, dim3(B), dim3(256), 0, st, norm_a, (const float*)part, npart, da, d);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    if (iter_n < 2) {
        hipLaunchKernelGGL(ns_bwd_reduce_kernel, dim3(nt * nt, B), dim3(256), 0, st, g, out, (const float*)D, a, part, d, nt);
        HK_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ns_bwd_final_kernel, dim3(nt, nt, B), dim3(256), 0, st, (const float*)D, norm_a,
                       (const float*)part, npart, da, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_ns_sqrtm_bwd(const float* a, const float* out, const float* norm_a, const float* ysave,
                               const float* zsave, const float* dout, float* da, int B, int d, int iter_n, void* ws,
                               size_t ws_bytes, hk_stream_t stream) {
    return ns_sqrtm_bwd_impl(a, out, norm_a, ysave, zsave, dout, da, B, d, iter_n, ws, ws_bytes, stream, false);
}

extern "C" int hk_ns_sqrtm_bwd_general(const float* a, const float* out, const float* norm_a, const float* ysave,
                                       const float* zsave, const float* dout, float* da, int B, int d, int iter_n,
                                       void* ws, size_t ws_bytes, hk_stream_t stream) {
    return ns_sqrtm_bwd_impl(a, out, norm_a, ysave, zsave, dout, da, B, d, iter_n, ws, ws_bytes, stream, true);
}

// ------------------------------------------------------------------ triu vec
extern "C" int hk_triu_vec_fwd(const float* x, float* y, int B, int d, hk_stream_t stream) {
    if (!x || !y || B <= 0 || d <= 0) return HK_ERR_BAD_ARG;
    hipLaunchKernelGGL(triu_fwd_kernel, dim3((d + TRIU_RPB - 1) / TRIU_RPB, B), dim3(256), 0, (hipStream_t)stream, x, y, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_triu_vec_bwd(const float* dy, float* dx, int B, int d, hk_stream_t stream) {
    if (!dy || !dx || B <= 0 || d <= 0) return HK_ERR_BAD_ARG;
    hipLaunchKernelGGL(triu_bwd_kernel, dim3((d + TRIU_RPB - 1) / TRIU_RPB, B), dim3(256), 0, (hipStream_t)stream, dy, dx, d);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
