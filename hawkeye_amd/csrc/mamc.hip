// MAMC n-pairs loss: the loss that follows the OSME head, forward value and d loss / d parts in one
// call.  replaces NPairsLoss.forward, model/loss/MAMC_loss.py:34-90 - there a python loop over the n = b*p anchors
// with ~30 tiny kernels and boolean-mask gathers (host syncs) per anchor.
//
//   x_hat = x / max(|x|, 1e-12)                    npairs_normalize_kernel   (one workgroup per row)
//   s     = x_hat x_hat^T                          f32-MFMA GEMM (hk_bgemm.h)
//   L_i, dL/ds[i,:]                                npairs_row_kernel         (one workgroup per anchor, row in LDS)
//   dL/dx_hat = (G + G^T) x_hat,  G = dL/ds        f32-MFMA GEMM, operand symmetrised in the loader
//   dL/dx through the normalisation, L = sum L_i/n npairs_finish_kernel
// All reductions run in a fixed order: bit-reproducible.
#include <cstdlib>

#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

__global__ __launch_bounds__(256) void npairs_normalize_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                              float* __restrict__ nrm, int D) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const float* xp = x + (long long)i * D;
    float ss = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) ss += xp[c] * xp[c];
    ss = block_sum<4>(ss, red);
    const float nv = fmaxf(sqrtf(ss), 1e-12f);                          // F.normalize: x / max(|x|_2, eps)
    for (int c = threadIdx.x; c < D; c += 256) xn[(long long)i * D + c] = xp[c] / nv;
    if (threadIdx.x == 0) nrm[i] = nv;
}

// Pair type of k relative to anchor i (MAMC_loss.py:49-55): 0 same attention & same class (contains i itself),
// 1 same attention & different class, 2 different attention & same class, 3 different & different.
__device__ __forceinline__ int pair_type(int ci, int ai, int ck, int ak) {
    return (ai == ak ? 0 : 2) + (ci == ck ? 0 : 1);
}
// is k (type tk) a negative of the positive j (type tj)?  :60-61, :71-72, :81-82
__device__ __forceinline__ bool is_negative(int tj, int tk) { return tj == 0 ? tk != 0 : (tj != 3 && tk == 3); }

__global__ __launch_bounds__(256) void npairs_row_kernel(const float* __restrict__ s, const int32_t* __restrict__ labels,
                                                        float* __restrict__ ds, float* __restrict__ loss_rows, int n,
                                                        int p) {
    HK_DYN_LDS(sm);                                   // row[n], inv[n] = 1 / (1 + S_j), type[n]
    __shared__ float red[4];
    float* row = sm;
    float* inv = sm + n;
    int* type = reinterpret_cast<int*>(sm + 2 * n);
    const int i = blockIdx.x;
    const int ci = labels[i / p], ai = i % p;
    for (int k = threadIdx.x; k < n; k += 256) {
        row[k] = s[(long long)i * n + k];
        type[k] = pair_type(ci, ai, labels[k / p], k % p);
    }
    __syncthreads();
    float li = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const int tj = type[j];
        float S = 0.f;
        if (tj != 3) {
            const float sj = row[j];
            for (int k = 0; k < n; ++k)
                if (is_negative(tj, type[k])) S += expf(row[k] - sj);
            li += logf(1.0f + S);
        }
        inv[j] = (tj != 3) ? 1.0f / (1.0f + S) : 0.f;
    }
    li = block_sum<4>(li, red);                       // (its barriers also publish inv[])
    if (threadIdx.x == 0) loss_rows[i] = li;
    const float rn = 1.0f / (float)n;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int tk = type[k];
        const float sk = row[k];
        float g = (tk != 3) ? inv[k] - 1.0f : 0.f;    // as a positive: -S_k / (1 + S_k)
        for (int j = 0; j < n; ++j)
            if (is_negative(type[j], tk)) g += expf(sk - row[j]) * inv[j];
        ds[(long long)i * n + k] = g * rn;
    }
}

__global__ __launch_bounds__(256) void npairs_finish_kernel(const float* __restrict__ xn, const float* __restrict__ dxn,
                                                           const float* __restrict__ nrm, const float* __restrict__ loss_rows,
                                                           float* __restrict__ dx, float* __restrict__ loss, int n, int D) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const float* a = xn + (long long)i * D;
    const float* g = dxn + (long long)i * D;
    float dot = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) dot += a[c] * g[c];
    dot = block_sum<4>(dot, red);
    const float nv = nrm[i];
    const bool clamped = !(nv > 1e-12f);              // |x| <= eps: x_hat = x / eps is linear in x
    for (int c = threadIdx.x; c < D; c += 256)
        dx[(long long)i * D + c] = clamped ? g[c] / nv : (g[c] - a[c] * dot) / nv;
    if (i == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int r = 0; r < n; ++r) t += loss_rows[r];
        loss[0] = t / (float)n;
    }
}

}  // namespace hk

using namespace hk;

extern "C" size_t hk_npairs_ws_bytes(int n, int D) {
    if (n <= 0 || D <= 0) return 0;
    return ((size_t)2 * n * D + (size_t)2 * n * n + (size_t)2 * n) * sizeof(float) + 256;
}

extern "C" int hk_npairs_loss(const float* x, const int32_t* labels, float* loss, float* dx, int b, int p, int D, void* ws,
                              size_t ws_bytes, hk_stream_t stream) {
    if (!x || !labels || !loss || !dx || b <= 0 || p <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    const int n = b * p;
    if (!ws || ws_bytes < hk_npairs_ws_bytes(n, D)) return HK_ERR_WORKSPACE;
    if ((size_t)3 * n * sizeof(float) > 60 * 1024) return HK_ERR_UNSUPPORTED;       // row kernel keeps 3n words in LDS
    hipStream_t st = (hipStream_t)stream;
    float* xn = (float*)ws;
    float* dxn = xn + (size_t)n * D;
    float* s = dxn + (size_t)n * D;
    float* ds = s + (size_t)n * n;
    float* nrm = ds + (size_t)n * n;
    float* loss_rows = nrm + n;

    hipLaunchKernelGGL(npairs_normalize_kernel, dim3(n), dim3(256), 0, st, x, xn, nrm, D);
    HK_LAUNCH_CHECK();
    {
        const LdPlain l = make_plain(xn, 0, D, n, D);
        const EpAffine ep = make_affine(s, 0, n, 1.0f, nullptr, 0.f, 0.f);
        const int rc = bgemm_launch<true, true>(l, l, ep, n, n, D, 1, st);
        if (rc != HK_OK) return rc;
    }
    hipLaunchKernelGGL(npairs_row_kernel, dim3(n), dim3(256), (size_t)3 * n * sizeof(float), st, (const float*)s, labels, ds,
                       loss_rows, n, p);
    HK_LAUNCH_CHECK();
    {
        LdSym la;
        la.p = ds; la.bs = 0; la.d = n;
        const LdPlain lb = make_plain(xn, 0, D, n, D);
        const EpAffine ep = make_affine(dxn, 0, D, 1.0f, nullptr, 0.f, 0.f);
        const int rc = bgemm_launch<true, false>(la, lb, ep, n, D, n, 1, st);
        if (rc != HK_OK) return rc;
    }
    hipLaunchKernelGGL(npairs_finish_kernel, dim3(n), dim3(256), 0, st, (const float*)xn, (const float*)dxn,
                       (const float*)nrm, (const float*)loss_rows, dx, loss, n, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
