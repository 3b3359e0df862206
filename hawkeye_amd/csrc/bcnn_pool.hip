// BCNN bilinear pooling, forward + backward (replaces model/methods/BCNN.py:13-27).
//
//   G = X X^T / M ; z = sqrt(G + 1e-5) ; n = max(|z|_2, 1e-12) ; y = z / n
//
// Forward: the l2 norm does not need z:  |z|^2 = sum_ij (G_ij + 1e-5)
//   = (1/M) sum_hw (sum_c x[c,hw])^2 + C^2 * 1e-5   (exact identity), so a small
// column-sum kernel produces 1/n BEFORE the Gram kernel, whose epilogue writes
// the final y in one pass: X is read once, y written once, nothing else touches HBM.
//
// Backward (Appendix A of SURVEY.md):  t = <y,dy>,
//   dG = (dy - y t) / (2 n^2 y M),   dX = (dG + dG^T) X.
// With P_ij = (dy_ij + dy_ji) / (2 n^2 M y_ij):  dG + dG^T = P - (t / (n^2 M)) 11^T, so
//   dX = P X - (t / (n^2 M)) 1 colsum(X)^T.
// P is formed on the fly by the GEMM's A-operand loader from y / dy tiles (never
// materialised); t is accumulated by the same loader (each (y,dy) element exactly
// once, fixed order) and the rank-1 term is applied by a small second kernel.
#include <cstdlib>
#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// bcnn_fast.hip: panel-resident kernels for C % 64 == 0 and HW in {196,144,100,64}; HK_ERR_UNSUPPORTED otherwise
int bcnn_fast_gram(const float* x, const float* inv_norm, float* y, int B, int C, int HW, hipStream_t st);
int gram_fast_ssqrt(const float* x, float* y, float* part, int* nparts, int B, int C, int HW, hipStream_t st);
int bcnn_fast_gram_norm(const float* x, const float* part, int G, float* colsum, float* inv_norm, float* y, int B, int C,
                        int HW, hipStream_t st);
int bcnn_fast_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart, int B,
                  int C, int HW, hipStream_t st);
int bcnn_fast_bwd_fold(const float* x, const float* y, const float* dy, const float* inv_norm, const float* colsum,
                       const float* ta, const float* tb, const float* tc, int tK, float* dx, int B, int C, int HW,
                       hipStream_t st);
int bcnn_ssqrt_fast_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm, const float* ta,
                             const float* tb, const float* tc, int tK, int t_inv2, float* dx, int B, int C, int HW,
                             hipStream_t st);
int gram_fast_raw(const float* x, float* mu, float alpha, float* g, int B, int C, int HW, hipStream_t st);
int bcnn_ssqrt_fast_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, const float* tpart, int nt,
                        float* dx, int B, int C, int HW, hipStream_t st);
static inline bool force_generic() { return tuning().bcnn_generic == 1; }   // A/B lever (hk_tuning_set)

// colsum[b,hw] = sum_c x[b,c,hw];  inv_norm[b] = 1 / max(sqrt(sum_hw colsum^2 / M + C*C*1e-5), 1e-12)
// One workgroup (1024 threads) per sample; threads stride over channel rows with
// a fixed hw column each, LDS tree over the row groups: deterministic.
__global__ __launch_bounds__(1024) void bcnn_colsum_norm_kernel(const float* __restrict__ x,
                                                                float* __restrict__ colsum,
                                                                float* __restrict__ inv_norm, int C, int HW) {
    HK_DYN_LDS16(sm);  // [groups][HW] + 16
    const int b = blockIdx.x;
    const float* xb = x + (long long)b * C * HW;
    const int tid = threadIdx.x;
    // column-major assignment: thread handles column (tid % ncol) for rows tid/ncol, +groups, ...
    const int ncol = HW < 1024 ? HW : 1024;
    const int groups = 1024 / ncol;  // >= 1
    const int col = tid % ncol, grp = tid / ncol;
    float* part = sm;                // [groups][HW]
    float* red = sm + groups * HW;   // 16 floats
    for (int c0 = 0; c0 < HW; c0 += ncol) {
        const int hw = c0 + col;
        float s = 0.f;
        if (grp < groups && hw < HW)
            for (int c = grp; c < C; c += groups) s += xb[(long long)c * HW + hw];
        if (grp < groups && hw < HW) part[grp * HW + hw] = s;
    }
    __syncthreads();
    float ssq = 0.f;
    for (int hw = tid; hw < HW; hw += 1024) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += part[g * HW + hw];
        colsum[(long long)b * HW + hw] = s;
        ssq += s * s;
    }
    const float tot = block_sum<16>(ssq, red);
    if (tid == 0) {
        const float n2 = tot / (float)HW + (float)C * (float)C * 1e-5f;
        inv_norm[b] = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    }
}

// Two-stage variant with 8x more parallelism (the single kernel above runs on only B workgroups = a quarter of the
// chip at B = 64): stage 1 sums 64-channel groups (grid B x C/64, coalesced row reads, thread per column), stage 2
// adds the group partials in a fixed order and produces inv_norm.  Same summation tree for every launch: deterministic.
__global__ __launch_bounds__(256) void bcnn_colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                  int C, int HW, int G) {
    const int b = blockIdx.y, g = blockIdx.x;
    const int c0 = g * 64, c1 = (c0 + 64 < C) ? c0 + 64 : C;
    const float* xb = x + ((long long)b * C + c0) * HW;
    float* pp = part + ((long long)b * G + g) * HW;
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int c = 0;
        const int n = c1 - c0;
        for (; c + 4 <= n; c += 4) {
            s0 += xb[(long long)(c + 0) * HW + hw];
            s1 += xb[(long long)(c + 1) * HW + hw];
            s2 += xb[(long long)(c + 2) * HW + hw];
            s3 += xb[(long long)(c + 3) * HW + hw];
        }
        for (; c < n; ++c) s0 += xb[(long long)c * HW + hw];
        pp[hw] = (s0 + s1) + (s2 + s3);
    }
}

// The same stage with 16-byte loads (HW % 4 == 0, x 16-byte aligned): a 64-channel group is one contiguous block of
// 64 * HW floats; thread t owns the column quad t % (HW / 4) and the rows t / (HW / 4), + R, + 2 R, ... (R = 256 / (HW / 4)
// row phases), all of its loads independent and in flight together; the R partial quads of a column quad are added in
// phase order through LDS.  (The scalar version kept 4 loads in flight per thread: 7.7 us for 25.7 MB.)
__global__ __launch_bounds__(256) void bcnn_colsum_partial4_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                   int C, int HW, int G) {
    __shared__ float4 red[256];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int c0 = g * 64, n = (c0 + 64 < C) ? 64 : C - c0;
    const int Q = HW / 4, R = 256 / Q;                    // column quads, row phases (HW >= 4: Q <= 64 -> R >= 4)
    const int q = tid % Q, r = tid / Q;
    const float4* xb = reinterpret_cast<const float4*>(x + ((long long)b * C + c0) * HW);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) {
#pragma unroll 8
        for (int c = r; c < n; c += R) {
            const float4 v = xb[(long long)c * Q + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[tid] = s;
    __syncthreads();
    if (tid < Q) {
        float4 t = red[tid];
        for (int k = 1; k < R; ++k) {
            const float4 v = red[tid + k * Q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        reinterpret_cast<float4*>(part + ((long long)b * G + g) * HW)[tid] = t;
    }
}

__global__ __launch_bounds__(256) void bcnn_norm_finalize_kernel(const float* __restrict__ part, float* __restrict__ colsum,
                                                                 float* __restrict__ inv_norm, int C, int HW, int G) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* pp = part + (long long)b * G * HW;
    float ssq = 0.f;
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += pp[(long long)g * HW + hw];
        colsum[(long long)b * HW + hw] = s;
        ssq += s * s;
    }
    const float tot = block_sum<4>(ssq, red);
    if (threadIdx.x == 0) {
        const float n2 = tot / (float)HW + (float)C * (float)C * 1e-5f;
        inv_norm[b] = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    }
}

// Gram epilogue: y = sqrt(acc / M + 1e-5) * inv_norm[b]
struct EpBcnn {
    float* y;
    const float* inv_norm;
    int C;
    float m;  // HW as float
    __device__ __forceinline__ void operator()(int b, int i, int j, float v) const {
        y[(long long)b * C * C + (long long)i * C + j] = sqrtf(v / m + 1e-5f) * inv_norm[b];
    }
};

// A-operand loader of the backward GEMM:  P[i][k] = (dy[i][k] + dy[k][i]) / y[i][k] * coef[b],
// coef = inv_norm^2 / (2 M).  Also accumulates t-partials sum y*dy for the tile
// column tn == 0 (each row-block of y/dy is then visited exactly once).
struct LdBcnnP {
    const float* y;
    const float* dy;
    const float* inv_norm;
    float* tpart;  // [B][tilesM]
    int C;
    float inv2m;   // 1 / (2 M)
    int vec;
    float coef;
    float tacc;
    int active;
    const float* tsum;   // signed-sqrt variant: partial sums of t = <y, dy> [B][nt]; nullptr = the sqrt(x + 1e-5) variant
    int nt;
    float t2;
    __device__ __forceinline__ void begin(int b, int, int tn) {
        const float in = inv_norm[b];
        coef = in * in * inv2m;
        tacc = 0.f;
        active = (tn == 0) && !tsum;
        t2 = 0.f;
        if (tsum) {
            for (int c = 0; c < nt; ++c) t2 += tsum[(long long)b * nt + c];
            t2 *= 2.0f;
        }
    }
    __device__ __forceinline__ float4 ld4(int b, int r, int c) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < C && c < C) {
            const long long base = (long long)b * C * C;
            const float* yq = y + base + (long long)r * C + c;
            const float* dq = dy + base + (long long)r * C + c;
            float yv[4] = {1.f, 1.f, 1.f, 1.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec && c + 3 < C) {
                const float4 a = *reinterpret_cast<const float4*>(yq);
                const float4 d = *reinterpret_cast<const float4*>(dq);
                yv[0] = a.x; yv[1] = a.y; yv[2] = a.z; yv[3] = a.w;
                dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (c + t < C) { yv[t] = yq[t]; dv[t] = dq[t]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (c + t < C) {
                    const float dt = dy[base + (long long)(c + t) * C + r];
                    if (tsum) p[t] = yv[t] == 0.f ? 0.f : (dv[t] + dt - t2 * yv[t]) / fabsf(yv[t]) * coef;
                    else p[t] = (dv[t] + dt) / yv[t] * coef;
                    if (active) tacc += yv[t] * dv[t];
                }
            }
        }
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    __device__ __forceinline__ void finish(int b, int tm, int tn, int tilesM, float* red) {
        if (tn != 0 || tsum) return;  // uniform per workgroup
        const float s = block_sum<4>(tacc, red);
        if (threadIdx.x == 0) tpart[(long long)b * tilesM + tm] = s;
    }
};

// dx[b,c,hw] -= (t[b] * inv_norm[b]^2 / M) * colsum[b,hw],  t[b] = sum of the partials (fixed order)
__global__ __launch_bounds__(256) void bcnn_rank1_fix_kernel(float* __restrict__ dx, const float* __restrict__ tpart,
                                                             const float* __restrict__ inv_norm,
                                                             const float* __restrict__ colsum, int C, int HW,
                                                             int tilesM, long long per_sample) {
    const int b = blockIdx.y;
    float t = 0.f;
    for (int i = 0; i < tilesM; ++i) t += tpart[(long long)b * tilesM + i];
    const float in = inv_norm[b];
    const float k = t * in * in / (float)HW;
    float* d = dx + (long long)b * per_sample;
    const float* cs = colsum + (long long)b * HW;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per_sample; e += (long long)gridDim.x * 256)
        d[e] = fmaf(-k, cs[e % HW], d[e]);
}

// ----------------------------------------------------------------------------- signed-sqrt variant (BCNN.py:23-24)
// The reference keeps a second normalisation commented out next to the one it runs:
//     x = torch.sign(x) * torch.sqrt(torch.abs(x) + 1e-10)        (BCNN.py:23-24; the form CBCNN.py:132 applies)
// Its norm has no closed form from the column sums (sum |G_ij| is not a function of them once features can be
// negative), so it takes two elementwise passes after the raw Gram: u in place + partial sums of u^2, then the scale.
constexpr int SS_CHUNKS = 64;          // partial sums per image

// in place: u = sign(g) sqrt(|g| + 1e-10)  (sign(0) = 0);  part[b][chunk] = sum of u^2 over the chunk (fixed order)
__global__ __launch_bounds__(256) void ssqrt_apply_kernel(float* __restrict__ y, float* __restrict__ part, long long n) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long long len = (n + SS_CHUNKS - 1) / SS_CHUNKS, e0 = (long long)blockIdx.x * len;
    const long long e1 = e0 + len < n ? e0 + len : n;
    float* p = y + (long long)b * n;
    float s = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
        const float g = p[e];
        const float u = g == 0.f ? 0.f : copysignf(sqrtf(fabsf(g) + 1e-10f), g);
        p[e] = u;
        s += u * u;
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) part[(long long)b * SS_CHUNKS + blockIdx.x] = s;
}

// y *= inv_norm[b],  inv_norm[b] = 1 / max(sqrt(sum of the partials), 1e-12)   (F.normalize defaults)
__global__ __launch_bounds__(256) void ssqrt_scale_kernel(float* __restrict__ y, const float* __restrict__ part,
                                                          float* __restrict__ inv_norm, long long n, int nparts) {
    const int b = blockIdx.y;
    float s = 0.f;
    for (int c = 0; c < nparts; ++c) s += part[(long long)b * SS_CHUNKS + c];
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_norm[b] = inv;
    const long long len = (n + SS_CHUNKS - 1) / SS_CHUNKS, e0 = (long long)blockIdx.x * len;
    const long long e1 = e0 + len < n ? e0 + len : n;
    float* p = y + (long long)b * n;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) p[e] *= inv;
}

// inv_norm[b] alone (the scale itself is folded into the classifier that consumes u: hk_linear_fwd_scaled)
__global__ __launch_bounds__(64) void ssqrt_norm_kernel(const float* __restrict__ part, float* __restrict__ inv_norm, int B,
                                                        int nparts) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int c = 0; c < nparts; ++c) s += part[(long long)b * SS_CHUNKS + c];        // (ssqrt_scale_kernel's order)
    inv_norm[b] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// The backward on the UNSCALED u (y = inv u): with t = <y, dy> = inv <u, dy> the operand of the GEMM is
//     (dy + dy^T - 2 t y) / |y| * inv^2 / 2M = (dy + dy^T - 2 (inv^2 <u, dy>) u) / |u| * inv / 2M,
// i.e. the kernels below run unchanged on u with the t partials scaled by inv^2 and sqrt(inv) in the place of inv.
__global__ __launch_bounds__(64) void ssqrt_sqrt_inv_kernel(const float* __restrict__ inv_norm, float* __restrict__ sq, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < B) sq[b] = sqrtf(inv_norm[b]);
}

// part[b][chunk] = sum y * dy over the chunk (t = <y, dy> of the l2-normalisation backward)
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ part, long long n,
                                                          const float* __restrict__ inv = nullptr) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long long len = (n + SS_CHUNKS - 1) / SS_CHUNKS, e0 = (long long)blockIdx.x * len;
    const long long e1 = e0 + len < n ? e0 + len : n;
    const float* p = y + (long long)b * n;
    const float* q = dy + (long long)b * n;
    float s = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) s += p[e] * q[e];
    s = block_sum<4>(s, red);
    if (inv) s *= inv[b] * inv[b];                            // (unscaled u: see ssqrt_sqrt_inv_kernel)
    if (threadIdx.x == 0) part[(long long)b * SS_CHUNKS + blockIdx.x] = s;
}

}  // namespace hk

using namespace hk;

extern "C" size_t hk_bcnn_pool_ws_bytes(int B, int C, int HW) {
    const size_t tiles = (size_t)((C + 63) / 64);
    // backward: t partials [B][tiles]; forward: column-sum group partials [B][tiles][HW]
    return (size_t)B * tiles * (size_t)(HW > 1 ? HW : 1) * sizeof(float) + 256;
}

extern "C" int hk_bcnn_colsum_norm(const float* x, float* colsum, float* inv_norm, int B, int C, int HW, void* ws,
                                  size_t ws_bytes, hk_stream_t stream) {
    if (!x || !inv_norm || !colsum || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const int G = (C + 63) / 64;
    if (ws && ws_bytes >= (size_t)B * G * HW * sizeof(float) && G > 1) {      // two-stage, B*G workgroups
        if (HW % 4 == 0 && HW / 4 <= 64 && aligned16(x) && aligned16(ws))
            hipLaunchKernelGGL(bcnn_colsum_partial4_kernel, dim3(G, B), dim3(256), 0, (hipStream_t)stream, x, (float*)ws, C, HW, G);
        else
            hipLaunchKernelGGL(bcnn_colsum_partial_kernel, dim3(G, B), dim3(256), 0, (hipStream_t)stream, x, (float*)ws, C, HW, G);
        HK_LAUNCH_CHECK();
        hipLaunchKernelGGL(bcnn_norm_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, colsum,
                           inv_norm, C, HW, G);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    const int ncol = HW < 1024 ? HW : 1024;
    const int groups = 1024 / ncol;
    const size_t sm = ((size_t)groups * HW + 16) * sizeof(float);
    if (sm > 150 * 1024) return HK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bcnn_colsum_norm_kernel, dim3(B), dim3(1024), sm, (hipStream_t)stream, x, colsum, inv_norm, C, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bcnn_gram_norm(const float* x, const float* inv_norm, float* y, int B, int C, int HW,
                                 hk_stream_t stream) {
    if (!x || !y || !inv_norm || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!force_generic()) {
        const int rc = bcnn_fast_gram(x, inv_norm, y, B, C, HW, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    const LdPlain xa = make_plain(x, (long long)C * HW, HW, C, HW);
    EpBcnn ep;
    ep.y = y; ep.inv_norm = inv_norm; ep.C = C; ep.m = (float)HW;
    return bgemm_launch<true, true>(xa, xa, ep, C, C, HW, B, (hipStream_t)stream);
}

extern "C" int hk_bcnn_pool_fwd(const float* x, float* y, float* inv_norm, float* colsum, int B, int C, int HW,
                                void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !y || !inv_norm || !colsum || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const int G = (C + 63) / 64;
    // panel-resident Gram in ONE launch: every workgroup adds up its sample's columns itself (from the XCD's L2 for all
    // but the first of a sample's workgroups) with the arithmetic of the two kernels below - the same bits - and row
    // block 0 writes colsum / inv_norm (bcnn_fast.hip, GramNormSrc.direct; fwd_fold = -1: the two-launch route)
    if (!force_generic() && tuning().fwd_fold >= 0 && C % 64 == 0 && G > 1 && HW % 4 == 0 && HW / 4 <= 64 && aligned16(x)) {
        const int rc = bcnn_fast_gram_norm(x, nullptr, G, colsum, inv_norm, y, B, C, HW, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    // two launches - the 64-channel-group column sums, then the Gram kernel, which forms the norm from them in its
    // prologue and writes colsum / inv_norm
    if (!force_generic() && C % 64 == 0 && G > 1 && ws && ws_bytes >= (size_t)B * G * HW * sizeof(float) && HW % 4 == 0 &&
        HW / 4 <= 64 && aligned16(x) && aligned16(ws)) {
        hipLaunchKernelGGL(bcnn_colsum_partial4_kernel, dim3(G, B), dim3(256), 0, (hipStream_t)stream, x, (float*)ws, C, HW, G);
        HK_LAUNCH_CHECK();
        const int rc = bcnn_fast_gram_norm(x, (const float*)ws, G, colsum, inv_norm, y, B, C, HW, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    int rc = hk_bcnn_colsum_norm(x, colsum, inv_norm, B, C, HW, ws, ws_bytes, stream);
    if (rc != HK_OK) return rc;
    return hk_bcnn_gram_norm(x, inv_norm, y, B, C, HW, stream);
}

extern "C" int hk_bcnn_bwd_gemm(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx,
                                float* tpart, int B, int C, int HW, hk_stream_t stream) {
    if (!x || !y || !dy || !inv_norm || !dx || !tpart || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!force_generic()) {
        const int rc = bcnn_fast_bwd(x, y, dy, inv_norm, dx, tpart, B, C, HW, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    LdBcnnP pa;
    pa.y = y; pa.dy = dy; pa.inv_norm = inv_norm; pa.tpart = tpart; pa.C = C;
    pa.inv2m = 1.0f / (2.0f * (float)HW);
    pa.vec = (aligned16(y) && aligned16(dy) && (C % 4 == 0)) ? 1 : 0;
    pa.coef = 0.f; pa.tacc = 0.f; pa.active = 0;
    pa.tsum = nullptr; pa.nt = 0; pa.t2 = 0.f;
    const LdPlain xb = make_plain(x, (long long)C * HW, HW, C, HW);  // K x N, N contiguous
    const EpAffine ep = make_affine(dx, (long long)C * HW, HW, 1.0f, nullptr, 0.f, 0.f);
    // 64-row tiles only: tpart is indexed by 64-row block (hk_bcnn_bwd_rank1 sums ceil(C/64) partials)
    return bgemm_launch<true, false>(pa, xb, ep, C, HW, C, B, (hipStream_t)stream);
}

extern "C" int hk_bcnn_bwd_rank1(float* dx, const float* tpart, const float* inv_norm, const float* colsum, int B, int C,
                                 int HW, hk_stream_t stream) {
    if (!dx || !tpart || !inv_norm || !colsum || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const int tilesM = (C + 63) / 64;
    const long long per = (long long)C * HW;
    int gx = (int)((per + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(bcnn_rank1_fix_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dx, tpart, inv_norm, colsum,
                       C, HW, tilesM, per);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bcnn_pool_bwd(const float* x, const float* y, const float* dy, const float* inv_norm,
                                const float* colsum, float* dx, int B, int C, int HW, void* ws, size_t ws_bytes,
                                hk_stream_t stream) {
    if (!x || !y || !dy || !inv_norm || !colsum || !dx || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_bcnn_pool_ws_bytes(B, C, HW)) return HK_ERR_WORKSPACE;
    int rc = hk_bcnn_bwd_gemm(x, y, dy, inv_norm, dx, (float*)ws, B, C, HW, stream);
    if (rc != HK_OK) return rc;
    return hk_bcnn_bwd_rank1(dx, (const float*)ws, inv_norm, colsum, B, C, HW, stream);
}

// The same with t = <y, dy> handed over as a dot product of two small operands: t[b] = sum_k ta[b][k] (tb[b][k] - tc[k])
// (tc nullable).  When dy = g W is the input gradient of a linear layer on y, <y, dy> = sum_k g_k (logit_k - bias_k):
// ta = g, tb = the layer's output, tc = its bias, K its width.  One launch, no second pass over dX, no partial sums in
// the K loop.  Shapes the fast kernel does not serve: t is formed by a tiny kernel into ws and the two-launch route runs.
__global__ __launch_bounds__(64) void bcnn_tdot_kernel(const float* __restrict__ ta, const float* __restrict__ tb,
                                                       const float* __restrict__ tc, float* __restrict__ tpart, int K,
                                                       int slots) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float p = 0.f;
    for (int k = lane; k < K; k += 64) {
        const long long o = (long long)b * K + k;
        p = fmaf(ta[o], tb[o] - (tc ? tc[k] : 0.f), p);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m, 64);
    for (int i = lane; i < slots; i += 64) tpart[(long long)b * slots + i] = i == 0 ? p : 0.f;
}

extern "C" int hk_bcnn_pool_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm,
                                     const float* colsum, const float* ta, const float* tb, const float* tc, int K,
                                     float* dx, int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !y || !dy || !inv_norm || !colsum || !ta || !tb || !dx || K <= 0 || B <= 0 || C <= 0 || HW <= 0)
        return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_bcnn_pool_ws_bytes(B, C, HW)) return HK_ERR_WORKSPACE;
    if (!force_generic() && tuning().bwd_fold >= 0) {
        const int rc1 = bcnn_fast_bwd_fold(x, y, dy, inv_norm, colsum, ta, tb, tc, K, dx, B, C, HW, (hipStream_t)stream);
        if (rc1 != HK_ERR_UNSUPPORTED) return rc1;
    }
    // (the GEMM launch below overwrites ws with ITS partial sums of t: they are ignored - the dot product goes in after it)
    int rc = hk_bcnn_bwd_gemm(x, y, dy, inv_norm, dx, (float*)ws, B, C, HW, stream);
    if (rc != HK_OK) return rc;
    const int slots = (C + 63) / 64;
    hipLaunchKernelGGL(bcnn_tdot_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ta, tb, tc, (float*)ws, K, slots);
    HK_LAUNCH_CHECK();
    return hk_bcnn_bwd_rank1(dx, (const float*)ws, inv_norm, colsum, B, C, HW, stream);
}

// ------------------------------------------------------------------ signed-sqrt variant
extern "C" size_t hk_bcnn_ssqrt_ws_bytes(int B, int C, int HW) {
    (void)C; (void)HW;
    return (size_t)B * (SS_CHUNKS + 1) * sizeof(float) + 256;       // partial sums + sqrt(inv_norm) of the unscaled backward
}

static int ssqrt_pool_fwd_impl(const float* x, float* y, float* inv_norm, int B, int C, int HW, void* ws, size_t ws_bytes,
                               hk_stream_t stream, bool scale) {
    if (!x || !y || !inv_norm || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_bcnn_ssqrt_ws_bytes(B, C, HW)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)C * C;
    if (!force_generic()) {        // two passes: the Gram kernel applies the signed sqrt and sums u^2; then the scale
        int nparts = 0;
        const int rc2 = gram_fast_ssqrt(x, y, (float*)ws, &nparts, B, C, HW, st);
        if (rc2 == HK_OK) {
            if (scale) hipLaunchKernelGGL(ssqrt_scale_kernel, dim3(SS_CHUNKS, B), dim3(256), 0, st, y, (const float*)ws, inv_norm, n, nparts);
            else hipLaunchKernelGGL(ssqrt_norm_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, inv_norm, B, nparts);
            HK_LAUNCH_CHECK();
            return HK_OK;
        }
        if (rc2 != HK_ERR_UNSUPPORTED) return rc2;
    }
    int rc = force_generic() ? HK_ERR_UNSUPPORTED : gram_fast_raw(x, nullptr, 1.0f / (float)HW, y, B, C, HW, st);
    if (rc == HK_ERR_UNSUPPORTED) {
        const LdPlain xa = make_plain(x, (long long)C * HW, HW, C, HW);
        rc = bgemm_launch<true, true>(xa, xa, make_affine(y, (long long)C * C, C, 1.0f / (float)HW, nullptr, 0.f, 0.f), C, C,
                                      HW, B, st);
    }
    if (rc != HK_OK) return rc;
    hipLaunchKernelGGL(ssqrt_apply_kernel, dim3(SS_CHUNKS, B), dim3(256), 0, st, y, (float*)ws, n);
    HK_LAUNCH_CHECK();
    if (scale) hipLaunchKernelGGL(ssqrt_scale_kernel, dim3(SS_CHUNKS, B), dim3(256), 0, st, y, (const float*)ws, inv_norm, n, SS_CHUNKS);
    else hipLaunchKernelGGL(ssqrt_norm_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, inv_norm, B, SS_CHUNKS);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bcnn_ssqrt_pool_fwd(const float* x, float* y, float* inv_norm, int B, int C, int HW, void* ws,
                                      size_t ws_bytes, hk_stream_t stream) {
    return ssqrt_pool_fwd_impl(x, y, inv_norm, B, C, HW, ws, ws_bytes, stream, true);
}

extern "C" int hk_bcnn_ssqrt_pool_fwd_unscaled(const float* x, float* u, float* inv_norm, int B, int C, int HW, void* ws,
                                               size_t ws_bytes, hk_stream_t stream) {
    return ssqrt_pool_fwd_impl(x, u, inv_norm, B, C, HW, ws, ws_bytes, stream, false);
}

// The same u with the norm left ENTIRELY to the consumer: ONE launch - the Gram kernel with the signed sqrt in its epilogue -
// that writes u and, per image, *nparts partial sums of u^2 (ss_part [B][64], fixed order); hk_linear_fwd_ssq adds them up
// in its reduce launch (ssqrt_norm_kernel's order: the same inv_norm bits) and applies 1 / |u| to the logits.
// HK_ERR_UNSUPPORTED - nothing launched - outside the panel kernel's shapes: take hk_bcnn_ssqrt_pool_fwd_unscaled.
extern "C" int hk_bcnn_ssqrt_pool_fwd_parts(const float* x, float* u, float* ss_part, int* nparts, int B, int C, int HW,
                                            hk_stream_t stream) {
    if (!x || !u || !ss_part || !nparts || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (force_generic()) return HK_ERR_UNSUPPORTED;
    return gram_fast_ssqrt(x, u, ss_part, nparts, B, C, HW, (hipStream_t)stream);
}

static int ssqrt_pool_bwd_impl(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, int B, int C,
                               int HW, void* ws, size_t ws_bytes, hk_stream_t stream, bool unscaled) {
    if (!x || !y || !dy || !inv_norm || !dx || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_bcnn_ssqrt_ws_bytes(B, C, HW)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* tpart = (float*)ws;
    if (unscaled) {                              // y is u: t partials scaled by inv^2, sqrt(inv) in the place of inv
        float* sq = tpart + (size_t)B * SS_CHUNKS;
        hipLaunchKernelGGL(ssqrt_sqrt_inv_kernel, dim3((B + 63) / 64), dim3(64), 0, st, inv_norm, sq, B);
        HK_LAUNCH_CHECK();
        hipLaunchKernelGGL(dot_partial_kernel, dim3(SS_CHUNKS, B), dim3(256), 0, st, y, dy, tpart, (long long)C * C, inv_norm);
        HK_LAUNCH_CHECK();
        inv_norm = sq;
    } else {
        hipLaunchKernelGGL(dot_partial_kernel, dim3(SS_CHUNKS, B), dim3(256), 0, st, y, dy, tpart, (long long)C * C, (const float*)nullptr);
        HK_LAUNCH_CHECK();
    }
    if (!force_generic()) {
        const int rc = bcnn_ssqrt_fast_bwd(x, y, dy, inv_norm, tpart, SS_CHUNKS, dx, B, C, HW, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    LdBcnnP pa;
    pa.y = y; pa.dy = dy; pa.inv_norm = inv_norm; pa.tpart = nullptr; pa.C = C;
    pa.inv2m = 1.0f / (2.0f * (float)HW);
    pa.vec = (aligned16(y) && aligned16(dy) && (C % 4 == 0)) ? 1 : 0;
    pa.coef = 0.f; pa.tacc = 0.f; pa.active = 0;
    pa.tsum = tpart; pa.nt = SS_CHUNKS; pa.t2 = 0.f;
    const LdPlain xb = make_plain(x, (long long)C * HW, HW, C, HW);
    return bgemm_launch<true, false>(pa, xb, make_affine(dx, (long long)C * HW, HW, 1.0f, nullptr, 0.f, 0.f), C, HW, C, B, st);
}

// The signed-sqrt backward when dy = g W comes from a linear layer on the (normalised) pooled vector: t = <y, dy> =
// sum_k ta[b,k] (tb[b,k] - tc[k]) (ta = g, tb = logits, tc = bias) - the 2 x 4 C^2 bytes per image pass for its partial sums
// (dot_partial_kernel) is not launched.  unscaled != 0: `y` is the un-normalised u of hk_bcnn_ssqrt_pool_fwd_unscaled and dy
// the gradient hk_linear_bwd_scaled returned.  Shapes the one-launch kernel does not serve fall back to the entry points
// above (which add up y * dy themselves).
extern "C" int hk_bcnn_ssqrt_pool_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm,
                                           const float* ta, const float* tb, const float* tc, int K, int unscaled, float* dx,
                                           int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !y || !dy || !inv_norm || !ta || !tb || !dx || K <= 0 || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_bcnn_ssqrt_ws_bytes(B, C, HW)) return HK_ERR_WORKSPACE;
    if (!force_generic() && tuning().bwd_fold >= 0) {
        hipStream_t st = (hipStream_t)stream;
        // (unscaled: (dy + dy^T - 2 t inv u) / (inv |u|) inv^2 / 2M = (dy + dy^T - 2 (t inv) u) / |u| inv / 2M - the kernel takes
        //  the true inv_norm and applies it once to the coefficient and once to t: one launch, nothing in front of it)
        const int rc = bcnn_ssqrt_fast_bwd_tdot(x, y, dy, inv_norm, ta, tb, tc, K, unscaled ? 1 : 0, dx, B, C, HW, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    return ssqrt_pool_bwd_impl(x, y, dy, inv_norm, dx, B, C, HW, ws, ws_bytes, stream, unscaled != 0);
}

extern "C" int hk_bcnn_ssqrt_pool_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx,
                                      int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    return ssqrt_pool_bwd_impl(x, y, dy, inv_norm, dx, B, C, HW, ws, ws_bytes, stream, false);
}

extern "C" int hk_bcnn_ssqrt_pool_bwd_unscaled(const float* x, const float* u, const float* dy, const float* inv_norm, float* dx,
                                               int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    return ssqrt_pool_bwd_impl(x, u, dy, inv_norm, dx, B, C, HW, ws, ws_bytes, stream, true);
}
