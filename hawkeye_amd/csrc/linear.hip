// Classifier GEMMs adjacent to the pooling heads: out = y W^T + bias for a very wide feature vector
// (BCNN: J = 512^2 = 262144 -> 200 classes; MPN: 32896 -> 200; OSME: 100352 -> 1024) and its backward.
// replaces nn.Linear at model/methods/BCNN.py:42,54, CBCNN.py:26,34, MPNCOV.py:31, OSME.py:34,43.
//
// All three products are HBM-bound at these shapes (BCNN, B = 64: W 209.7 MB + y 67.1 MB against 6.7 GFLOP), so the
// design goal is to stream W and y exactly once with enough workgroups in flight:
//   forward   split-K: the J axis is cut into S slabs, slab s is one "batch" of the f32-MFMA GEMM (hk_bgemm.h) whose
//             operand loader offsets both operands by s * KS; partial [S][B][K] results are added in slab order by a
//             second kernel (deterministic, no atomics) which also adds the bias.
//   dy = g W          M = B, N = J, K = classes : one 64x64 tile per 64 features, W read once, dy written once
//   dW = g^T y        M = classes, N = J, K = B : y read once per class tile row (L2), dW written once
//   db = sum_b g
#include <cstdlib>

#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// Operand [R][J] (row-major, J contiguous) seen as S independent [R][KS] slabs along J: batch index = slab.
struct LdSlab {
    const float* p;
    int ld, R, J, KS;
    int vec;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int s, int r, int c) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const long long gc = (long long)s * KS + c;
        if (r < R && c < KS && gc < J) {
            const float* q = p + (long long)r * ld + gc;
            const bool full = c + 3 < KS && gc + 3 < J;
            if (vec && full) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                v.x = q[0];
                if (c + 1 < KS && gc + 1 < J) v.y = q[1];
                if (c + 2 < KS && gc + 2 < J) v.z = q[2];
                if (c + 3 < KS && gc + 3 < J) v.w = q[3];
            }
        }
        return v;
    }
};

// out[e] = bias[e % K] + sum_s part[s][e]   (e over B*K; slabs added as 4 interleaved chains combined in fixed order)
__global__ __launch_bounds__(256) void linear_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           float* __restrict__ out, int BK, int K, int S) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;
    float s = 0.f;
    if (e < BK) {
        const float* pp = part + e;
#pragma unroll 4
        for (int q = g; q < S; q += 4) s += pp[(long long)q * BK];
    }
    red[g][l] = s;
    __syncthreads();
    if (g == 0 && e < BK) out[e] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) + (bias ? bias[e % K] : 0.f);
}

__global__ __launch_bounds__(256) void linear_bias_grad_kernel(const float* __restrict__ g, float* __restrict__ db, int B,
                                                              int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[(long long)b * K + k];
    db[k] = s;
}

static inline void slab_plan(int B, int J, int K, int& KS, int& S) {
    const long long tiles = (long long)((B + 63) / 64) * ((K + 63) / 64);
    // exactly ONE wave of workgroups over the chip: 256 CUs x 4 resident 64x64x32 workgroups = 1024.  Measured at the BCNN
    // shape (4 tiles; BENCH_r01 sweep): 256 slabs = 1024 workgroups 132 us; 128 -> 170, 384 (1.5 waves: the round-1
    // choice) 171, 768 -> 152, 1024 -> 162
    long long want = 1024 / tiles;
    if (tuning().linear_slabs > 0) want = tuning().linear_slabs;
    const long long max_s = (J + 255) / 256;                       // at least 256 features (8 K-chunks) per slab
    if (want > max_s) want = max_s;
    if (want < 1) want = 1;
    KS = (int)(((J + want - 1) / want + 31) / 32 * 32);
    S = (J + KS - 1) / KS;
}

}  // namespace hk

using namespace hk;

extern "C" size_t hk_linear_ws_bytes(int B, int J, int K) {
    if (B <= 0 || J <= 0 || K <= 0) return 0;
    int KS, S;
    slab_plan(B, J, K, KS, S);
    return (size_t)S * B * K * sizeof(float) + 256;
}

extern "C" int hk_linear_fwd(const float* y, const float* w, const float* bias, float* out, int B, int J, int K, void* ws,
                             size_t ws_bytes, hk_stream_t stream) {
    if (!y || !w || !out || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_linear_ws_bytes(B, J, K)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int KS, S;
    slab_plan(B, J, K, KS, S);
    float* part = (float*)ws;
    LdSlab la, lb;
    la.p = y; la.ld = J; la.R = B; la.J = J; la.KS = KS; la.vec = (aligned16(y) && J % 4 == 0) ? 1 : 0;
    lb.p = w; lb.ld = J; lb.R = K; lb.J = J; lb.KS = KS; lb.vec = (aligned16(w) && J % 4 == 0) ? 1 : 0;
    const EpAffine ep = make_affine(part, (long long)B * K, K, 1.0f, nullptr, 0.f, 0.f);
    const int rc = bgemm_launch<true, true>(la, lb, ep, B, K, KS, S, st);     // slab = batch ; A [B][KS], B as [K][KS]
    if (rc != HK_OK) return rc;
    const int BK = B * K;
    hipLaunchKernelGGL(linear_reduce_kernel, dim3((BK + 63) / 64), dim3(256), 0, st, (const float*)part, bias, out, BK, K, S);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_linear_bwd(const float* y, const float* w, const float* g, float* dy, float* dw, float* db, int B, int J,
                             int K, hk_stream_t stream) {
    if (!y || !w || !g || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dy) {      // dy [B][J] = g [B][K] W [K][J]
        const LdPlain la = make_plain(g, 0, K, B, K);
        const LdPlain lb = make_plain(w, 0, J, K, J);
        const EpAffine ep = make_affine(dy, 0, J, 1.0f, nullptr, 0.f, 0.f);
        const int rc = bgemm_launch<true, false>(la, lb, ep, B, J, K, 1, st);
        if (rc != HK_OK) return rc;
    }
    if (dw) {      // dW [K][J] = g^T [K][B] y [B][J] : both operands stored k-major (k = sample)
        const LdPlain la = make_plain(g, 0, K, B, K);
        int rc;
        if (J % 64 == 0) {
            // every block of 64 features is one "batch sample": the XCD-affine block map then runs its class tiles
            // back to back on ONE XCD, so the 64-feature slice of y is fetched from HBM once and re-read from that L2
            // (with a flat tile order the class tiles of a slice are 4096 workgroups apart: y would be read K/64 times)
            LdSlab lb;
            lb.p = y; lb.ld = J; lb.R = B; lb.J = J; lb.KS = 64; lb.vec = (aligned16(y) && J % 4 == 0) ? 1 : 0;
            rc = bgemm_launch<false, false>(la, lb, make_affine(dw, 64, J, 1.0f, nullptr, 0.f, 0.f), K, 64, B, J / 64, st);
        } else {
            const LdPlain lb = make_plain(y, 0, J, B, J);
            rc = bgemm_launch<false, false>(la, lb, make_affine(dw, 0, J, 1.0f, nullptr, 0.f, 0.f), K, J, B, 1, st);
        }
        if (rc != HK_OK) return rc;
    }
    if (db) {
        hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((K + 255) / 256), dim3(256), 0, st, g, db, B, K);
        HK_LAUNCH_CHECK();
    }
    return HK_OK;
}
