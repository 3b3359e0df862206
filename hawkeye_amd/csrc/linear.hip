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
#include <type_traits>

#include "hk_bgemm.h"
#include "hk_linear_bwd.h"
#include "hk_linear_fwd.h"
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

// out[e] = bias[e % K] + rs[e / K] * sum_s part[s][e]   (e over B*K; slabs added as 4 interleaved chains combined in fixed
// order; rs: optional per-sample scale - the 1 / |z| of a pooled vector handed over unnormalised)
// ssq (optional, [B][64]): the row scale is not given but formed here - 1 / max(sqrt(sum of the sample's nssq partial sums of
// u^2), 1e-12), added in ssqrt_norm_kernel's order - and written to rs_out [B] by the thread that holds output (b, 0).
__global__ __launch_bounds__(256) void linear_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           const float* __restrict__ rs, float* __restrict__ out, int BK,
                                                           int K, int S, const float* __restrict__ ssq = nullptr, int nssq = 0,
                                                           float* __restrict__ rs_out = nullptr,
                                                           const float* __restrict__ ty = nullptr,
                                                           const float* __restrict__ tw = nullptr, int J = 0, int Jt0 = 0) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;
    float s = 0.f;
    if (e < BK) {
        // sixteen slabs of this chain in flight at a time (one dependent load per add was a chain of 64 L2 latencies:
        // 7 us for 13 MB at the BCNN shape); the adds keep their order q = g, g + 4, ..
        const float* pp = part + e;
        for (int q0 = g; q0 < S; q0 += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = q0 + 4 * u;
                v[u] = q < S ? pp[(long long)q * BK] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
    }
    red[g][l] = s;
    __syncthreads();
    if (g == 0 && e < BK) {
        float sum = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        // the features [Jt0, J) behind the last whole 32-feature chunk (J % 32 != 0: CBCNN's 6000 = 187 x 32 + 16), which the
        // slab kernel does not touch: at most 31 products per output, added here in feature order
        if (ty) {
            const float* yr = ty + (long long)(e / K) * J;
            const float* wr = tw + (long long)(e % K) * J;
            float t = 0.f;
            for (int j = Jt0; j < J; ++j) t += yr[j] * wr[j];
            sum += t;
        }
        float sc = rs ? rs[e / K] : 1.0f;
        if (ssq) {
            float t = 0.f;
            for (int c = 0; c < nssq; ++c) t += ssq[(long long)(e / K) * 64 + c];
            sc = 1.0f / fmaxf(sqrtf(t), 1e-12f);
            if (e % K == 0) rs_out[e / K] = sc;
        }
        out[e] = ((rs || ssq) ? sc * sum : sum) + (bias ? bias[e % K] : 0.f);
    }
}

// db[k] = sum_b g[b][k]: one wave per class, lanes over the samples, fixed shuffle tree (one thread per class looping over
// 64 dependent loads took 17 us)
__global__ __launch_bounds__(256) void linear_bias_grad_kernel(const float* __restrict__ g, float* __restrict__ db, int B,
                                                              int K) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= K) return;
    float s = 0.f;
    for (int b = lane; b < B; b += 64) s += g[(long long)b * K + k];
    s = wave_sum(s);
    if (lane == 0) db[k] = s;
}

// The features [J0, J) behind the last whole 64-feature chunk of linear_bwd64_kernel (J % 64 != 0: CBCNN's 6000 = 93 x 64 + 48):
// one workgroup per output row - the B rows of dy, then the K rows of dW - thread (j, part): tail feature j, every 16th term of
// the sum starting at `part` (all of a thread's loads in flight at once: a single thread walking 200 rows of W 24 KB apart
// took 20 us), the 16 parts added in order.  At most 63 features, so that the whole classifier stays on the one-launch kernel.
__global__ __launch_bounds__(1024) void linear_bwd_tail_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                               const float* __restrict__ y, float* __restrict__ dy,
                                                               float* __restrict__ dw, const float* __restrict__ row_scale, int B,
                                                               int J, int K, int J0) {
    __shared__ float red[16][64];
    const int r = blockIdx.x, jl = threadIdx.x & 63, part = threadIdx.x >> 6, j = J0 + jl;
    const bool is_dy = r < B;
    if ((is_dy && !dy) || (!is_dy && !dw)) return;                    // (uniform)
    const int k = r - B;
    const int n = is_dy ? K : B;                                       // terms of the sum
    float s = 0.f;
    if (j < J) {
        for (int t0 = part; t0 < n; t0 += 16 * 8) {
            float a[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 16 * u;
                const bool ok = t < n;
                const int tc = ok ? t : 0;
                if (is_dy) { a[u] = g[(long long)r * K + tc]; v[u] = w[(long long)tc * J + j]; }
                else { a[u] = row_scale ? row_scale[tc] * g[(long long)tc * K + k] : g[(long long)tc * K + k]; v[u] = y[(long long)tc * J + j]; }
                if (!ok) a[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a[u] * v[u];
        }
    }
    red[part][jl] = s;
    __syncthreads();
    if (part == 0 && j < J) {
        float t = red[0][jl];
        for (int q = 1; q < 16; ++q) t += red[q][jl];
        if (is_dy) dy[(long long)r * J + j] = t;
        else dw[(long long)k * J + j] = t;
    }
}

// the wide-classifier plan: slabs of KS features for linear_skinny_kernel; false when the generic path serves the shape
static inline bool skinny_plan(int B, int J, int K, int& KS, int& S, int& nt, int& ngrp, int& nrg) {
    if (tuning().linear_slabs < 0) return false;                 // knob: -1 forces the generic split-K path
    if (J % 4 != 0) return false;                                // (rows must start on 16 bytes for the LDS-DMA pieces; J % 32 != 0:
                                                                 //  the whole chunks here, the tail in linear_reduce_kernel)
    // up to 16 samples (OSME, N = 10): one 16-row tile and 16 class tiles per workgroup (with the 64-row instance the
    // idle rows are matrix-pipe time: 273 us against 203 us on the generic path).  17-32 samples: generic path.
    const bool small = B <= 16;
    nt = small ? 16 : (K <= 208 ? 13 : 15);
    ngrp = (K + nt * 16 - 1) / (nt * 16);
    nrg = small ? 1 : (B + 63) / 64;
    const long long chunks = (long long)(J / 32) * ngrp * nrg;   // chunk-tasks in all
    // Not for: fewer than 128 chunks (J < 4096: nothing to split); 17-32 samples (fewer than half of the 64 sample rows in
    // use).  A forced slab count also forces this path (tests).
    // Round 6: the lower limit was 4096 chunks - MPN's 64 x 32896 -> 200 (1028 chunks) ran the generic split-K tiles at
    // 30.5 us; on this kernel with one slab per CU it is 21.4 us (tools/r6_lab.py lin_mpn_sweep: 86 slabs 29.6, 129: 24.0,
    // 172: 21.8, 206: 21.4, 257 - a second wave of workgroups - 30.1), the yaml batch of 8: 26.0 -> 12.2 us, 64 x 8192 -> 200:
    // 17.3 -> 13.0 us with two chunks per slab (one chunk per slab: 15.8 - the pipeline never starts).
    if ((chunks < 128 || (B > 16 && B < 33)) && tuning().linear_slabs <= 0) return false;
    long long want = 256 / ((long long)ngrp * nrg);              // one workgroup per CU
    if (want < 1) want = 1;
    long long kc = (J / 32 + want - 1) / want;                   // chunks per slab
    if (kc < 2) kc = 2;
    if (tuning().linear_slabs > 0) kc = (J / 32 + tuning().linear_slabs - 1) / tuning().linear_slabs;
    KS = (int)(kc * 32);
    S = (int)((J / 32 + kc - 1) / kc);
    return true;
}

static inline void slab_plan(int B, int J, int K, int& KS, int& S) {
    int nt, ngrp, nrg;
    if (skinny_plan(B, J, K, KS, S, nt, ngrp, nrg)) return;
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

static int linear_fwd_impl(const float* y, const float* w, const float* bias, const float* row_scale, const float* ssq, int nssq,
                           float* rs_out, float* out, int B, int J, int K, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!y || !w || !out || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_linear_ws_bytes(B, J, K)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int KS, S;
    slab_plan(B, J, K, KS, S);
    float* part = (float*)ws;
    int nt, ngrp, nrg;
    if (aligned16(y) && aligned16(w) && skinny_plan(B, J, K, KS, S, nt, ngrp, nrg)) {
        const dim3 grid(xcd_grid(S, ngrp), nrg);
        const int walk = tuning().lin_walk < 0 ? 1 : tuning().lin_walk;          // interleaved chunks (see linear_skinny_kernel)
        const size_t lds = (size_t)4 * ((nt == 16 ? 16 : 64) * 32 + nt * 16 * 32) * sizeof(float);
        if (nt == 13) HK_ALLOW_BIG_LDS((&linear_skinny_kernel<13, 4>), lds);
        else if (nt == 15) HK_ALLOW_BIG_LDS((&linear_skinny_kernel<15, 4>), lds);
        else HK_ALLOW_BIG_LDS((&linear_skinny_kernel<16, 1>), lds);
        if (nt == 13)
            hipLaunchKernelGGL((linear_skinny_kernel<13, 4>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp, walk);
        else if (nt == 15)
            hipLaunchKernelGGL((linear_skinny_kernel<15, 4>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp, walk);
        else
            hipLaunchKernelGGL((linear_skinny_kernel<16, 1>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp, walk);
        HK_LAUNCH_CHECK();
        const int BK = B * K;
        const int Jt0 = (J / 32) * 32;                       // first feature of the tail (== J: none)
        hipLaunchKernelGGL(linear_reduce_kernel, dim3((BK + 63) / 64), dim3(256), 0, st, (const float*)part, bias, row_scale, out, BK, K, S, ssq, nssq, rs_out,
                           Jt0 < J ? y : (const float*)nullptr, w, J, Jt0);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    LdSlab la, lb;
    la.p = y; la.ld = J; la.R = B; la.J = J; la.KS = KS; la.vec = (aligned16(y) && J % 4 == 0) ? 1 : 0;
    lb.p = w; lb.ld = J; lb.R = K; lb.J = J; lb.KS = KS; lb.vec = (aligned16(w) && J % 4 == 0) ? 1 : 0;
    const EpAffine ep = make_affine(part, (long long)B * K, K, 1.0f, nullptr, 0.f, 0.f);
    const int rc = bgemm_launch<true, true>(la, lb, ep, B, K, KS, S, st);     // slab = batch ; A [B][KS], B as [K][KS]
    if (rc != HK_OK) return rc;
    const int BK = B * K;
    hipLaunchKernelGGL(linear_reduce_kernel, dim3((BK + 63) / 64), dim3(256), 0, st, (const float*)part, bias, row_scale, out, BK, K, S, ssq, nssq, rs_out);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_linear_fwd_scaled(const float* y, const float* w, const float* bias, const float* row_scale, float* out,
                                    int B, int J, int K, void* ws, size_t ws_bytes, hk_stream_t stream) {
    return linear_fwd_impl(y, w, bias, row_scale, nullptr, 0, nullptr, out, B, J, K, ws, ws_bytes, stream);
}

// out = (1 / |u|) (u W^T) + bias with |u|^2 handed over as nparts partial sums per sample (ss_part [B][64], from
// hk_bcnn_ssqrt_pool_fwd_parts); inv_norm [B] is WRITTEN (the backward's operand).
extern "C" int hk_linear_fwd_ssq(const float* u, const float* w, const float* bias, const float* ss_part, int nparts,
                                 float* inv_norm, float* out, int B, int J, int K, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!ss_part || !inv_norm || nparts <= 0 || nparts > 64) return HK_ERR_BAD_ARG;
    return linear_fwd_impl(u, w, bias, nullptr, ss_part, nparts, inv_norm, out, B, J, K, ws, ws_bytes, stream);
}

extern "C" int hk_linear_fwd(const float* y, const float* w, const float* bias, float* out, int B, int J, int K, void* ws,
                             size_t ws_bytes, hk_stream_t stream) {
    return hk_linear_fwd_scaled(y, w, bias, nullptr, out, B, J, K, ws, ws_bytes, stream);
}

// row_scale != nullptr: dW = (row_scale g)^T y (see linear_bwd64_kernel); only the one-launch kernel supports it -
// HK_ERR_UNSUPPORTED (nothing launched) for every other shape: the caller then normalises y itself
static int linear_bwd_impl(const float* y, const float* w, const float* g, const float* row_scale, float* dy, float* dw,
                           float* db, int B, int J, int K, hk_stream_t stream) {
    if (!y || !w || !g || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // wide classifier, up to 64 samples and 208 classes: both products in one launch of linear_bwd64_kernel (the knob
    // linear_slabs = -1 keeps the generic tiles)
    // (the kernels address W / dW [K][J] and y / dy [B][J] through 32-bit byte offsets of buffer descriptors: both must
    //  stay below 2^30 elements - a wider problem takes the generic tiles)
    // (round 6: from 4096 features up - it was 16384 - and any J % 4 == 0: the features behind the last whole chunk are
    //  linear_bwd_tail_kernel's.  CBCNN's 6000 -> 200 ran two generic-tile launches + the bias gradient: 19.3 / 24.0 us at
    //  B = 16 / 64)
    if (B <= 64 && K <= 208 && J % 4 == 0 && (long long)J >= 4096 && (long long)K * J < (1ll << 30) &&
        (long long)B * J < (1ll << 30) && (dy || dw) &&
        tuning().linear_slabs >= 0 && aligned16(y) && aligned16(w) && aligned16(dy) && aligned16(dw)) {
        const int nchunk = J / 64;
        const int CPS = (nchunk + 255) / 256, S = (nchunk + CPS - 1) / CPS;      // one workgroup per CU
        const int walk = tuning().lin_walk < 0 ? 1 : tuning().lin_walk;      // interleaved chunks: 141 vs 153 us at BCNN's shape
        const int nks = (K + 3) / 4 == 50 ? 50 : 52;
        const size_t ldsb = (size_t)4 * (nks / 2 + 8) * 1024;        // four stages of half a chunk
#define HK_LAUNCH_BWD64(NKS_, MODE_)                                                                           \
    do {                                                                                                       \
        HK_ALLOW_BIG_LDS((&linear_bwd64_kernel<NKS_, MODE_>), ldsb);                                           \
        hipLaunchKernelGGL((linear_bwd64_kernel<NKS_, MODE_>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, \
                           S, walk, row_scale);                                                                \
    } while (0)
        const int mode = dy && dw ? 0 : (dy ? 1 : 2);            // both products / dy only / dW only
        if (nks == 50) {
            if (mode == 0) HK_LAUNCH_BWD64(50, 0); else if (mode == 1) HK_LAUNCH_BWD64(50, 1); else HK_LAUNCH_BWD64(50, 2);
        } else {
            if (mode == 0) HK_LAUNCH_BWD64(52, 0); else if (mode == 1) HK_LAUNCH_BWD64(52, 1); else HK_LAUNCH_BWD64(52, 2);
        }
#undef HK_LAUNCH_BWD64
        HK_LAUNCH_CHECK();
        if (J % 64 != 0) {
            hipLaunchKernelGGL(linear_bwd_tail_kernel, dim3(B + K), dim3(1024), 0, st, g, w, y, dy, dw, row_scale, B, J, K, nchunk * 64);
            HK_LAUNCH_CHECK();
        }
        if (db && !dw) {                                  // (db rides on the dW role)
            hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((K + 3) / 4), dim3(256), 0, st, g, db, B, K);
            HK_LAUNCH_CHECK();
        }
        return HK_OK;
    }
    if (row_scale) return HK_ERR_UNSUPPORTED;
    // up to 16 samples, up to 1024 outputs (OSME): linear_bwd16_kernel, a pure stream of W / dW
    if (B <= 16 && K <= 1024 && K >= 256 && J % 64 == 0 && (long long)J >= 16384 && (long long)K * J < (1ll << 30) &&
        (long long)B * J < (1ll << 30) && (dy || dw) && tuning().linear_slabs >= 0 &&
        aligned16(y) && aligned16(w) && aligned16(dy) && aligned16(dw)) {
        const int nchunk = J / 64;
        const int CPS = (nchunk + 255) / 256, S = (nchunk + CPS - 1) / CPS;
        const int walk = tuning().lin_walk < 0 ? 0 : tuning().lin_walk;      // contiguous slabs: 154-174 vs 164-188 us at OSME's shape
        const size_t ldsb = (size_t)2 * 8 * 1024 * sizeof(float);
        if (dy && dw) hipLaunchKernelGGL((linear_bwd16_kernel<true, true>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S, walk);
        else if (dy) hipLaunchKernelGGL((linear_bwd16_kernel<true, false>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S, walk);
        else hipLaunchKernelGGL((linear_bwd16_kernel<false, true>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S, walk);
        HK_LAUNCH_CHECK();
        if (db && !dw) {
            hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((K + 3) / 4), dim3(256), 0, st, g, db, B, K);
            HK_LAUNCH_CHECK();
        }
        return HK_OK;
    }
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
        hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((K + 3) / 4), dim3(256), 0, st, g, db, B, K);
        HK_LAUNCH_CHECK();
    }
    return HK_OK;
}

extern "C" int hk_linear_bwd(const float* y, const float* w, const float* g, float* dy, float* dw, float* db, int B, int J,
                             int K, hk_stream_t stream) {
    return linear_bwd_impl(y, w, g, nullptr, dy, dw, db, B, J, K, stream);
}

extern "C" int hk_linear_bwd_scaled(const float* y, const float* w, const float* g, const float* row_scale, float* dy, float* dw,
                                    float* db, int B, int J, int K, hk_stream_t stream) {
    if (!row_scale) return HK_ERR_BAD_ARG;
    return linear_bwd_impl(y, w, g, row_scale, dy, dw, db, B, J, K, stream);
}
