// Gram backward, second structure:  dX[I rows] = sum_K P(I,K) X(K)  for 128-row blocks I on ONE workgroup per CU.
//
// Why (profiles/r1c_sq_wait_counters.csv, BENCH_r01, profiles/r2_candidates.json): the 64-row kernel
// (bcnn_bwd_panel_kernel) and its two variants all sit at 85-93 us = 0.45-0.49 of the fp32 MFMA peak with the matrix
// pipe 66 % busy.  What they share:  (1) a P-tile BUILD phase between MFMA phases (transposing scatter of dy(K,I), 16
// v_rcp, three or four workgroup barriers per K-block),  (2) one ds_read_b32 of the X operand per MFMA (each wave
// owns only 16 rows), i.e. ~1 LDS wave-instruction per 32 matrix-pipe cycles from each of 8 waves,  (3) every X block
// is staged by 8 workgroups per image.  This kernel removes all three:
//   * no P tile: LDS holds the RAW tiles  S1 = dy(I,K) [i][k],  S2 = dy(K,I) [k][i],  W = coef / y(I,K) [i][k]  exactly
//     as they lie in HBM (coalesced 16-byte loads, 16-byte LDS stores, no transposition), and the MFMA A operand is
//     formed when the fragment is read:  a = (S1[i][k] + S2[k][i]) * W[i][k]  - for the 16x16x4 A layout lanes run
//     along i, so the "transposed" read of S2 is a conflict-free ds_read_b32 of consecutive addresses;
//   * each wave owns 32 rows x 13 column tiles (26 accumulators): an X fragment feeds two MFMAs, half the LDS reads;
//   * 128-row blocks: X is staged by 4 workgroups per image instead of 8;
//   * K-blocks of 32 channels, two LDS stages (2 x 77 KB), ONE barrier per K-block; the next block's global loads are
//     issued before the MFMAs of the current one and land in the other stage between its two halves.
// Same arithmetic per element as the 64-row kernel (sum, rcp, coef folded into W) - dX agrees to rounding, not bit for
// bit (W = rcp(y) * coef is rounded before the multiplication).
// MODE 0 BCNN   P = (dy + dy^T) / y * inv^2 / (2M)          MODE 1 COV   P = (g + g^T) / M, X centred
// MODE 2 CBP    P = dG + dG^T gathered from dc               MODE 3 signed-sqrt BCNN (BCNN.py:23-24): see bcnn_pool.hip
#pragma once
#include "hk_common.h"

namespace hk {

struct BwdExtra {
    const float* mu;     // [B][C]      (COV)
    const int* h1;       // [C]         (CBP)
    const int* h2;
    const float* s1;
    const float* s2;
    const float* dc;     // [B][D]
    int D;
    const float* tb;     // [B][nt]     (signed sqrt: partial sums of t = <y, dy>, added in order)
    int nt;
};

// t = <y, dy> of sample b from its partial sums (every workgroup adds them itself, fixed order)
__device__ __forceinline__ float bwd_t_of(const BwdExtra& ex, int b) {
    float t = 0.f;
    for (int c = 0; c < ex.nt; ++c) t += ex.tb[(long long)b * ex.nt + c];
    return t;
}

template <int HW, int MODE>
__global__ __launch_bounds__(256, 1) void bcnn_bwd128_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                             float* __restrict__ tpart, int C, int nI, int B,
                                                             BwdExtra ex) {
    constexpr int NT = (HW + 15) / 16;          // 16-column output tiles
    constexpr int KB = 32;                      // channels per K-block
    constexpr int P1 = KB + 4;                  // pitch of the [i][k] tiles (ds_read_b128: pitch / 4 odd)
    constexpr int P2 = 128 + 4;                 // pitch of the [k][i] tile
    constexpr bool HAS_W = MODE == 0 || MODE == 3;
    constexpr bool HAS_S2 = MODE != 2;
    constexpr int S1_SZ = 128 * P1, W_SZ = HAS_W ? 128 * P1 : 0, S2_SZ = HAS_S2 ? KB * P2 : 0;
    constexpr int XN4 = KB * HW / 4;            // float4 of one X block
    constexpr int NSX = (XN4 + 255) / 256;
    constexpr int X_SZ = (XN4 + 3) / 4 * 16;    // floats, rounded to 64 B
    constexpr int STAGE = S1_SZ + W_SZ + S2_SZ + X_SZ;
    HK_DYN_LDS16(lds);

    int b, I;
    if (!xcd_map(blockIdx.x, B, nI, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const long long cc = (long long)b * C * C;
    const float* xb = x + (long long)b * C * HW;
    const int nkb = C / KB;
    float coef = 1.0f / (float)HW;                         // COV
    if (HAS_W) {
        const float in = inv_norm[b];
        coef = in * in / (2.0f * (float)HW);
    }
    const float t2 = MODE == 3 ? 2.0f * bwd_t_of(ex, b) : 0.f;

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tacc = 0.f;

    // staging registers: named (an indexed array that is loaded and stored in different conditional blocks is not
    // promoted to registers)
    f32x4 ry0, ry1, ry2, ry3, rd0, rd1, rd2, rd3, rt0, rt1, rt2, rt3;
    f32x4 rx[NSX];
    float rmu[NSX];                                            // COV: channel mean of each staged X vector
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    ry0 = ry1 = ry2 = ry3 = rd0 = rd1 = rd2 = rd3 = rt0 = rt1 = rt2 = rt3 = z4;

    const int r1 = tid >> 3, c1 = 4 * (tid & 7);           // [i][k] tiles: row r1 + 32 u, k offset c1
    const int r2 = tid >> 5, c2 = 4 * (tid & 31);          // [k][i] tile : k row r2 + 8 u, i offset c2

    auto ld1 = [&](const float* base, int kb, int u) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(base + cc + (long long)(I * 128 + r1 + 32 * u) * C + kb * KB + c1);
    };
    auto ld2 = [&](int kb, int u) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(dy + cc + (long long)(kb * KB + r2 + 8 * u) * C + I * 128 + c2);
    };
    auto gather = [&](int kb, int u) -> f32x4 {               // CBP: P(I,K) from the dc vector (CBCNN.py backward)
        const int i = I * 128 + r1 + 32 * u;
        const int h1i = ex.h1[i], h2i = ex.h2[i];
        const float s1i = ex.s1[i], s2i = ex.s2[i];
        const float* dcb = ex.dc + (long long)b * ex.D;
        f32x4 p;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = kb * KB + c1 + t;
            int ba = h1i + ex.h2[k]; if (ba >= ex.D) ba -= ex.D;
            int bb = ex.h1[k] + h2i; if (bb >= ex.D) bb -= ex.D;
            p[t] = s1i * ex.s2[k] * dcb[ba] + ex.s1[k] * s2i * dcb[bb];
        }
        return p;
    };
#define HK_BW_GLOAD(kb)                                                                                        \
    do {                                                                                                       \
        if (HAS_W) { ry0 = ld1(y, kb, 0); ry1 = ld1(y, kb, 1); ry2 = ld1(y, kb, 2); ry3 = ld1(y, kb, 3); }     \
        if (MODE != 2) {                                                                                       \
            rd0 = ld1(dy, kb, 0); rd1 = ld1(dy, kb, 1); rd2 = ld1(dy, kb, 2); rd3 = ld1(dy, kb, 3);            \
            rt0 = ld2(kb, 0); rt1 = ld2(kb, 1); rt2 = ld2(kb, 2); rt3 = ld2(kb, 3);                            \
        } else {                                                                                               \
            rd0 = gather(kb, 0); rd1 = gather(kb, 1); rd2 = gather(kb, 2); rd3 = gather(kb, 3);                \
        }                                                                                                      \
        const f32x4* xs_ = reinterpret_cast<const f32x4*>(xb + (long long)(kb) * KB * HW);                     \
        _Pragma("unroll") for (int u = 0; u < NSX; ++u) {                                                      \
            const int f_ = tid + 256 * u, fc_ = f_ < XN4 ? f_ : XN4 - 1;                                       \
            rx[u] = xs_[fc_];                                                                                  \
            rmu[u] = MODE == 1 ? ex.mu[(long long)b * C + (kb) * KB + (4 * fc_) / HW] : 0.f;                   \
        }                                                                                                      \
    } while (0)

    // W = coef / y (v_rcp_f32, 1 ulp: parity budget 1e-4);  signed sqrt: |y| in the denominator, 0 where y == 0
    auto wof = [&](f32x4 yv) -> f32x4 {
        f32x4 w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 3) w[t] = yv[t] == 0.f ? 0.f : __builtin_amdgcn_rcpf(fabsf(yv[t])) * coef;
            else w[t] = __builtin_amdgcn_rcpf(yv[t]) * coef;
        }
        return w;
    };
    auto sst1 = [&](float* S1, float* Wt, int u, f32x4 yv, f32x4 dv) {
        const int o = (r1 + 32 * u) * P1 + c1;
        if (MODE == 1) dv *= coef;
        if (MODE == 3) dv -= t2 * yv;                          // (dy_ij + dy_ji - 2 t y_ij) / |y_ij|: the whole -2 t y term here
        *reinterpret_cast<f32x4*>(S1 + o) = dv;
        if (HAS_W) *reinterpret_cast<f32x4*>(Wt + o) = wof(yv);
        if (MODE == 0) tacc += (yv[0] * dv[0] + yv[1] * dv[1]) + (yv[2] * dv[2] + yv[3] * dv[3]);
    };
#define HK_BW_SSTORE(st)                                                                                       \
    do {                                                                                                       \
        float* S1_ = lds + (st) * STAGE;                                                                       \
        float* W_ = S1_ + S1_SZ;                                                                               \
        float* S2_ = W_ + W_SZ;                                                                                \
        float* X_ = S2_ + S2_SZ;                                                                               \
        sst1(S1_, W_, 0, ry0, rd0); sst1(S1_, W_, 1, ry1, rd1); sst1(S1_, W_, 2, ry2, rd2); sst1(S1_, W_, 3, ry3, rd3); \
        if (HAS_S2) {                                                                                          \
            const float sc_ = MODE == 1 ? coef : 1.0f;                                                         \
            *reinterpret_cast<f32x4*>(S2_ + (r2 + 0) * P2 + c2) = rt0 * sc_;                                   \
            *reinterpret_cast<f32x4*>(S2_ + (r2 + 8) * P2 + c2) = rt1 * sc_;                                   \
            *reinterpret_cast<f32x4*>(S2_ + (r2 + 16) * P2 + c2) = rt2 * sc_;                                  \
            *reinterpret_cast<f32x4*>(S2_ + (r2 + 24) * P2 + c2) = rt3 * sc_;                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int u = 0; u < NSX; ++u) {                                                      \
            const int f_ = tid + 256 * u;                                                                      \
            /* the mean is subtracted HERE, not after the load: touching the value there parks the wave on it */ \
            if (f_ < XN4) reinterpret_cast<f32x4*>(X_)[f_] = MODE == 1 ? rx[u] - rmu[u] : rx[u];               \
        }                                                                                                      \
    } while (0)

    HK_BW_GLOAD(0);
    HK_BW_SSTORE(0);
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        const bool more = kb + 1 < nkb;
        if (more) HK_BW_GLOAD(kb + 1);
        const float* S1 = lds + cur * STAGE;
        const float* Wt = S1 + S1_SZ;
        const float* S2 = Wt + W_SZ;
        const float* X = S2 + S2_SZ;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KB / 16; ++s) {
            // A fragments of the wave's two 16-row blocks for k = 16 s + 4 lq + t
            float a[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wave * 32 + i * 16 + l15;
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(S1 + row * P1 + 16 * s + 4 * lq);
                f32x4 wv = (f32x4){1.f, 1.f, 1.f, 1.f};
                if (HAS_W) wv = *reinterpret_cast<const f32x4*>(Wt + row * P1 + 16 * s + 4 * lq);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = d1[t];
                    if (HAS_S2) v += S2[(16 * s + 4 * lq + t) * P2 + row];
                    a[i][t] = HAS_W ? v * wv[t] : v;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* bp = X + (16 * s + 4 * lq + t) * HW + l15;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float bv = bp[16 * n];
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][t], bv, acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][t], bv, acc[1][n], 0, 0, 0);
                }
            }
            if (s == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) HK_BW_SSTORE(cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#undef HK_BW_GLOAD
#undef HK_BW_SSTORE

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float* dxb = dx + (long long)b * C * HW + (long long)(I * 128 + wave * 32 + i * 16 + lq * 4) * HW;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = 16 * n + l15;
            if (col < HW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dxb[(long long)r * HW + col] = acc[i][n][r];
            }
        }
    }
    if (MODE == 0) {                                           // t partials: slot 2I (+ a zero in 2I + 1: the consumer
        __syncthreads();                                       // adds C / 64 slots per image)
        const float tsum = block_sum<4>(tacc, lds);
        if (tid == 0) {
            tpart[(long long)b * (2 * nI) + 2 * I] = tsum;
            tpart[(long long)b * (2 * nI) + 2 * I + 1] = 0.f;
        }
    }
}

template <int HW, int MODE>
static inline size_t bwd128_lds_bytes() {
    constexpr bool HAS_W = MODE == 0 || MODE == 3;
    constexpr bool HAS_S2 = MODE != 2;
    constexpr int stage = 128 * 36 + (HAS_W ? 128 * 36 : 0) + (HAS_S2 ? 32 * 132 : 0) + ((32 * HW / 4 + 3) / 4 * 16);
    return (size_t)2 * stage * sizeof(float);
}

// HK_ERR_UNSUPPORTED unless C % 128 == 0 (the caller then takes the 64-row kernel)
template <int HW, int MODE>
static int bwd128_launch(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart,
                         int B, int C, const BwdExtra& ex, hipStream_t st) {
    if (C % 128 != 0) return HK_ERR_UNSUPPORTED;
    const size_t lds = bwd128_lds_bytes<HW, MODE>();
    static bool attr_set = false;                           // > 64 KB of dynamic LDS needs the opt-in
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bcnn_bwd128_kernel<HW, MODE>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nI = C / 128;
    hipLaunchKernelGGL((bcnn_bwd128_kernel<HW, MODE>), dim3(xcd_grid(B, nI)), dim3(256), lds, st, x, y, dy, inv_norm, dx,
                       tpart, C, nI, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
