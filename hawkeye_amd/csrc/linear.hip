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
    if (g == 0 && e < BK) out[e] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) + (bias ? bias[e % K] : 0.f);
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

// ---------------------------------------------------------------------------------------------------------------------
// Forward for the WIDE classifiers (BCNN 262 144 -> 200, OSME 100 352 -> 1024): a workgroup owns one slab of features and
// ALL of its (up to 64) samples x a group of NT 16-column class tiles, so y is read once and W once, through LDS-DMA.
// The generic split-K path above runs at 133-144 us on the BCNN shape = 1.95 TB/s (profiles/r2_pool_kernels_pmc.csv:
// 53 % of its wave time parked on loads, y fetched by four class tiles, one chunk of register prefetch); the product is
// balanced between the matrix pipe and HBM (24 FLOP/B), so both have to be kept busy:
//   * 512 threads = 8 waves: wave w owns samples 16 (w & 3) .. + 15 and the class tiles of half w >> 2 (7 + 6 of 13, or
//     8 + 8 of 16), 16x16x4 MFMA, A operand = y rows, B operand = W rows - both tiles are [row][32 features] exactly as
//     they lie in memory, read back with ds_read_b128 through the XOR swizzle of hk_bwd128d.h (slot row * 8 + (k4 ^ (row & 7)));
//   * chunks of 32 features, FOUR LDS stages (4 x 34.8 KB for 13 class tiles, 4 x 38.9 KB for 15), the pieces of chunk
//     c + 3 are issued during chunk c; the barrier that ends a chunk waits with s_waitcnt vmcnt(n) for everything but
//     the n pieces the wave has just issued, so a piece has two whole chunks to arrive;
//   * the fragments of chunk c + 1 (complete one barrier earlier) are read behind the last MFMAs of chunk c.
// Partial results [S][B][K] as before, added in slab order by linear_reduce_kernel: deterministic.

// MT: 16-sample row tiles per workgroup.  4: up to 64 samples, wave w owns row tile w & 3 and one half of the NT class tiles.
// 1: up to 16 samples (OSME: N = 10) - every wave owns the same 16 rows and NT / 8 of the class tiles; the product is then
// a pure stream of W (2 KB of LDS-DMA pieces per MFMA-cycle-pair), the matrix pipe idles.
template <int NT, int MT>
__global__ __launch_bounds__(512, 2) void linear_skinny_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                               float* __restrict__ part, int B, int J, int K, int KS,
                                                               int S, int ngrp) {
    static_assert(MT == 4 || (MT == 1 && NT % 8 == 0), "one row tile: the class tiles are dealt to the eight waves");
    constexpr int NH = MT == 4 ? (NT + 1) / 2 : NT / 8;   // class tiles per wave (at most)
    constexpr int NS = 4;                                // LDS stages (chunk c + 1 must be complete one barrier early: >= 4)
    constexpr int CH = 32;                               // features per chunk
    constexpr int MR = 16 * MT;                          // sample rows per workgroup
    constexpr int A_SZ = MR * CH, B_SZ = NT * 16 * CH;   // floats
    constexpr int STAGE = A_SZ + B_SZ;
    constexpr int NPA = 2 * MT, NPB = NT * 2, NP = NPA + NPB; // 1 KB pieces per chunk: 8 rows x 32 floats each
    constexpr int PPW = (NP + 7) / 8;                    // pieces per wave (at most)
    HK_DYN_LDS16(lds);

    int slab, grp;
    if (!xcd_map(blockIdx.x, S, ngrp, slab, grp)) return;
    const int rg = blockIdx.y;                                  // group of MR samples
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int rb = MT == 4 ? (wave & 3) : 0, half = wave >> 2;
    const int nt0 = MT == 4 ? half * NH : wave * NH, nloc = MT == 4 ? (half ? NT - NH : NH) : NH;
    const long long f0 = (long long)slab * KS;                  // first feature of the slab
    const int nfeat = (J - f0) < KS ? (int)(J - f0) : KS;       // (a multiple of 32: J % 32 == 0, KS % 32 == 0)
    const int nch = nfeat / CH;

    // this lane's source offsets (floats, from y / w + f0 + 32 c) in the pieces its wave issues: piece p = wave + 8 u;
    // p < 8: sample rows 8 p .. 8 p + 7 (clamped to the last sample), else class rows 8 (p - 8) .. (clamped to K - 1);
    // LDS slot j = lane & 7 of row r holds the feature quad j ^ (r & 7)
    long long src[PPW];
    int npc = 0;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int p = wave + 8 * u;
        const int r8 = lane >> 3, q4 = 4 * ((lane & 7) ^ (r8 & 7));
        if (p < NPA) {
            int row = rg * MR + 8 * p + r8;
            row = row < B ? row : B - 1;
            src[u] = (long long)row * J + q4;
        } else {
            int n = grp * (NT * 16) + 8 * (p - NPA) + r8;
            n = n < K ? n : K - 1;
            src[u] = (long long)n * J + q4;
        }
        if (p < NP) ++npc;
    }
    // pieces of chunk c into stage st (float offset); part 0 / 1: first / second half of the wave's pieces
    auto dma = [&](int c, int st, int part) {
        const long long fo = f0 + (long long)c * CH;
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            if ((u < (PPW + 1) / 2) != (part == 0)) continue;
            const int p = wave + 8 * u;
            if (p < NP) glds16((p < NPA ? y : w) + src[u] + fo, lds + st + 256 * p);
        }
    };
    auto vm_barrier = [&](bool all) {
        if (all) HK_VM_BARRIER(0);
        else if (npc == 6) HK_VM_BARRIER(6);
        else if (npc == 5) HK_VM_BARRIER(5);
        else if (npc == 4) HK_VM_BARRIER(4);
        else if (npc == 3) HK_VM_BARRIER(3);
        else HK_VM_BARRIER(0);
    };

    f32x4 acc[NH];
#pragma unroll
    for (int n = 0; n < NH; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragments of feature step s (0 / 1) of the chunk in stage st: a = y[16 rb + l15][16 s + 4 lq ..+3],
    // b[n] = W[16 (nt0 + n) + l15][same features]
    const int arow = 16 * rb + l15;
    const int aoff = arow * CH, asw = arow & 7;
    const int boff = A_SZ + (16 * nt0 + l15) * CH, bsw = l15 & 7;      // (16 (nt0 + n) is a multiple of 8)
    // prologue: chunks 0 .. NS - 2 into stages 0 .. NS - 2
    for (int c = 0; c < NS - 1 && c < nch; ++c) { dma(c, c * STAGE, 0); dma(c, c * STAGE, 1); }
    vm_barrier(true);
    // The chunk loop, instantiated per number of class tiles of the wave (NL = 7 / 6 of 13, 8 / 7 of 15): with the tile
    // count a run-time value the eighth fragment read and every seventh MFMA sat behind a (uniform) branch in the middle
    // of the MFMA stream (82.4 -> 77.3 us at the BCNN shape in one alternating run, profiles/r3_lab_call25.json).
    // Within a half chunk the compiler places the seven reads of the NEXT fragments behind the last MFMAs of the
    // current ones and waits for them at once: the wave parks for one LDS latency per half chunk while the other wave
    // of its SIMD has the matrix pipe.  Pinning the reads ahead of the MFMAs (no wait left) measured SLOWER - 80.6 us
    // with the reads before the group, 88.1 us with the reads behind its first seven MFMAs.
    auto run = [&](auto nl_tag) {
        constexpr int NL = decltype(nl_tag)::value;
        auto frag = [&](int st, int s, f32x4& a, f32x4 (&b)[NL]) {
            const float* base = lds + st;
            a = *reinterpret_cast<const f32x4*>(base + aoff + (((4 * s + lq) ^ asw) << 2));
#pragma unroll
            for (int n = 0; n < NL; ++n)
                b[n] = *reinterpret_cast<const f32x4*>(base + boff + n * 16 * CH + (((4 * s + lq) ^ bsw) << 2));
        };
        auto mma = [&](const f32x4& a, const f32x4 (&b)[NL]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n][t], acc[n], 0, 0, 0);
        };
        f32x4 a0, a1, b0[NL], b1[NL];
        frag(0, 0, a0, b0);
        int cur = 0;                                             // stage of chunk c (float offset), nxt = chunk c + 1
        for (int c = 0; c < nch; ++c) {
            const int nxt = cur + STAGE < NS * STAGE ? cur + STAGE : 0;
            const int dst = cur >= STAGE ? cur - STAGE : (NS - 1) * STAGE;  // stage of chunk c - 1 = chunk c + NS - 1
            const bool load = c + NS - 1 < nch;                  // uniform
            frag(cur, 1, a1, b1);
            mma(a0, b0);
            if (load) dma(c + NS - 1, dst, 0);
            __builtin_amdgcn_sched_barrier(0);
            frag(c + 1 < nch ? nxt : cur, 0, a0, b0);            // complete and published by the previous barrier
            mma(a1, b1);
            if (load) dma(c + NS - 1, dst, 1);
            __builtin_amdgcn_sched_barrier(0);
            vm_barrier(!load);
            cur = nxt;
        }
    };
    if constexpr (MT == 4 && NT % 2 == 1) {
        if (nloc == NH) run(std::integral_constant<int, NH>{});
        else run(std::integral_constant<int, NH - 1>{});
    } else {
        run(std::integral_constant<int, NH>{});
    }

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* pb = part + (long long)slab * B * K;
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        const int col = grp * (NT * 16) + 16 * (nt0 + n) + l15;
        if (n < nloc && col < K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rg * MR + 16 * rb + 4 * lq + r;
                if (row < B) pb[(long long)row * K + col] = acc[n][r];
            }
        }
    }
}

// the wide-classifier plan: slabs of KS features for linear_skinny_kernel; false when the generic path serves the shape
static inline bool skinny_plan(int B, int J, int K, int& KS, int& S, int& nt, int& ngrp, int& nrg) {
    if (tuning().linear_slabs < 0) return false;                 // knob: -1 forces the generic split-K path
    if (J % 32 != 0) return false;
    // up to 16 samples (OSME, N = 10): one 16-row tile and 16 class tiles per workgroup (with the 64-row instance the
    // idle rows are matrix-pipe time: 273 us against 203 us on the generic path).  17-32 samples: generic path.
    const bool small = B <= 16;
    nt = small ? 16 : (K <= 208 ? 13 : 15);
    ngrp = (K + nt * 16 - 1) / (nt * 16);
    nrg = small ? 1 : (B + 63) / 64;
    const long long chunks = (long long)(J / 32) * ngrp * nrg;   // chunk-tasks in all
    // Not for: too little work to amortise a 4-stage pipeline per workgroup; 17-32 samples (fewer than half of the 64
    // sample rows in use).  A forced slab count also forces this path (tests).
    if ((chunks < 256 * 16 || (B > 16 && B < 33)) && tuning().linear_slabs <= 0) return false;
    long long want = 256 / ((long long)ngrp * nrg);              // one workgroup per CU
    if (want < 1) want = 1;
    if (tuning().linear_slabs > 0) want = tuning().linear_slabs;
    KS = (int)(((J / 32 + want - 1) / want) * 32);
    S = (J + KS - 1) / KS;
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

extern "C" int hk_linear_fwd(const float* y, const float* w, const float* bias, float* out, int B, int J, int K, void* ws,
                             size_t ws_bytes, hk_stream_t stream) {
    if (!y || !w || !out || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_linear_ws_bytes(B, J, K)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int KS, S;
    slab_plan(B, J, K, KS, S);
    float* part = (float*)ws;
    int nt, ngrp, nrg;
    if (aligned16(y) && aligned16(w) && skinny_plan(B, J, K, KS, S, nt, ngrp, nrg)) {
        const dim3 grid(xcd_grid(S, ngrp), nrg);
        const size_t lds = (size_t)4 * ((nt == 16 ? 16 : 64) * 32 + nt * 16 * 32) * sizeof(float);
        if (nt == 13) HK_ALLOW_BIG_LDS((&linear_skinny_kernel<13, 4>), lds);
        else if (nt == 15) HK_ALLOW_BIG_LDS((&linear_skinny_kernel<15, 4>), lds);
        else HK_ALLOW_BIG_LDS((&linear_skinny_kernel<16, 1>), lds);
        if (nt == 13)
            hipLaunchKernelGGL((linear_skinny_kernel<13, 4>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp);
        else if (nt == 15)
            hipLaunchKernelGGL((linear_skinny_kernel<15, 4>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp);
        else
            hipLaunchKernelGGL((linear_skinny_kernel<16, 1>), grid, dim3(512), lds, st, y, w, part, B, J, K, KS, S, ngrp);
        HK_LAUNCH_CHECK();
        const int BK = B * K;
        hipLaunchKernelGGL(linear_reduce_kernel, dim3((BK + 63) / 64), dim3(256), 0, st, (const float*)part, bias, out, BK, K, S);
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
    hipLaunchKernelGGL(linear_reduce_kernel, dim3((BK + 63) / 64), dim3(256), 0, st, (const float*)part, bias, out, BK, K, S);
    HK_LAUNCH_CHECK();
    return HK_OK;
}


extern "C" int hk_linear_bwd(const float* y, const float* w, const float* g, float* dy, float* dw, float* db, int B, int J,
                             int K, hk_stream_t stream) {
    if (!y || !w || !g || B <= 0 || J <= 0 || K <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // wide classifier, up to 64 samples and 208 classes: both products in one launch of linear_bwd64_kernel (the knob
    // linear_slabs = -1 keeps the generic tiles)
    if (B <= 64 && K <= 208 && J % 64 == 0 && (long long)J >= 16384 && (long long)K * J < (1ll << 30) && (dy || dw) &&
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
                           S, walk);                                                                           \
    } while (0)
        const int mode = dy && dw ? 0 : (dy ? 1 : 2);            // both products / dy only / dW only
        if (nks == 50) {
            if (mode == 0) HK_LAUNCH_BWD64(50, 0); else if (mode == 1) HK_LAUNCH_BWD64(50, 1); else HK_LAUNCH_BWD64(50, 2);
        } else {
            if (mode == 0) HK_LAUNCH_BWD64(52, 0); else if (mode == 1) HK_LAUNCH_BWD64(52, 1); else HK_LAUNCH_BWD64(52, 2);
        }
#undef HK_LAUNCH_BWD64
        HK_LAUNCH_CHECK();
        if (db && !dw) {                                  // (db rides on the dW role)
            hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((K + 3) / 4), dim3(256), 0, st, g, db, B, K);
            HK_LAUNCH_CHECK();
        }
        return HK_OK;
    }
    // up to 16 samples, up to 1024 outputs (OSME): linear_bwd16_kernel, a pure stream of W / dW
    if (B <= 16 && K <= 1024 && K >= 256 && J % 64 == 0 && (long long)J >= 16384 && (dy || dw) && tuning().linear_slabs >= 0 &&
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
