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
#include "hk_bwd128d.h"      // glds16 (LDS-DMA)
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
#define HK_VMCNT_IMM(n) (((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))     /* gfx9 s_waitcnt: vmcnt only */
#define HK_VM_BARRIER(n)                                                                                       \
    do {                                                                                                       \
        asm volatile("" ::: "memory");                 /* no LDS access moves across */                        \
        __builtin_amdgcn_s_waitcnt(HK_VMCNT_IMM(n));                                                           \
        __builtin_amdgcn_s_barrier();                                                                          \
        asm volatile("" ::: "memory");                                                                         \
    } while (0)

#ifdef HK_LAB   // tools/linear_lab.py, timing only (results are wrong): bits 1 = no MFMAs, 2 = no LDS-DMA inside the loop, 4 = no fragment reads
__device__ int g_lin_lab = 0;
#endif
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
#ifdef HK_LAB
    const int labv = __builtin_amdgcn_readfirstlane(g_lin_lab);
#else
    constexpr int labv = 0;
#endif
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
            if (labv & 4) return;
            const float* base = lds + st;
            a = *reinterpret_cast<const f32x4*>(base + aoff + (((4 * s + lq) ^ asw) << 2));
#pragma unroll
            for (int n = 0; n < NL; ++n)
                b[n] = *reinterpret_cast<const f32x4*>(base + boff + n * 16 * CH + (((4 * s + lq) ^ bsw) << 2));
        };
        auto mma = [&](const f32x4& a, const f32x4 (&b)[NL]) {
            if (labv & 1) return;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n][t], acc[n], 0, 0, 0);
        };
        f32x4 a0, a1, b0[NL], b1[NL];
#ifdef HK_LAB
        a0 = a1 = (f32x4){1.f, 1.f, 1.f, 1.f};
        for (int n = 0; n < NL; ++n) b0[n] = b1[n] = (f32x4){1.f, 1.f, 1.f, 1.f};
#endif
        frag(0, 0, a0, b0);
        int cur = 0;                                             // stage of chunk c (float offset), nxt = chunk c + 1
        for (int c = 0; c < nch; ++c) {
            const int nxt = cur + STAGE < NS * STAGE ? cur + STAGE : 0;
            const int dst = cur >= STAGE ? cur - STAGE : (NS - 1) * STAGE;  // stage of chunk c - 1 = chunk c + NS - 1
            const bool load = c + NS - 1 < nch && !(labv & 2);   // uniform
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

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the wide classifier, up to 64 samples and up to 208 classes (BCNN 262 144 -> 200): BOTH products in ONE
// launch, one 8-wave workgroup per CU walking its slab of features in 64-feature chunks.
//     dy [B][J] = g W      (reads W, writes dy)          dW [K][J] = g^T y      (reads y, writes dW)
// The two products share nothing but g [B][K] (51 KB), so inside a workgroup they are two ROLES of four waves each, one
// wave of each role per SIMD: the W / dy stream and the y / dW stream run side by side (553 MB of combined traffic
// against two serial passes at 2.4-3.1 TB/s in round 3) and the matrix pipe of every SIMD always has a wave of the other
// role to issue from.  What makes the MFMA stream dense - 204 MFMAs per wave and chunk against 13 / 4 LDS-DMA pieces,
// 50 / 16 fragment reads, 4 / 16 stores and ONE barrier:
//   * g never moves: role dy keeps A = g[16 st + l15][4 s + lq] of its sample tile st for all NKS class steps in
//     registers, role dW keeps A = g^T of its three (+ a quarter of the thirteenth) class tiles for all 16 sample steps;
//   * the B operand is the streamed tile exactly as it lies in memory.  A tile row is 64 consecutive features; LDS-DMA
//     piece p holds rows 4 p .. 4 p + 3 (lane l: row 4 p + (l >> 4), features 4 (l & 15) ..+3), so the fragment of class /
//     sample step s is ONE linear ds_read_b128 at 1 KB s + 16 lane - no swizzle, no conflicts - whose four floats feed the
//     four MFMAs of the step: MFMA t computes the output columns {4 n + t}.  A lane's four accumulators therefore hold
//     four CONSECUTIVE features of a row and leave as 16-byte stores (256-byte runs per row) straight from registers;
//   * two LDS stages (2 x 66 KB).  The pieces of chunk c + 1 are issued behind the first MFMA groups of chunk c and have
//     the rest of the chunk (~6 us) to land; the barrier that ends chunk c waits for them (vmcnt(0): the stores of chunk
//     c - 1, issued right behind the previous barrier, are a whole chunk old by then) and publishes the stage.
// The last class tile (classes 192 ..) is dealt to the four dW waves by output column quarter (wave i: columns {4 n + i},
// 4-byte stores: 4 % of dW).  db = sum_b g falls out of the dW role's resident fragments in the workgroup of slab 0.
// A role whose result is not wanted (dy == nullptr: stage-1 training of the classifier alone) exits at once; finished
// waves do not take part in s_barrier.  Deterministic: no atomics, fixed summation order.
template <int NKS>
__global__ __launch_bounds__(512, 2) void linear_bwd64_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              const float* __restrict__ y, float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int J,
                                                              int K, int CPS, int S) {
    constexpr int CH = 64;                               // features per chunk
    constexpr int WP = NKS, YP = 16;                     // 1 KB pieces of the W tile [4 NKS][64] / the y tile [64][64]
    constexpr int STAGE = (WP + YP) * 256;               // floats
    constexpr int NPW = (WP + 3) / 4;                    // W pieces per dy wave (at most)
    HK_DYN_LDS16(lds);
    const int slab = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = slab * CPS;
    int nch = J / CH - c0;
    nch = nch < CPS ? nch : CPS;
    if (slab >= S || nch <= 0) return;
    const bool role_dy = wave < 4;                               // wave-uniform
    if (role_dy ? dy == nullptr : dw == nullptr) return;
    const long long f0 = (long long)c0 * CH;                     // first feature of the slab
    const float* Lf = lds + 4 * lane;                            // this lane's 16 bytes of a piece

#define HK_LB_WAIT_BARRIER() HK_VM_BARRIER(0)

    if (role_dy) {
        const int st = wave;                                     // sample tile
        // pieces p = wave + 4 u of the W tile: class row 4 p + lq (clamped to K - 1), features 4 l15 ..+3 (byte offsets
        // from the chunk's base: K J 4 < 4 GB is checked by the launcher)
        unsigned wo[NPW];
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            int c = 4 * (wave + 4 * u) + lq;
            c = c < K ? c : K - 1;
            wo[u] = 4u * ((unsigned)c * (unsigned)J + 4u * l15);
        }
        const char* wbase = reinterpret_cast<const char*>(w + f0);
        auto dma = [&](int c, int sto, int u) {                  // piece u of chunk c into the stage at float offset sto
            const int p = wave + 4 * u;                          // (4 u + 3 < WP folds at compile time: only the last u branches)
            if (4 * u + 3 < WP || p < WP)
                glds16(reinterpret_cast<const float*>(wbase + (long long)c * (CH * 4) + wo[u]), lds + sto + 256 * p);
        };
#pragma unroll
        for (int u = 0; u < NPW; ++u) dma(0, 0, u);
        // resident A fragments: ga[s] = g[16 st + l15][4 s + lq]  (zero beyond B samples / K classes)
        float ga[NKS];
        {
            const int b = 16 * st + l15;
            const float* gb = g + (long long)(b < B ? b : B - 1) * K;        // (unconditional loads + selects: no branches)
            // every load issued before the first use (clamped addresses; HK_PIN_LOADED keeps the compiler from sinking a
            // load into the select that follows it - it did, with an s_waitcnt vmcnt(0) per element: 50 L2 round trips)
#pragma unroll
            for (int s = 0; s < NKS; ++s) ga[s] = gb[4 * s + lq < K ? 4 * s + lq : K - 1];
#pragma unroll
            for (int s = 0; s < NKS; ++s) HK_PIN_LOADED(ga[s]);
#pragma unroll
            for (int s = 0; s < NKS; ++s) ga[s] = (b < B && 4 * s + lq < K) ? ga[s] : 0.f;
        }
        HK_LB_WAIT_BARRIER();
        auto chunk = [&](int c, int cur, auto load_tag) {
            constexpr bool LOAD = decltype(load_tag)::value;
            const int nxt = cur ? 0 : STAGE;
            const float* T = Lf + cur;
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 fr[2];                                        // fragment of step s in fr[s & 1], read one step ahead
            fr[0] = *reinterpret_cast<const f32x4*>(T);
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s + 1 < NKS) fr[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 1));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], fr[s & 1][t], acc[t], 0, 0, 0);
                if (LOAD && s < NPW) dma(c + 1, nxt, s);
            }
            HK_LB_WAIT_BARRIER();
            // C/D layout: row = 4 lq + r, column l15 of MFMA t = feature 4 l15 + t
            float* o = dy + f0 + (long long)c * CH + 4 * l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 16 * st + 4 * lq + r;
                if (b < B) *reinterpret_cast<f32x4*>(o + (long long)b * J) = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            }
        };
        int cur = 0;
        for (int c = 0; c + 1 < nch; ++c) {
            chunk(c, cur, std::true_type{});
            cur = cur ? 0 : STAGE;
        }
        chunk(nch - 1, cur, std::false_type{});
    } else {
        const int wv = wave - 4;                                 // class tiles 3 wv .. 3 wv + 2, and column quarter wv of tile 12
        // pieces q = wv + 4 u of the y tile: sample row 4 q + lq (clamped to B - 1)
        unsigned yo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int b = 4 * (wv + 4 * u) + lq;
            b = b < B ? b : B - 1;
            yo[u] = 4u * ((unsigned)b * (unsigned)J + 4u * l15);
        }
        const char* ybase = reinterpret_cast<const char*>(y + f0);
        auto dma = [&](int c, int sto, int u) {
            glds16(reinterpret_cast<const float*>(ybase + (long long)c * (CH * 4) + yo[u]), lds + sto + 256 * (WP + wv + 4 * u));
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) dma(0, 0, u);
        // resident A fragments: gt[i][s] = g[4 s + lq][16 (3 wv + i) + l15], g13[s] = g[4 s + lq][192 + l15]
        float gt[3][16], g13[16];
        int cls4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cls4[i] = i < 3 ? 16 * (3 * wv + i) + l15 : 192 + l15;
#pragma unroll
        for (int s = 0; s < 16; ++s) {                           // (all loads, then pins, then selects: see role dy)
            const float* gb = g + (long long)(4 * s + lq < B ? 4 * s + lq : B - 1) * K;
#pragma unroll
            for (int i = 0; i < 4; ++i) (i < 3 ? gt[i][s] : g13[s]) = gb[cls4[i] < K ? cls4[i] : K - 1];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) HK_PIN_LOADED((i < 3 ? gt[i][s] : g13[s]));
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float& r = i < 3 ? gt[i][s] : g13[s];
                r = (4 * s + lq < B && cls4[i] < K) ? r : 0.f;
            }
        if (db != nullptr && slab == 0) {
            // db[k] = sum_b g[b][k]: the 16 sample steps of this lane in order, then the four lq groups (fixed tree)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) sum += i < 3 ? gt[i][s] : g13[s];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (lq == 0 && cls4[i] < K && (i < 3 || wv == 0)) db[cls4[i]] = sum;
            }
        }
        HK_LB_WAIT_BARRIER();
        auto chunk = [&](int c, int cur, auto load_tag) {
            constexpr bool LOAD = decltype(load_tag)::value;
            const int nxt = cur ? 0 : STAGE;
            const float* T = Lf + cur + 256 * WP;
            const float* T1 = lds + cur + 256 * WP + 64 * lq + 4 * l15 + wv;     // y[4 s + lq][4 l15 + wv] at + 256 s
            f32x4 acc[3][4], acc13 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 fr[2];                                        // fragments of step s in fr[s & 1] / f1[s & 1], read one step ahead
            float f1[2];
            fr[0] = *reinterpret_cast<const f32x4*>(T);
            f1[0] = T1[0];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
                    fr[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 1));
                    f1[(s + 1) & 1] = T1[256 * (s + 1)];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[i][s], fr[s & 1][t], acc[i][t], 0, 0, 0);
                acc13 = __builtin_amdgcn_mfma_f32_16x16x4f32(g13[s], f1[s & 1], acc13, 0, 0, 0);
                if (LOAD && s < 4) dma(c + 1, nxt, s);
            }
            HK_LB_WAIT_BARRIER();
            float* o = dw + f0 + (long long)c * CH + 4 * l15;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cls = 16 * (3 * wv + i) + 4 * lq + r;
                    if (cls < K)
                        *reinterpret_cast<f32x4*>(o + (long long)cls * J) = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = 192 + 4 * lq + r;
                if (cls < K) o[(long long)cls * J + wv] = acc13[r];
            }
        };
        int cur = 0;
        for (int c = 0; c + 1 < nch; ++c) {
            chunk(c, cur, std::true_type{});
            cur = cur ? 0 : STAGE;
        }
        chunk(nch - 1, cur, std::false_type{});
    }
#undef HK_LB_WAIT_BARRIER
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the wide classifier for up to 16 samples and up to 1024 outputs (OSME part FCs: N = 10, 100 352 -> 1024):
// a pure stream - W (411 MB) is read once, dW (411 MB) written once, 3 GFLOP of MFMA work next to 170 us of HBM time.
// One sample tile means no operand is shared between waves, so nothing goes through LDS on the way in: wave w owns the
// classes 128 w .. 128 w + 127 and loads its W rows with global_load_dwordx4 STRAIGHT into the B-fragment layout (lane:
// class row 4 s + lq, features 4 l15 ..+3 - 256-byte runs per row, the same "MFMA t = columns {4 n + t}" trick as
// linear_bwd64_kernel), eight class steps (8 KB) per group, the next group requested before the current one is used.
//   dy partial of the wave's 128 classes: 32 steps x 4 MFMAs, A = g[l15][class] resident; the eight partials of a chunk
//     are added in wave order through LDS (two 32 KB images, one barrier per chunk) - deterministic;
//   dW of its 8 class tiles: 4 sample steps x 4 MFMAs each, A = g^T resident, B = the chunk's y rows (4 loads per wave
//     and chunk, L2 hits for seven of the eight waves), 16-byte stores from the accumulators.
template <bool DY, bool DW>
__global__ __launch_bounds__(512, 2) void linear_bwd16_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              const float* __restrict__ y, float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int J,
                                                              int K, int CPS, int S) {
    constexpr int CH = 64;
    HK_DYN_LDS16(lds);                                           // 2 x [8 waves][16][64] dy partials
    const int slab = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = slab * CPS;
    int nch = J / CH - c0;
    nch = nch < CPS ? nch : CPS;
    if (slab >= S || nch <= 0) return;
    const int cb = 128 * wave;                                   // first class of this wave
    const long long f0 = (long long)c0 * CH;

    // resident A fragments (zero beyond B samples / K classes; loads, pins, selects: see linear_bwd64_kernel)
    float ga[32];                                                // dy:  g[l15][cb + 4 s + lq]
    float gt[8][4];                                              // dW:  g[4 s + lq][cb + 16 i + l15]
    {
        const float* gb = g + (long long)(l15 < B ? l15 : B - 1) * K;
#pragma unroll
        for (int s = 0; s < 32; ++s) ga[s] = gb[cb + 4 * s + lq < K ? cb + 4 * s + lq : K - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                gt[i][s] = g[(long long)(4 * s + lq < B ? 4 * s + lq : B - 1) * K + (cb + 16 * i + l15 < K ? cb + 16 * i + l15 : K - 1)];
#pragma unroll
        for (int s = 0; s < 32; ++s) HK_PIN_LOADED(ga[s]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) HK_PIN_LOADED(gt[i][s]);
#pragma unroll
        for (int s = 0; s < 32; ++s) ga[s] = (l15 < B && cb + 4 * s + lq < K) ? ga[s] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) gt[i][s] = (4 * s + lq < B && cb + 16 * i + l15 < K) ? gt[i][s] : 0.f;
    }
    if (DW && db != nullptr && slab == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float sum = (gt[i][0] + gt[i][1]) + (gt[i][2] + gt[i][3]);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (lq == 0 && cb + 16 * i + l15 < K) db[cb + 16 * i + l15] = sum;
        }
    }
    if (cb >= K && !DY) return;                                  // (a wave without classes still takes part in the dy reduction)

    // this lane's rows: W rows cb + 32 q + 4 u + lq of group q, y rows 4 s + lq - 32-bit element offsets from a wave-uniform
    // base (the launcher checks 32 J and 16 J fit; a class beyond K - 1 reads row K - 1: its g is zero)
    const float* wl = w + f0 + 4 * l15;
    const float* yl = y + f0 + 4 * l15;
    int yrow[4], orow[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) yrow[s] = (4 * s + lq < B ? 4 * s + lq : B - 1) * J;
#pragma unroll
    for (int r = 0; r < 4; ++r) orow[r] = (4 * lq + r) * J + 4 * l15;   // row 4 lq + r of a 16-class output tile
    auto load_w = [&](f32x4 (&dst)[8], int c, int q) {           // group q (0..3) of chunk c
        const int cls0 = cb + 32 * q;
        const float* base = wl + (long long)cls0 * J + (long long)c * CH;
        int lqv = lq;
        HK_PIN_LOADED(lqv);      // opaque: the 32 row offsets are recomputed per group (2 VALU ops each) instead of being
                                 // hoisted out of the chunk loop as 32 live register pairs
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int rel = 4 * u + lqv;
            rel = cls0 + rel < K ? rel : K - 1 - cls0;
            dst[u] = *reinterpret_cast<const f32x4*>(base + rel * J);
        }
    };
    auto load_y = [&](f32x4 (&dst)[4], int c) {
        const float* base = yl + (long long)c * CH;
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const f32x4*>(base + yrow[s]);
    };

    f32x4 wf[2][8], yf[4];       // (y rows: one register set, re-requested behind the chunk's last use - a second set spills)
    if (DY) load_w(wf[0], 0, 0);
    if (DW) load_y(yf, 0);
    for (int c = 0; c < nch; c += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {                         // (unrolled by two chunks: yf[cc] / the LDS image are static)
            const int ch = c + cc;
            if (ch >= nch) break;
            f32x4 accy[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) accy[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // request the next group: q + 1 of this chunk, or group 0 of the next chunk
                if (DY) {
                    if (q < 3) load_w(wf[(q + 1) & 1], ch, q + 1);
                    else if (ch + 1 < nch) load_w(wf[0], ch + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);           // the requests stay AHEAD of this group's MFMAs (left alone, the
                                                                 // scheduler sinks each load to its first use: no prefetch at all)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            accy[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[8 * q + u], wf[q & 1][u][t], accy[t], 0, 0, 0);
                }
                if (DW) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 acc[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[2 * q + i][s], yf[s][t], acc[t], 0, 0, 0);
                        // (wave-uniform tile base + four 32-bit lane offsets: with per-row 64-bit addresses the compiler
                        //  hoists all 64 of them out of the chunk loop and spills)
                        const int cls0 = cb + 32 * q + 16 * i;
                        float* o = dw + f0 + (long long)ch * CH + (long long)cls0 * J;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cls0 + 4 * lq + r < K) *reinterpret_cast<f32x4*>(o + orow[r]) = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                        __builtin_amdgcn_sched_barrier(0);       // one output tile at a time (the stream is HBM-bound: registers, not ILP)
                    }
                }
            }
            if (DW && ch + 1 < nch) {
                load_y(yf, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DY) {
                // the wave's partial [16][64] -> LDS image cc, slot wave; then wave w adds rows 2 w, 2 w + 1 over the slots
                float* img = lds + cc * (8 * 1024) + wave * 1024;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(img + (4 * lq + r) * 64 + 4 * l15) = (f32x4){accy[0][r], accy[1][r], accy[2][r], accy[3][r]};
                __syncthreads();
                if (lane < 32) {
                    const int row = 2 * wave + (lane >> 4);
                    const float* src = lds + cc * (8 * 1024) + row * 64 + 4 * l15;
                    f32x4 sum = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int v = 1; v < 8; ++v) sum += *reinterpret_cast<const f32x4*>(src + v * 1024);
                    if (row < B) *reinterpret_cast<f32x4*>(dy + (long long)row * J + f0 + (long long)ch * CH + 4 * l15) = sum;
                }
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

#ifdef HK_LAB
extern "C" int hk_lab_set_linear_mode(int v) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lin_lab), &v, sizeof(int)) == hipSuccess ? HK_OK : HK_ERR_UNSUPPORTED;
}
#endif

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
        const int nks = (K + 3) / 4 == 50 ? 50 : 52;
        const size_t ldsb = (size_t)2 * (nks + 16) * 1024;
        if (nks == 50) {
            HK_ALLOW_BIG_LDS(&linear_bwd64_kernel<50>, ldsb);
            hipLaunchKernelGGL(linear_bwd64_kernel<50>, dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S);
        } else {
            HK_ALLOW_BIG_LDS(&linear_bwd64_kernel<52>, ldsb);
            hipLaunchKernelGGL(linear_bwd64_kernel<52>, dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S);
        }
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
        const size_t ldsb = (size_t)2 * 8 * 1024 * sizeof(float);
        if (dy && dw) hipLaunchKernelGGL((linear_bwd16_kernel<true, true>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S);
        else if (dy) hipLaunchKernelGGL((linear_bwd16_kernel<true, false>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S);
        else hipLaunchKernelGGL((linear_bwd16_kernel<false, true>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S);
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
