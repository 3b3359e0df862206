// Backward of the wide classifiers (replaces autograd's two GEMMs behind nn.Linear at
// model/methods/BCNN.py:42,54, CBCNN.py:26,34, MPNCOV.py:31, OSME.py:34,43):  dy = g W,  dW = g^T y,  db = sum_b g.
// linear_bwd64_kernel: up to 64 samples x up to 208 classes (BCNN / CBCNN / MPN heads); linear_bwd16_kernel: up to 16
// samples x up to 1024 outputs (OSME part FCs).  Launched by hk_linear_bwd (linear.hip).
// LABV (timing-only instances for tools/probe/linear_lab.hip - results are wrong, the product instantiates LABV = 0 only):
// bit 0 no MFMAs, bit 1 no LDS-DMA / global loads inside the chunk loop, bit 2 no stores, bit 3 no fragment reads.
#pragma once
#include <type_traits>

#include "hk_common.h"

namespace hk {

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the wide classifier, up to 64 samples and up to 208 classes (BCNN 262 144 -> 200): BOTH products in ONE
// launch, one 8-wave workgroup per CU walking 64-feature chunks.
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
//   * two LDS stages (2 x 66 KB).  The pieces of chunk c + 1 are issued behind the first MFMA steps of chunk c and have
//     the rest of the chunk (~6 us) to land; the barrier that ends chunk c waits for them and publishes the stage;
//   * nothing but MFMAs between two barriers that is not spread out: fragments are read two (dy) / one (dW) step ahead
//     of their use, and the results of chunk c leave DURING chunk c + 1, one store behind each MFMA step (16 stores in a
//     burst behind the barrier kept the wave off the matrix pipe for ~1.6 us per chunk: store issue, not bandwidth).
// The last class tile (classes 192 ..) is dealt to the four dW waves by output column quarter (wave i: columns {4 n + i},
// 4-byte stores: 4 % of dW).  db = sum_b g falls out of the dW role's resident fragments in the workgroup of chunk 0.
// A role whose result is not wanted (dy == nullptr: stage-1 training of the classifier alone) exits at once; finished
// waves do not take part in s_barrier.  Deterministic: no atomics, fixed summation order.
template <int NKS, int LABV = 0>
__global__ __launch_bounds__(512, 2) void linear_bwd64_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              const float* __restrict__ y, float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int J,
                                                              int K, int CPS, int S, int walk) {
    constexpr int CH = 64;                               // features per chunk
    constexpr int WP = NKS, YP = 16;                     // 1 KB pieces of the W tile [4 NKS][64] / the y tile [64][64]
    constexpr int STAGE = (WP + YP) * 256;               // floats
    constexpr int NPW = (WP + 3) / 4;                    // W pieces per dy wave (at most)
    // every store instruction of the dW role is issued whatever K is (no tile of this wave is empty): its stores can be
    // COUNTED in the wait that ends a chunk.  50 class steps <=> 197 <= K <= 200.
    constexpr bool COUNTED = NKS == 50;
    HK_DYN_LDS16(lds);
    const int slab = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    // the workgroup's chunks: cfirst + i cstep.  walk 1 (default): workgroup s takes chunks s, s + S, s + 2 S .. - at any
    // moment the S workgroups read one contiguous S x 256-byte run of every W / y row (DRAM pages shared between
    // neighbours).  walk 0: a contiguous slab of CPS chunks per workgroup.
    const int nchunk = J / CH;
    const int cfirst = walk ? slab : slab * CPS, cstep = walk ? S : 1;
    int nch = walk ? (nchunk - slab + S - 1) / S : nchunk - cfirst;
    nch = nch < CPS ? nch : CPS;
    if (slab >= S || nch <= 0) return;
    const bool role_dy = wave < 4;                               // wave-uniform
    if (role_dy ? dy == nullptr : dw == nullptr) return;
    const long long f0 = (long long)cfirst * CH;                 // first feature of the workgroup
    const long long fstep = (long long)cstep * CH;               // features from one of its chunks to the next
    const float* Lf = lds + 4 * lane;                            // this lane's 16 bytes of a piece
    const bool st_ok = !(LABV & 4) || B < 0;                     // (timing-only instances: the stores stay in the code, never run)
    using T_ = std::true_type;
    using F_ = std::false_type;

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
                glds16(reinterpret_cast<const float*>(wbase + (long long)c * (fstep * 4) + wo[u]), lds + sto + 256 * p);
        };
#pragma unroll
        for (int u = 0; u < NPW; ++u) dma(0, 0, u);
        // resident A fragments: ga[s] = g[16 st + l15][4 s + lq]  (zero beyond B samples / K classes)
        float ga[NKS];
        {
            const int b = 16 * st + l15;
            const float* gb = g + (long long)(b < B ? b : B - 1) * K;
            // every load issued before the first use (clamped addresses; HK_PIN_LOADED keeps the compiler from sinking a
            // load into the select that follows it - it did, with an s_waitcnt vmcnt(0) per element: 50 L2 round trips)
#pragma unroll
            for (int s = 0; s < NKS; ++s) ga[s] = gb[4 * s + lq < K ? 4 * s + lq : K - 1];
#pragma unroll
            for (int s = 0; s < NKS; ++s) HK_PIN_LOADED(ga[s]);
#pragma unroll
            for (int s = 0; s < NKS; ++s) ga[s] = (b < B && 4 * s + lq < K) ? ga[s] : 0.f;
        }
        // this lane's rows of the output tile: sample 16 st + 4 lq + r, features 4 l15 ..+3 (32-bit element offsets)
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r] = (16 * st + 4 * lq + r) * J + 4 * l15;
        f32x4 out[4];                                            // the finished rows of the previous chunk
        auto store_row = [&](int c, int r) {                     // row r of chunk c's tile
            if (16 * st + 4 * lq + r < B && st_ok) *reinterpret_cast<f32x4*>(dy + f0 + (long long)c * fstep + orow[r]) = out[r];
        };
        HK_VM_BARRIER(0);
        auto chunk = [&](int c, int cur, auto load_tag, auto prev_tag) {
            constexpr bool LOAD = decltype(load_tag)::value, PREV = decltype(prev_tag)::value;
            const int nxt = cur ? 0 : STAGE;
            const float* T = Lf + cur;
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 fr[3];                                        // fragment of step s in fr[s % 3], read two steps ahead
            fr[0] = fr[1] = fr[2] = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (!(LABV & 8)) {
                fr[0] = *reinterpret_cast<const f32x4*>(T);
                fr[1] = *reinterpret_cast<const f32x4*>(T + 256);
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                if (s + 2 < NKS && !(LABV & 8)) fr[(s + 2) % 3] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 2));
                if (LOAD && s < NPW && !(LABV & 2)) dma(c + 1, nxt, s);
                if (PREV && s >= NPW && s < NPW + 4) store_row(c - 1, s - NPW);
                if (!(LABV & 1)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], fr[s % 3][t], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);               // a step's requests stay in ITS step
            }
            // C/D layout: row = 4 lq + r, column l15 of MFMA t = feature 4 l15 + t
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            HK_VM_BARRIER(0);      // (the previous chunk's stores went out behind steps 13 .. 16 of 50: long complete)
        };
        int cur = 0;
        if (nch == 1) {
            chunk(0, 0, F_{}, F_{});
        } else {
            chunk(0, 0, T_{}, F_{});
            cur = STAGE;
            for (int c = 1; c + 1 < nch; ++c) {
                chunk(c, cur, T_{}, T_{});
                cur = cur ? 0 : STAGE;
            }
            chunk(nch - 1, cur, F_{}, T_{});
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) store_row(nch - 1, r);
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
            glds16(reinterpret_cast<const float*>(ybase + (long long)c * (fstep * 4) + yo[u]), lds + sto + 256 * (WP + wv + 4 * u));
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
        if (db != nullptr && cfirst == 0) {
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
        // this lane's rows of a 16-class output tile: class 4 lq + r of the tile, features 4 l15 ..+3
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r] = (4 * lq + r) * J + 4 * l15;
        f32x4 out[3][4], out13;                                  // the finished rows of the previous chunk
        auto store_row = [&](int c, int e) {                     // store e (0 .. 11: tile e / 4, row e % 4; 12 .. 15: tile 12, row e - 12)
            float* o = dw + f0 + (long long)c * fstep;
            if (e < 12) {
                const int cls0 = 16 * (3 * wv + e / 4);
                if (cls0 + 4 * lq + e % 4 < K && st_ok) *reinterpret_cast<f32x4*>(o + (long long)cls0 * J + orow[e % 4]) = out[e / 4][e % 4];
            } else {
                if (192 + 4 * lq + (e - 12) < K && st_ok) o[(long long)192 * J + orow[e - 12] + wv] = out13[e - 12];
            }
        };
        HK_VM_BARRIER(0);
        auto chunk = [&](int c, int cur, auto load_tag, auto prev_tag) {
            constexpr bool LOAD = decltype(load_tag)::value, PREV = decltype(prev_tag)::value;
            const int nxt = cur ? 0 : STAGE;
            const float* T = Lf + cur + 256 * WP;
            f32x4 acc[3][4], acc13 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // tile 12: this wave's output columns {4 n + wv} = y[4 s + lq][4 l15 + wv], its own 4-byte read (a select on the
            // wave-uniform wv turns into a jump table in the middle of the MFMA stream; the read's 4-way bank conflict
            // is 16 reads per chunk)
            const float* T1 = lds + cur + 256 * WP + 4 * lane + wv;
            f32x4 fr[2];                                        // fragments of step s in fr[s & 1] / f1[s & 1], read one step ahead
            float f1[2];
            fr[0] = fr[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
            f1[0] = f1[1] = 1.f;
            if (!(LABV & 8)) {
                fr[0] = *reinterpret_cast<const f32x4*>(T);
                f1[0] = T1[0];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16 && !(LABV & 8)) {
                    fr[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 1));
                    f1[(s + 1) & 1] = T1[256 * (s + 1)];
                }
                // the next chunk's pieces first (steps 0 .. 3), then the previous chunk's 16 stores (steps 4 .. 15: 1-2 a step)
                if (LOAD && s < 4 && !(LABV & 2)) dma(c + 1, nxt, s);
                if (PREV && s >= 4) {
                    store_row(c - 1, s - 4);
                    if (s >= 12) store_row(c - 1, s);
                }
                if (!(LABV & 1)) {
                    const f32x4 f = fr[s & 1];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[i][s], f[t], acc[i][t], 0, 0, 0);
                    acc13 = __builtin_amdgcn_mfma_f32_16x16x4f32(g13[s], f1[s & 1], acc13, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[i][r] = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            out13 = acc13;
            // the pieces of chunk c + 1 were issued BEFORE this chunk's 16 stores: with every store certain to issue
            // (COUNTED) the wait leaves exactly those 16 in flight (vmcnt retires in issue order); otherwise it drains
            if (COUNTED && PREV) HK_VM_BARRIER(16); else HK_VM_BARRIER(0);
        };
        int cur = 0;
        if (nch == 1) {
            chunk(0, 0, F_{}, F_{});
        } else {
            chunk(0, 0, T_{}, F_{});
            cur = STAGE;
            for (int c = 1; c + 1 < nch; ++c) {
                chunk(c, cur, T_{}, T_{});
                cur = cur ? 0 : STAGE;
            }
            chunk(nch - 1, cur, F_{}, T_{});
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) store_row(nch - 1, e);
    }
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
template <bool DY, bool DW, int LABV = 0>
__global__ __launch_bounds__(512, 2) void linear_bwd16_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              const float* __restrict__ y, float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int J,
                                                              int K, int CPS, int S, int walk) {
    constexpr int CH = 64;
    HK_DYN_LDS16(lds);                                           // 2 x [8 waves][16][64] dy partials
    const int slab = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int nchunk = J / CH;                                   // (the chunk walk: see linear_bwd64_kernel)
    const int cfirst = walk ? slab : slab * CPS, cstep = walk ? S : 1;
    int nch = walk ? (nchunk - slab + S - 1) / S : nchunk - cfirst;
    nch = nch < CPS ? nch : CPS;
    if (slab >= S || nch <= 0) return;
    const int cb = 128 * wave;                                   // first class of this wave
    const long long f0 = (long long)cfirst * CH;
    const long long fstep = (long long)cstep * CH;

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
        const float* base = wl + (long long)cls0 * J + (long long)c * fstep;
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
        const float* base = yl + (long long)c * fstep;
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
                        float* o = dw + f0 + (long long)ch * fstep + (long long)cls0 * J;
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
                    if (row < B) *reinterpret_cast<f32x4*>(dy + (long long)row * J + f0 + (long long)ch * fstep + 4 * l15) = sum;
                }
            }
        }
    }
}

}  // namespace hk
