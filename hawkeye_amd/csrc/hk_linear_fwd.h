// Forward of the wide classifiers (replaces nn.Linear.forward at model/methods/BCNN.py:42,54, CBCNN.py:26,34,
// OSME.py:34,43): linear_skinny_kernel, launched by hk_linear_fwd (linear.hip).
// LABV (timing-only instances for tools/probe/linear_lab.hip - results are wrong, the product instantiates LABV = 0 only):
// bit 0 no MFMAs, bit 1 no LDS-DMA inside the chunk loop, bit 3 no fragment reads, bit 5 nt policy on the LDS-DMA loads.
#pragma once
#include <type_traits>

#include "hk_common.h"

namespace hk {

// ---------------------------------------------------------------------------------------------------------------------
// Forward for the WIDE classifiers (BCNN 262 144 -> 200, OSME 100 352 -> 1024): a workgroup owns one slab of features and
// ALL of its (up to 64) samples x a group of NT 16-column class tiles, so y is read once and W once, through LDS-DMA.
// The generic split-K path above runs at 133-144 us on the BCNN shape = 1.95 TB/s (profiles/r2_pool_kernels_pmc.csv:
// 53 % of its wave time parked on loads, y fetched by four class tiles, one chunk of register prefetch); the product is
// balanced between the matrix pipe and HBM (24 FLOP/B), so both have to be kept busy:
//   * 512 threads = 8 waves: wave w owns samples 16 (w & 3) .. + 15 and the class tiles of half w >> 2 (7 + 6 of 13, or
//     8 + 8 of 16), 16x16x4 MFMA, A operand = y rows, B operand = W rows - both tiles are [row][32 features] exactly as
//     they lie in memory, read back with ds_read_b128 through the XOR swizzle of the Gram backward (slot row * 8 + (k4 ^ (row & 7)));
//   * chunks of 32 features, FOUR LDS stages (4 x 34.8 KB for 13 class tiles, 4 x 38.9 KB for 15), the pieces of chunk
//     c + 3 are issued during chunk c; the barrier that ends a chunk waits with s_waitcnt vmcnt(n) for everything but
//     the n pieces the wave has just issued, so a piece has two whole chunks to arrive;
//   * the fragments of chunk c + 1 (complete one barrier earlier) are read behind the last MFMAs of chunk c.
// Partial results [S][B][K] as before, added in slab order by linear_reduce_kernel: deterministic.

// MT: 16-sample row tiles per workgroup.  4: up to 64 samples, wave w owns row tile w & 3 and one half of the NT class tiles.
// 1: up to 16 samples (OSME: N = 10) - every wave owns the same 16 rows and NT / 8 of the class tiles; the product is then
// a pure stream of W (2 KB of LDS-DMA pieces per MFMA-cycle-pair), the matrix pipe idles.
template <int NT, int MT, int LABV = 0>
__global__ __launch_bounds__(512, 2) void linear_skinny_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                               float* __restrict__ part, int B, int J, int K, int KS,
                                                               int S, int ngrp, int walk) {
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
    // the slab's chunks.  walk 0: the KS consecutive features slab KS ..; walk 1: the chunks slab, slab + S, slab + 2 S ..
    // of the whole feature axis - at any moment the S workgroups read one contiguous S x 128-byte run of every row (DRAM
    // pages shared between neighbours).  Either way partial[slab] is a fixed set of chunks added in a fixed order.
    const int nchunk = J / CH;
    const long long f0 = walk ? (long long)slab * CH : (long long)slab * KS;        // first feature of the slab
    const long long fstep = walk ? (long long)S * CH : CH;                            // features from one chunk to the next
    int nch;
    if (walk) nch = slab < nchunk ? (nchunk - slab + S - 1) / S : 0;
    else {
        const int nfeat = (J - f0) < KS ? (int)(J - f0) : KS;   // (KS % 32 == 0; J % 32 != 0: whole chunks only, the tail is
                                                                //  linear_reduce_kernel's)
        nch = nfeat > 0 ? nfeat / CH : 0;
    }

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
        const long long fo = f0 + (long long)c * fstep;
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            if ((u < (PPW + 1) / 2) != (part == 0)) continue;
            const int p = wave + 8 * u;
            if (p < NP) glds16<(LABV & 32) ? 2 : 0>((p < NPA ? y : w) + src[u] + fo, lds + st + 256 * p);
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
    // with the reads before the group, 88.1 us with the reads behind its first seven MFMAs.  Round 4 (tools/linear_lab.py,
    // profiles/r4_linear_lab.json): the LDS-DMA stream alone takes 44.5 us (6.2 TB/s with the interleaved chunk walk, 50.3
    // with contiguous slabs), MFMAs + fragment reads + barriers alone 55-58 us, the whole kernel 68-71 us.  Reading a chunk's
    // fragments only AFTER the barrier that publishes it (so that the barrier waits for one chunk less and a piece has two
    // more chunk times to land) measured 78.6 us: memory latency is not what the overlap loses; issuing all pieces from the
    // four waves with one class tile less (6 of 13) instead of from all eight: 71.6 vs 70.2 us - nor is it who issues them.
    // Halving the fragment reads - a wave takes TWO sample tiles x its class half on ONE of the chunk's two feature steps
    // (9 reads per 56 MFMAs instead of 16; the steps meet through LDS after the loop) - measured 96 vs 84 us in one
    // alternating run (7 + 7 / 6 + 6 class tiles on the two waves of a SIMD instead of 7 + 6 explains half of it): removed.
    auto run = [&](auto nl_tag) {
        constexpr int NL = decltype(nl_tag)::value;
        auto frag = [&](int st, int s, f32x4& a, f32x4 (&b)[NL]) {
            if (LABV & 8) return;
            const float* base = lds + st;
            a = *reinterpret_cast<const f32x4*>(base + aoff + (((4 * s + lq) ^ asw) << 2));
#pragma unroll
            for (int n = 0; n < NL; ++n)
                b[n] = *reinterpret_cast<const f32x4*>(base + boff + n * 16 * CH + (((4 * s + lq) ^ bsw) << 2));
        };
        auto mma = [&](const f32x4& a, const f32x4 (&b)[NL]) {
            if (LABV & 1) return;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n][t], acc[n], 0, 0, 0);
        };
        f32x4 a0, a1, b0[NL], b1[NL];
        if (LABV & 8) {                                          // (timing-only: fragments that no two MFMAs share)
            a0 = (f32x4){1.f + lane, 2.f, 3.f, 4.f};
            a1 = a0 * 0.5f;
#pragma unroll
            for (int n = 0; n < NL; ++n) { b0[n] = a0 * (float)(n + 2); b1[n] = a1 * (float)(n + 3); }
        }
        int cur = 0;                                             // stage of chunk c (float offset), nxt = chunk c + 1
        frag(0, 0, a0, b0);
        for (int c = 0; c < nch; ++c) {
            const int nxt = cur + STAGE < NS * STAGE ? cur + STAGE : 0;
            const int dst = cur >= STAGE ? cur - STAGE : (NS - 1) * STAGE;  // stage of chunk c - 1 = chunk c + NS - 1
            const bool load = c + NS - 1 < nch && !(LABV & 2);   // uniform
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

}  // namespace hk
