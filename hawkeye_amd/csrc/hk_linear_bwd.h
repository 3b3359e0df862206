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
// wave of each role per SIMD: the W / dy stream and the y / dW stream run side by side (553 MB of combined traffic; the
// stream alone takes 108 us = 5.2 TB/s for this half-read half-write mix, the 13.4 GFLOP alone 109 us - the two sides
// are of equal length, so everything here is about keeping BOTH busy all the time):
//   * g never moves: it is staged once through LDS (coalesced, zero-padded to [64][209]) and role dy keeps
//     A = g[16 st + l15][4 s + lq] of its sample tile st for all NKS class steps in registers, role dW keeps A = g^T of its
//     three (+ a quarter of the thirteenth) class tiles for all 16 sample steps;
//   * the B operand is the streamed tile exactly as it lies in memory.  A tile row is 64 consecutive features; LDS-DMA
//     piece p holds rows 4 p .. 4 p + 3 (lane l: row 4 p + (l >> 4), features 4 (l & 15) ..+3), so the fragment of class /
//     sample step s is ONE linear ds_read_b128 at 1 KB s + 16 lane - no swizzle, no conflicts - whose four floats feed the
//     four MFMAs of the step: MFMA t computes the output columns {4 n + t}.  A lane's four accumulators therefore hold
//     four CONSECUTIVE features of a row and leave as 16-byte stores (256-byte runs per row) straight from registers;
//   * the pipeline unit is HALF a chunk - classes 0 .. 2 NKS - 1 of W and samples 0 .. 31 of y, then the other halves,
//     accumulated in the same registers - on FOUR LDS stages of 33 KB: the pieces of unit u + 3 are issued behind the
//     first MFMA steps of unit u, so ~100 KB per CU are always on their way (with whole chunks on two stages the next
//     chunk was requested in a burst and memory idled for the rest of the chunk: 143 us);
//   * results leave with the nt (streamed, do not keep in L2) policy: 138.7 -> 132.1 us (nt on the LDS-DMA loads as well
//     gave nothing more: 134);
//   * nothing but MFMAs between two barriers that is not spread out: fragments are read two (dy) / one (dW) step ahead
//     of their use, and the results of chunk c leave DURING chunk c + 1, a store behind an MFMA step (16 stores in a burst
//     behind the barrier kept a wave off the matrix pipe for ~1.6 us per chunk: store issue, not bandwidth);
//   * the barrier that ends unit u waits for the pieces of unit u + 1 only: s_waitcnt vmcnt(n) with n = everything this
//     wave has issued since (two units of pieces, three units of stores; vmcnt retires in issue order).  The count is
//     exact because every store ALWAYS issues: ragged sample / class edges are cut by the buffer descriptor's bounds
//     check (buf_store16), not by a branch around the instruction.
// The last class tile (classes 192 ..) is dealt to the four DY waves by output column quarter (wave i: columns 16 i .. 16 i + 15,
// one more MFMA in eight of its 25 steps, 4-byte stores: 4 % of dW) - the dW waves hold 48 resident fragments, 48
// accumulators and the 48 finished values of the previous chunk, the dy waves have room.  db = sum_b g falls out of the
// resident fragments in the workgroup of chunk 0.  Deterministic: no atomics, fixed summation order.
// MODE 0: both products; 1: dy only (the dW waves leave, class tile 12 is skipped); 2: dW only (the dy waves compute
// nothing but their quarter of class tile 12, no W traffic).
template <int NKS, int MODE = 0, int LABV = 0>
__global__ __launch_bounds__(512, 2) void linear_bwd64_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              const float* __restrict__ y, float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int J,
                                                              int K, int CPS, int S, int walk,
                                                              const float* __restrict__ row_scale = nullptr) {
    // row_scale (optional, [B]): dW = (row_scale g)^T y - y is a pooled vector handed over unnormalised and row_scale its
    // 1 / |z| (hk_linear_bwd_scaled); dy and db are those of the unscaled g
    static_assert(NKS % 2 == 0 && NKS <= 52, "two halves of class steps; up to 208 classes");
    constexpr int CH = 64;                               // features per chunk
    constexpr int NKH = NKS / 2;                         // class steps = 1 KB W pieces per half chunk
    constexpr int YPH = 8;                               // y pieces per half chunk (32 samples)
    constexpr int STAGE = (NKH + YPH) * 256;             // floats of one LDS stage (half a chunk)
    constexpr int NPW = (NKH + 3) / 4;                   // W pieces per dy wave and unit (at most)
    constexpr int GP = 209;                              // row pitch of the g image
    static_assert(64 * GP <= 2 * STAGE, "the g image fits stages 2 and 3");
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
    const int U = 2 * nch;                                       // pipeline units
    const bool role_dy = wave < 4;                               // wave-uniform
    constexpr bool DO_DY = MODE != 2, DO_DW = MODE != 1;
    const long long f0 = (long long)cfirst * CH;                 // first feature of the workgroup
    const long long fstep = (long long)cstep * CH;               // features from one of its chunks to the next
    const float* Lf = lds + 4 * lane;                            // this lane's 16 bytes of a piece
    const bool st_ok = !(LABV & 4) || B < 0;                     // (timing-only instances: the stores stay in the code, never run)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- g -> LDS image [64][GP] in stages 2 and 3 (zero beyond B samples / K classes), coalesced
    {
        constexpr int NE = (64 * GP + 511) / 512;
        float tmp[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 512 * i, b = e / GP, c = e - b * GP;
            tmp[i] = g[(long long)(b < B ? b : B - 1) * K + (c < K ? c : K - 1)];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) HK_PIN_LOADED(tmp[i]);      // (every load before the first select: see HK_PIN_LOADED)
        float* gi = lds + 2 * STAGE;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 512 * i, b = e / GP, c = e - b * GP;
            if (e < 64 * GP) gi[e] = (b < B && c < K) ? tmp[i] : 0.f;
        }
    }
    const float* gi = lds + 2 * STAGE;

    if (role_dy) {
        const int st = wave;                                     // sample tile; also the column quarter (16 st ..) of class tile 12
        // pieces p = wave + 4 i of a W half tile: class row 4 (h NKH + p) + lq (clamped to K - 1), features 4 l15 ..+3
        // (byte offsets from the chunk's base: K J 4 < 4 GB is checked by the launcher)
        unsigned wo[2][NPW];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                int c = 4 * (h * NKH + wave + 4 * i) + lq;
                c = c < K ? c : K - 1;
                wo[h][i] = 4u * ((unsigned)c * (unsigned)J + 4u * l15);
            }
        const int npc = (NKH - wave + 3) / 4;                    // pieces of this wave per unit: NPW or NPW - 1
        const char* wbase = reinterpret_cast<const char*>(w + f0);
        auto dma = [&](int v, int h, int i) __attribute__((always_inline)) {   // piece i of unit v (= chunk v >> 1, half h == v & 1)
            const int p = wave + 4 * i;                          // (4 i + 3 < NKH folds at compile time: only the last i branches)
            if (DO_DY && (4 * i + 3 < NKH || p < NKH))
                glds16<(LABV & 32) ? 2 : 0>(reinterpret_cast<const float*>(wbase + (long long)(v >> 1) * (fstep * 4) + wo[h][i]),
                       lds + (v & 3) * STAGE + 256 * p);
        };
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma(0, 0, i);
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma(1, 1, i);
        HK_LDS_BARRIER();                                        // the g image is written
        // resident A fragments: ga[s] = g[16 st + l15][4 s + lq];  class tile 12: g13[s] = g[4 s + lq][192 + l15]
        float ga[DO_DY ? NKS : 1], g13[DO_DW ? 16 : 1];
        if (DO_DY) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) ga[s] = gi[(16 * st + l15) * GP + 4 * s + lq];
        }
        if (DO_DW) {
#pragma unroll
            for (int s = 0; s < 16; ++s) g13[s] = gi[(4 * s + lq) * GP + 192 + l15];
            if (db != nullptr && cfirst == 0 && wave == 0) {     // db of the classes 192 ..: see the dW role
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) sum += g13[s];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (lq == 0 && 192 + l15 < K) db[192 + l15] = sum;
            }
            if (row_scale != nullptr) {
#pragma unroll
                for (int s = 0; s < 16; ++s) g13[s] *= row_scale[4 * s + lq < B ? 4 * s + lq : B - 1];
            }
        }
        HK_LDS_BARRIER();                                        // ... and read by everybody: stages 2, 3 are free
        if (U > 2) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) dma(2, 0, i);
        }
        // this lane's rows of the dy tile: sample 16 st + 4 lq + r, features 4 l15 ..+3; of class tile 12: class 192 + 4 lq + r,
        // feature 16 st + l15 (byte offsets; rows beyond B / K lie beyond the descriptor's end: dropped by the hardware)
        const buf_rsrc_t rs = buf_rsrc(dy, DO_DY ? (long long)B * J : 0);
        const buf_rsrc_t rs12 = buf_rsrc(dw, DO_DW ? (long long)K * J : 0);
        unsigned orow[4], orow12[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            orow[r] = 4u * ((unsigned)(16 * st + 4 * lq + r) * (unsigned)J + 4u * l15);
            orow12[r] = 4u * ((unsigned)(192 + 4 * lq + r) * (unsigned)J + 16u * st + l15);
        }
        f32x4 out[4], out13;                                     // the finished rows of the previous chunk
        auto store_row = [&](int c, int r) __attribute__((always_inline)) {       // row r of chunk c's dy tile
            if (DO_DY && st_ok) buf_store16<(LABV & 16) ? 0 : 2>(rs, orow[r] + 4u * (unsigned)(f0 + (long long)c * fstep), out[r]);
        };
        auto store_12 = [&](int c, int r) __attribute__((always_inline)) {        // row r of chunk c's quarter of class tile 12
            if (DO_DW && st_ok) buf_store4(rs12, orow12[r] + 4u * (unsigned)(f0 + (long long)c * fstep), out13[r]);
        };
        constexpr int PW = DO_DY ? 2 * NPW : 0;                  // pieces issued in two units (by a wave with NPW of them)
        constexpr int SW = 3 * ((DO_DY ? 2 : 0) + (DO_DW ? 2 : 0));   // stores issued in three units
        // unit 0 has landed when all but the pieces of units 1 and 2 have
        if (U > 2 && !(LABV & 15)) { if (npc == NPW) HK_VM_BARRIER(PW); else HK_VM_BARRIER(PW >= 2 ? PW - 2 : 0); }
        else HK_VM_BARRIER(0);
        f32x4 acc[4], acc13;
        // H: half (u & 1); LOAD: unit u + 3 exists; PREV: chunk (u >> 1) - 1 has rows to store; WK: how the end-of-unit wait counts
        auto unit = [&](int u, auto h_tag, auto load_tag, auto prev_tag, auto wk_tag) __attribute__((always_inline)) {
            constexpr int H = decltype(h_tag)::value, WK = decltype(wk_tag)::value;
            constexpr bool LOAD = decltype(load_tag)::value != 0, PREV = decltype(prev_tag)::value != 0;
            const float* T = Lf + (u & 3) * STAGE;
            // class tile 12: y[32 H + 4 s + lq][16 st + l15] at + 256 s (row 4 s + lq of piece s is 64 lq floats into it).
            // Round 5: the wave takes the 16 CONSECUTIVE features 16 st .. of the tile, not every fourth one (4 l15 + st):
            // the 32 lanes of a ds_read_b32 group then fall on 16 banks twice (2-way) instead of 8 banks four times -
            // these reads were the kernel's 25.8 % LDS conflict cycles - and its stores are 64-byte runs of a dW row
            // instead of every fourth word of a 256-byte run shared with the three other waves.  Same sums, same bits.
            // (the lane part is re-derived here from an opaque copy of the lane id - three VALU ops per unit: kept live across
            //  the whole role it is one register too many for the 256 this kernel has, and the spill code lands between units)
            int ln_ = lane;
            HK_PIN_LOADED(ln_);
            const float* T1 = lds + (u & 3) * STAGE + 256 * NKH + 16 * st + 64 * (ln_ >> 4) + (ln_ & 15);
            if (H == 0) {
                acc13 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            f32x4 fr[3];                                        // fragment of step s in fr[s % 3], read two steps ahead
            float f1[2];
            fr[0] = fr[1] = fr[2] = (f32x4){1.f, 1.f, 1.f, 1.f};
            f1[0] = f1[1] = 1.f;
            if (!(LABV & 8)) {
                if (DO_DY) {
                    fr[0] = *reinterpret_cast<const f32x4*>(T);
                    fr[1] = *reinterpret_cast<const f32x4*>(T + 256);
                }
                if (DO_DW) f1[0] = T1[0];
            }
#pragma unroll
            for (int s = 0; s < NKH; ++s) {
                if (!(LABV & 8)) {
                    if (DO_DY && s + 2 < NKH) fr[(s + 2) % 3] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 2));
                    if (DO_DW && s + 1 < 8) f1[(s + 1) & 1] = T1[256 * (s + 1)];
                }
                if (LOAD && s < NPW && !(LABV & 2)) dma(u + 3, H ^ 1, s);
                if (PREV && (s == 8 || s == 16)) store_row((u >> 1) - 1, 2 * H + (s == 16));
                if (PREV && (s == 12 || s == 20)) store_12((u >> 1) - 1, 2 * H + (s == 20));
                if (!(LABV & 1)) {
                    if (DO_DY) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[DO_DY ? H * NKH + s : 0], fr[s % 3][t], acc[t], 0, 0, 0);
                    }
                    if (DO_DW && s < 8)
                        acc13 = __builtin_amdgcn_mfma_f32_16x16x4f32(g13[DO_DW ? 8 * H + s : 0], f1[s & 1], acc13, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);               // a step's requests stay in ITS step
            }
            if (H == 1) {                                        // C/D layout: row = 4 lq + r, column l15 of MFMA t = feature 4 l15 + t
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                out13 = acc13;
            }
            // younger than the pieces of unit u + 1: the pieces of units u + 2, u + 3 and (WK 2) the stores of the units
            // u - 2, u - 1, u (all of them behind their unit's last piece)
            if (WK == 0 || (LABV & 15)) HK_VM_BARRIER(0);
            else if (WK == 1) { if (npc == NPW) HK_VM_BARRIER(PW); else HK_VM_BARRIER(PW >= 2 ? PW - 2 : 0); }
            else { if (npc == NPW) HK_VM_BARRIER(PW + SW); else HK_VM_BARRIER(PW >= 2 ? PW - 2 + SW : SW); }
        };
        auto run = [&](int u) __attribute__((always_inline)) {                                  // any unit, by its place in the pipeline
            const bool load = u + 3 < U, prev = u >= 2;
            if (u & 1) {
                if (load && u >= 4) unit(u, I1{}, I1{}, I1{}, I2{});
                else if (load && prev) unit(u, I1{}, I1{}, I1{}, I1{});
                else if (load) unit(u, I1{}, I1{}, I0{}, I1{});
                else if (prev) unit(u, I1{}, I0{}, I1{}, I0{});
                else unit(u, I1{}, I0{}, I0{}, I0{});
            } else {
                if (load && u >= 4) unit(u, I0{}, I1{}, I1{}, I2{});
                else if (load && prev) unit(u, I0{}, I1{}, I1{}, I1{});
                else if (load) unit(u, I0{}, I1{}, I0{}, I1{});
                else if (prev) unit(u, I0{}, I0{}, I1{}, I0{});
                else unit(u, I0{}, I0{}, I0{}, I0{});
            }
        };
        int u = 0;
        for (; u < 4 && u < U; ++u) run(u);
        for (; u + 4 < U; u += 2) {                              // steady state (u is even)
            unit(u, I0{}, I1{}, I1{}, I2{});
            unit(u + 1, I1{}, I1{}, I1{}, I2{});
        }
        for (; u < U; ++u) run(u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            store_row(nch - 1, r);
            store_12(nch - 1, r);
        }
    } else {
        const int wv = wave - 4;                                 // class tiles 3 wv .. 3 wv + 2
        // pieces q = wv + 4 i of a y half tile: sample row 32 h + 4 q + lq (clamped to B - 1)
        unsigned yo[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int b = 32 * h + 4 * (wv + 4 * i) + lq;
                b = b < B ? b : B - 1;
                yo[h][i] = 4u * ((unsigned)b * (unsigned)J + 4u * l15);
            }
        const char* ybase = reinterpret_cast<const char*>(y + f0);
        auto dma = [&](int v, int h, int i) __attribute__((always_inline)) {
            glds16<(LABV & 32) ? 2 : 0>(reinterpret_cast<const float*>(ybase + (long long)(v >> 1) * (fstep * 4) + yo[h][i]),
                   lds + (v & 3) * STAGE + 256 * (NKH + wv + 4 * i));
        };
        if (DO_DW) {
            dma(0, 0, 0); dma(0, 0, 1);
            dma(1, 1, 0); dma(1, 1, 1);
        }
        HK_LDS_BARRIER();                                        // the g image is written
        // resident A fragments: gt[i][s] = g[4 s + lq][16 (3 wv + i) + l15]
        float gt[3][16];
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 3; ++i) gt[i][s] = gi[(4 * s + lq) * GP + 16 * (3 * wv + i) + l15];
        if (DO_DW && db != nullptr && cfirst == 0) {
            // db[k] = sum_b g[b][k]: the 16 sample steps of this lane in order, then the four lq groups (fixed tree)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) sum += gt[i][s];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const int cls = 16 * (3 * wv + i) + l15;
                if (lq == 0 && cls < K) db[cls] = sum;
            }
        }
        if (row_scale != nullptr) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float r = row_scale[4 * s + lq < B ? 4 * s + lq : B - 1];
#pragma unroll
                for (int i = 0; i < 3; ++i) gt[i][s] *= r;
            }
        }
        HK_LDS_BARRIER();                                        // ... and read by everybody: stages 2, 3 are free
        if (!DO_DW) return;                                      // (finished waves do not take part in s_barrier)
        if (U > 2) { dma(2, 0, 0); dma(2, 0, 1); }
        // this lane's rows of a 16-class output tile: class 4 lq + r of the tile, features 4 l15 ..+3 (byte offsets; classes
        // beyond K lie beyond the descriptor's end)
        const buf_rsrc_t rs = buf_rsrc(dw, (long long)K * J);
        unsigned orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r] = 4u * ((unsigned)(4 * lq + r) * (unsigned)J + 4u * l15);
        f32x4 out[3][4];                                         // the finished rows of the previous chunk
        auto store_row = [&](int c, int e) __attribute__((always_inline)) {       // store e: tile e / 4, row e % 4
            if (!st_ok) return;
            const unsigned fo = 4u * (unsigned)(f0 + (long long)c * fstep);
            buf_store16<(LABV & 16) ? 0 : 2>(rs, orow[e % 4] + fo + 4u * (unsigned)(16 * (3 * wv + e / 4)) * (unsigned)J, out[e / 4][e % 4]);
        };
        if (U > 2 && !(LABV & 15)) HK_VM_BARRIER(4); else HK_VM_BARRIER(0);
        f32x4 acc[3][4];
        auto unit = [&](int u, auto h_tag, auto load_tag, auto prev_tag, auto wk_tag) __attribute__((always_inline)) {
            constexpr int H = decltype(h_tag)::value, WK = decltype(wk_tag)::value;
            constexpr bool LOAD = decltype(load_tag)::value != 0, PREV = decltype(prev_tag)::value != 0;
            const float* T = Lf + (u & 3) * STAGE + 256 * NKH;
            if (H == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            f32x4 fr[2];                                        // fragment of step s in fr[s & 1], read one step ahead
            fr[0] = fr[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (!(LABV & 8)) fr[0] = *reinterpret_cast<const f32x4*>(T);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8 && !(LABV & 8)) fr[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(T + 256 * (s + 1));
                // both pieces of unit u + 3 first, then a store of the previous chunk in each of the next six steps (every
                // store is younger than the unit's last piece: the count in the wait below)
                if (LOAD && s == 0 && !(LABV & 2)) { dma(u + 3, H ^ 1, 0); dma(u + 3, H ^ 1, 1); }
                if (PREV && s < 6) store_row((u >> 1) - 1, 6 * H + s);
                if (!(LABV & 1)) {
                    const f32x4 f = fr[s & 1];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[i][8 * H + s], f[t], acc[i][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (H == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[i][r] = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            }
            // younger than the pieces of unit u + 1: two pieces in each of the units u + 2, u + 3 and (WK 2) six stores in
            // each of the units u - 2, u - 1, u
            if (WK == 0 || (LABV & 15)) HK_VM_BARRIER(0);
            else if (WK == 1) HK_VM_BARRIER(4);
            else HK_VM_BARRIER(22);
        };
        auto run = [&](int u) __attribute__((always_inline)) {
            const bool load = u + 3 < U, prev = u >= 2;
            if (u & 1) {
                if (load && u >= 4) unit(u, I1{}, I1{}, I1{}, I2{});
                else if (load && prev) unit(u, I1{}, I1{}, I1{}, I1{});
                else if (load) unit(u, I1{}, I1{}, I0{}, I1{});
                else if (prev) unit(u, I1{}, I0{}, I1{}, I0{});
                else unit(u, I1{}, I0{}, I0{}, I0{});
            } else {
                if (load && u >= 4) unit(u, I0{}, I1{}, I1{}, I2{});
                else if (load && prev) unit(u, I0{}, I1{}, I1{}, I1{});
                else if (load) unit(u, I0{}, I1{}, I0{}, I1{});
                else if (prev) unit(u, I0{}, I0{}, I1{}, I0{});
                else unit(u, I0{}, I0{}, I0{}, I0{});
            }
        };
        int u = 0;
        for (; u < 4 && u < U; ++u) run(u);
        for (; u + 4 < U; u += 2) {
            unit(u, I0{}, I1{}, I1{}, I2{});
            unit(u + 1, I1{}, I1{}, I1{}, I2{});
        }
        for (; u < U; ++u) run(u);
#pragma unroll
        for (int e = 0; e < 12; ++e) store_row(nch - 1, e);
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
                            if (cls0 + 4 * lq + r < K)
                                __builtin_nontemporal_store((f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]}, reinterpret_cast<f32x4*>(o + orow[r]));
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
