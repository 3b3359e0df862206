// Gram backward, third structure: the 128-row 8-wave kernel of hk_bwd128.h with the staging done by LDS-DMA
// (global_load_lds_dwordx4: HBM -> LDS without passing through registers), for the modes whose A operand is
// (dy + dy^T) * coef / y  (MODE 0 BCNN, MODE 3 signed-sqrt BCNN).
//
// Why (profiles/r2_bwd_lab.json): in the register-staged kernel the MFMA stream + barriers alone run 62.7 us, the
// staging (10 global loads, 10 LDS stores, 8 rcp + 16 mul / fma per thread and K-block, all in the MFMA issue stream of
// a SIMD that holds two waves) costs another 11 us.  Here a K-block is staged by 73 wave-instructions for the whole
// workgroup (9-10 per wave): no staging registers, no ds_write, no address arithmetic per element; what the staging did
// on the way through the registers moves to where the fragment is formed:
//   a = (S1[i][k] + S2[k][i]) * (rcp(Y[i][k]) * coef)        t += Y[i][k] * S1[i][k]   (half-0 waves)
// LDS-DMA writes a wave's 64 x 16 bytes to CONSECUTIVE LDS addresses (M0 base + 16 lane), so a tile cannot be padded;
// bank conflicts of the fragment reads are avoided by a swizzle applied to the SOURCE address of each lane and to the
// READ address (the same involution on both sides, LDS destination linear):
//   S1, Y  [128 i][32 k]  16-byte slot of (i, k4 = k / 4):  i * 8  + (k4 ^ (i & 7))        read: ds_read_b128, 8 lanes =
//                                                                                          8 rows -> 8 distinct slots
//   S2     [32 k][128 i]  slot of (k, i4 = i / 4):          k * 32 + (i4 ^ 4 ((k >> 2) & 1))  read: ds_read_b32, lanes
//                                                          0-31 = k, k + 4 -> the two halves of the banks
//   X      [32 k][HW]     linear (4 k HW = 16 mod 32 at HW = 196: conflict-free as it is)
// Two LDS stages of 73 KB; the pieces of K-block kb + 1 are issued behind MFMA groups 0-4 of K-block kb and are waited
// for by the barrier that ends it (the compiler emits s_waitcnt vmcnt(0) before s_barrier).  The K loop is unrolled by
// two so that every LDS address of a stage is a compile-time offset.
// Same arithmetic per element of dX as hk_bwd128.h and the 64-row kernel: bit-identical dX.  The t partial sums are
// added in a different order (tolerance, not bit-identity).
// Measured (B = 64, C = 512, 14x14, alternating rounds in one process, tools/bwd_ab.py): 64-row kernel 78-80 us,
// register-staged 128-row 74 us, this kernel 69.8 us (0.60 of the fp32 MFMA peak).  Timing-only builds
// (tools/bwd_lab2.py): this kernel fed from L2 only 67.0 us, no staging at all 66.4 us, no per-group fragment reads
// either 62.9 us - the staging itself is now free, streaming the 230 MB from HBM costs ~3 us, the rest is this tiling's
// MFMA stream (13 column tiles for 12.25, 16 barriers, prologue / epilogue).
// Tried and removed: FOUR stages of 16 channels with the pieces issued three K-blocks ahead, an explicit
// s_waitcnt vmcnt(n) + s_barrier instead of __syncthreads (so that only the pieces needed next are waited for) and the
// next K-block's first fragments read before the barrier: 69.9 us / 66.5 us from L2 - the same; the barrier was not
// waiting for HBM.  A diagonal walk of the column blocks (row block I at column block (S - I) mod nI, so that the
// partner row blocks of an image read dy(I, J) and dy(J, I) within the same four K-blocks): 70.5 us against 69.4 us and
// MORE L2 misses (FETCH_SIZE x 2 = 253 MB against 208 MB, profiles/r2_bwd_diag1_fetch.csv) - the four row blocks of an
// image no longer read the same X block at the same time.
#pragma once
#include "hk_bwd128.h"

namespace hk {

// LABV (HK_LAB builds, timing only - results are wrong): 1 = every K-block is staged from the addresses of K-block 0 (the
// pieces come from L2 with a short latency: separates "waiting for HBM at the barrier" from the cost of the staging itself)
template <int HW, int MODE, int LABV = 0>
__global__ __launch_bounds__(512, 2) void bcnn_bwd128d_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                              float* __restrict__ tpart, int C, int nI, int B,
                                                              BwdExtra ex) {
    static_assert(MODE == 0 || MODE == 3, "LDS-DMA staging: modes with the (dy + dy^T) * coef / y operand only");
    constexpr int NT = (HW + 15) / 16;          // 16-column output tiles
    constexpr int NH = (NT + 1) / 2;            // tiles of the first column half (the second has NT - NH)
    constexpr int KB = 32;                      // channels per K-block
    constexpr int T_SZ = 128 * KB;              // floats of one of the three dy / y tiles (16 pieces of 1 KB)
    constexpr int XN4 = KB * HW / 4;            // float4 of one X block
    constexpr int NXP = (XN4 + 63) / 64;        // X pieces (the last one may be partial: clamped source, LDS padding)
    constexpr int X_SZ = NXP * 256;
    constexpr int STAGE = 3 * T_SZ + X_SZ;
    static_assert(NXP <= 32, "X pieces are dealt to the 8 waves four deep");
    HK_DYN_LDS16(lds);

    int b, I;
    if (!xcd_map(blockIdx.x, B, nI, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wrow = (wave & 3) * 32;
    const int half = wave >> 2;                                 // wave-uniform: the tile guards below are scalar branches
    const int nt0 = half * NH, nloc = half ? NT - NH : NH;      // this wave's column tiles: nt0 .. nt0 + nloc - 1
    const long long cc = (long long)b * C * C;
    const float* xb = x + (long long)b * C * HW;
    const int nkb = C / KB;                                     // a multiple of 4 (C % 128 == 0)
    const float in = inv_norm[b];
    const float coef = in * in / (2.0f * (float)HW);
    const float t2 = MODE == 3 ? 2.0f * bwd_t_of(ex, b) : 0.f;

    f32x4 acc[2][NH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tacc = 0.f;

    // ---- sources of this lane's 16 bytes in the pieces its wave issues (32-bit offsets from bases that advance with kb)
    // S1 / Y pieces 2 wave, 2 wave + 1: rows 16 wave + (lane >> 3) (+ 8), slot j = lane & 7 holds k4 = j ^ (row & 7)
    const int o1 = (16 * wave + (lane >> 3)) * C + 4 * ((lane & 7) ^ ((lane >> 3) & 7));
    // S2 pieces 2 wave, 2 wave + 1: k rows 4 wave + (lane >> 5) (+ 2), slot j = lane & 31 holds i4 = j ^ 4 (wave & 1)
    const int o2 = (4 * wave + (lane >> 5)) * C + 4 * ((lane & 31) ^ ((wave & 1) << 2));
    // X pieces wave, wave + 8, wave + 16, wave + 24: float4 index 64 p + lane, clamped inside the block
    int ox[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = 64 * (wave + 8 * u) + lane;
        ox[u] = 4 * (f < XN4 ? f : XN4 - 1);
    }
    const float* ybase = y + cc + (long long)I * 128 * C;      // + kb * KB
    const float* dbase = dy + cc + (long long)I * 128 * C;     // + kb * KB
    const float* tbase = dy + cc + I * 128;                    // + kb * KB * C
    // the pieces of K-block kb into stage `st_` (a compile-time float offset), in five parts (one behind each of the
    // first five MFMA groups); LDS destinations are wave-uniform
#define HK_BD_DMA(kb, st_, part)                                                                              \
    do {                                                                                                       \
        float* S_ = lds + (st_) + 512 * wave;                                                                  \
        if ((part) == 0) { glds16(dbase + (kb) * KB + o1, S_);                                                 \
                           glds16(dbase + (kb) * KB + o1 + 8 * C, S_ + 256); }                                 \
        if ((part) == 1) { glds16(ybase + (kb) * KB + o1, S_ + T_SZ);                                          \
                           glds16(ybase + (kb) * KB + o1 + 8 * C, S_ + T_SZ + 256); }                          \
        if ((part) == 2) { glds16(tbase + (long long)(kb) * KB * C + o2, S_ + 2 * T_SZ);                       \
                           glds16(tbase + (long long)(kb) * KB * C + o2 + 2 * C, S_ + 2 * T_SZ + 256); }       \
        if ((part) == 3) { const float* xk_ = xb + (long long)(kb) * KB * HW;                                  \
                           float* X_ = lds + (st_) + 3 * T_SZ + 256 * wave;                                    \
                           glds16(xk_ + ox[0], X_);                                                            \
                           if (NXP > 8 && wave + 8 < NXP) glds16(xk_ + ox[1], X_ + 2048); }                    \
        if ((part) == 4) { const float* xk_ = xb + (long long)(kb) * KB * HW;                                  \
                           float* X_ = lds + (st_) + 3 * T_SZ + 256 * wave;                                    \
                           if (NXP > 16 && wave + 16 < NXP) glds16(xk_ + ox[2], X_ + 4096);                    \
                           if (NXP > 24 && wave + 24 < NXP) glds16(xk_ + ox[3], X_ + 6144); }                  \
    } while (0)

    // A fragments of the wave's two 16-row blocks for k = 16 s + 4 lq + t, formed from the raw tiles
#define HK_BD_AFRAG(A_, s_)                                                                                    \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
            const int row_ = wrow + i * 16 + l15;                                                              \
            const int sl_ = row_ * 32 + (((4 * (s_) + lq) ^ (row_ & 7)) << 2);                                 \
            f32x4 d1_ = *reinterpret_cast<const f32x4*>(S1 + sl_);                                             \
            const f32x4 yv_ = *reinterpret_cast<const f32x4*>(Yt + sl_);                                       \
            if (MODE == 0 && half == 0)                                                                        \
                tacc += (yv_[0] * d1_[0] + yv_[1] * d1_[1]) + (yv_[2] * d1_[2] + yv_[3] * d1_[3]);             \
            if (MODE == 3) d1_ -= t2 * yv_;       /* (dy_ij + dy_ji - 2 t y_ij) / |y_ij|: the whole -2 t y term here */ \
            const float* s2p_ = S2 + (16 * (s_) + 4 * lq) * 128 + ((((row_ >> 2) ^ ((lq & 1) << 2))) << 2) + (row_ & 3); \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
                float w_;                                                                                      \
                if (MODE == 3) w_ = yv_[t] == 0.f ? 0.f : __builtin_amdgcn_rcpf(fabsf(yv_[t])) * coef;         \
                else w_ = __builtin_amdgcn_rcpf(yv_[t]) * coef;                                                \
                A_[i][t] = (d1_[t] + s2p_[t * 128]) * w_;                                                      \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#define HK_BD_BFRAG(B_, s_, t_)                                                                                \
    do {                                                                                                       \
        const float* bp_ = X + (16 * (s_) + 4 * lq + (t_)) * HW + 16 * nt0 + l15;                              \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) B_[n] = (n < nloc) ? bp_[16 * n] : 0.f;                 \
    } while (0)
#define HK_BD_MFMA(A_, B_, t_)                                                                                 \
    do {                                                                                                       \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) {                                                       \
            if (n < NH - 1 || n < nloc) {                                                                      \
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[0][t_], B_[n], acc[0][n], 0, 0, 0);        \
                acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[1][t_], B_[n], acc[1][n], 0, 0, 0);        \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
    // One K-block out of stage CUR_ (0 / STAGE): eight MFMA groups (s = 0, 1; t = 0..3), the operand fragments of a
    // group read while the previous group's MFMAs run, the pieces of K-block kb_ + 1 issued behind groups 0-4 into the
    // other stage.  LOAD_ is compile-time.
#define HK_BD_KBLOCK(kb_, CUR_, LOAD_)                                                                         \
    do {                                                                                                       \
        const float* S1 = lds + (CUR_);                                                                        \
        const float* Yt = S1 + T_SZ;                                                                           \
        const float* S2 = Yt + T_SZ;                                                                           \
        const float* X = S2 + T_SZ;                                                                            \
        constexpr int NXT_ = STAGE - (CUR_);                                                                   \
        float a0[2][4], a1[2][4], bA[NH], bB[NH];                                                              \
        HK_BD_AFRAG(a0, 0);                                                                                    \
        HK_BD_BFRAG(bA, 0, 0);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const int kn_ = LABV == 1 ? 0 : (kb_) + 1;                                                             \
        HK_BD_BFRAG(bB, 0, 1); HK_BD_MFMA(a0, bA, 0); if (LOAD_) HK_BD_DMA(kn_, NXT_, 0);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_BFRAG(bA, 0, 2); HK_BD_MFMA(a0, bB, 1); if (LOAD_) HK_BD_DMA(kn_, NXT_, 1);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_BFRAG(bB, 0, 3); HK_BD_MFMA(a0, bA, 2); if (LOAD_) HK_BD_DMA(kn_, NXT_, 2);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_AFRAG(a1, 1);                                                                                    \
        HK_BD_BFRAG(bA, 1, 0); HK_BD_MFMA(a0, bB, 3); if (LOAD_) HK_BD_DMA(kn_, NXT_, 3);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_BFRAG(bB, 1, 1); HK_BD_MFMA(a1, bA, 0); if (LOAD_) HK_BD_DMA(kn_, NXT_, 4);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_BFRAG(bA, 1, 2); HK_BD_MFMA(a1, bB, 1);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_BFRAG(bB, 1, 3); HK_BD_MFMA(a1, bA, 2);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BD_MFMA(a1, bB, 3);                                                                                 \
        __syncthreads();                                                                                       \
    } while (0)

    // prologue: K-block 0 into stage 0
#pragma unroll
    for (int part = 0; part < 5; ++part) HK_BD_DMA(0, 0, part);
    __syncthreads();
    int kb = 0;
    for (; kb + 2 < nkb; kb += 2) {                                     // steady state, two K-blocks per trip
        HK_BD_KBLOCK(kb, 0, true);
        HK_BD_KBLOCK(kb + 1, STAGE, true);
    }
    HK_BD_KBLOCK(kb, 0, true);
    HK_BD_KBLOCK(kb + 1, STAGE, false);                                 // last block: nothing left to stage
#undef HK_BD_KBLOCK
#undef HK_BD_MFMA
#undef HK_BD_BFRAG
#undef HK_BD_AFRAG
#undef HK_BD_DMA

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float* dxb = dx + (long long)b * C * HW + (long long)(I * 128 + wrow + i * 16 + lq * 4) * HW;
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int col = 16 * (nt0 + n) + l15;
            if (n < nloc && col < HW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dxb[(long long)r * HW + col] = acc[i][n][r];
            }
        }
    }
    if (MODE == 0) {                                           // t partials: slot 2I (+ a zero in 2I + 1: the consumer
        const float tsum = block_sum<8>(tacc, lds);            // adds C / 64 slots per image); the K loop ended on a barrier
        if (tid == 0) {
            tpart[(long long)b * (2 * nI) + 2 * I] = tsum;
            tpart[(long long)b * (2 * nI) + 2 * I + 1] = 0.f;
        }
    }
}

template <int HW>
static inline size_t bwd128d_lds_bytes() {
    constexpr int nxp = (32 * HW / 4 + 63) / 64;
    return (size_t)2 * (3 * 128 * 32 + nxp * 256) * sizeof(float);
}

// HK_ERR_UNSUPPORTED unless C % 128 == 0 (the caller then takes another kernel)
template <int HW, int MODE>
static int bwd128d_launch(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart,
                          int B, int C, const BwdExtra& ex, hipStream_t st) {
    if (C % 128 != 0 || (long long)C * C >= (1ll << 31) || !aligned16(x) || !aligned16(y) || !aligned16(dy))
        return HK_ERR_UNSUPPORTED;
    const size_t lds = bwd128d_lds_bytes<HW>();
    HK_ALLOW_BIG_LDS((&bcnn_bwd128d_kernel<HW, MODE>), lds);
    const int nI = C / 128;
#ifdef HK_LAB
    if (MODE == 0 && tuning().bwd_v == 10) {
        HK_ALLOW_BIG_LDS((&bcnn_bwd128d_kernel<HW, 0, 1>), lds);
        hipLaunchKernelGGL((bcnn_bwd128d_kernel<HW, 0, 1>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm, dx, tpart, C, nI, B, ex);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
#endif
    hipLaunchKernelGGL((bcnn_bwd128d_kernel<HW, MODE>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm, dx,
                       tpart, C, nI, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
