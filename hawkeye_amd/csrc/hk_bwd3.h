// Gram backward:  dX[I rows] = sum_K P(I,K) X(K), every operand staged by LDS-DMA (128-row blocks, eight waves, two LDS
// stages of 32-channel K-blocks), with what round 2's timing-only builds said that structure paid for removed:
//
//   * the 13th column tile.  HW = 196 = 12 x 16 + 4: thirteen 16-column MFMA tiles do 13 / 12.25 of the work, split 7 / 6
//     over the two waves of a SIMD.  Here the matrix pipe computes columns 0 .. 191 (six tiles per wave: balanced) and
//     the four columns 192 .. 195 are a VALU side product of the A fragments every lane already holds:
//         r[i][c] += a[i][k] * X[k][192 + 2 half + c]        (2 rows x 2 columns per lane, its own k = 16 s + 4 lq + t)
//     - one ds_read_b64 and four v_fma per MFMA group - and the four lq-partials of a row are added at the end
//     (shuffles, fixed tree).  Columns 0 .. 191 are bit-identical to the other backward kernels (same MFMA order); the
//     last four differ in summation order (rounding level).
//   * the store burst.  Every workgroup reaches its epilogue at once; 56 dword stores per lane (64-byte segments) were
//     issue-bound.  The 128 x 196 output block is CONTIGUOUS in dX, so the accumulators go to LDS (free after the K
//     loop) as the HBM image and leave as 49 fully coalesced 16-byte stores per thread.
//   * one kernel for the covariance backward too (MODE 1: P = (g + g^T) / M, 64-row blocks so that C = 256 fills the
//     chip).  X must be centred there, dX = P (X - mu 1^T) = P X - (P mu) 1^T: the vector P mu is one more VALU column,
//         m[i] += a[i][k] * mu[k],
//     subtracted from the whole row while the block is copied out - X is staged raw by LDS-DMA, no mean look-ups.
//
// LDS tiles and swizzles (S1 / Y [rows][32 k] with the 16-byte slot
// i * 8 + (k4 ^ (i & 7)); S2 [32 k][rows] with slot i4 ^ 4 ((k >> 2) & 1); X linear; two stages; the pieces of K-block
// kb + 1 issued behind MFMA groups 0-4 of K-block kb).
// MODE 0 BCNN   a = (dy_ik + dy_ki) * rcp(y_ik) * inv^2 / (2M)       MODE 3 signed-sqrt BCNN (BCNN.py:23-24)
// MODE 1 COV    a = (g_ik + g_ki) / M, X centred through the mu column
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
    // CBP, hk_bwd3c.h with dc == nullptr: dc is computed by the kernel itself from the saved forward state
    const float* cy;     // [B][D]  y
    const float* cdy;    // [B][D]  dL/dy
    const float* ccraw;  // [B][D]  bins before the signed square root
    const float* cinv;   // [B]     1 / max(|u|, 1e-12)
    // BCNN, rank-1 term folded into the GEMM kernel (TK != 0 below)
    const float* colsum; // [B][HW]  column sums of X (the forward's)
    const float* ta;     // TK 1: t[b] = sum_k ta[b][k] (tb2[b][k] - tc[k]), tK terms (tc nullable)
    const float* tb2;
    const float* tc;
    int tK;
    int t_inv2;          // signed sqrt, TK 1: `y` is the UN-normalised u = y / inv (inv_norm = the true 1 / |u|): the coefficient
                         // carries one factor inv less and the kernel's t is inv times the dot product
};

// t = <y, dy> of sample b from its partial sums (every workgroup adds them itself, fixed order)
__device__ __forceinline__ float bwd_t_of(const BwdExtra& ex, int b) {
    float t = 0.f;
    for (int c = 0; c < ex.nt; ++c) t += ex.tb[(long long)b * ex.nt + c];
    return t;
}

// RB: 16-row blocks per wave (2: 128-row workgroup blocks, 1: 64-row).  REMV: the HW % 16 == 4 remainder columns on
// the VALU (false: a last, partly idle MFMA tile like the other kernels).  EPI: LDS-staged 16-byte stores.
// ROWW (128-row blocks): wave w owns rows 16 w .. 16 w + 15 and ALL column tiles, instead of 32 rows x half of the tiles.
// The two waves of a SIMD then form DIFFERENT A fragments (the other split has both of them do the same rcp / mul / add
// on the same rows): on gfx950 an fp32 MFMA runs at the vector-FMA rate and VALU work next to it is time taken from the
// matrix pipe, so the duplicated fragment arithmetic cost ~4 % of a K-block; the price is one ds_read per MFMA instead
// of one per two.  COEFL: the factor inv^2 / 2M multiplies the accumulators once instead of every fragment element
// (rounding-level difference from the other backward kernels).
// TK: where t = <y, dy>, the scalar of the l2-normalisation's backward, comes from.
//   0: BCNN: the kernel adds up its partial sums (-> tpart) and bcnn_rank1_fix_kernel applies the rank-1 term
//      dX -= (t inv^2 / M) 1 colsum^T in a second pass over dX; signed sqrt: from the partial sums ex.tb of a pass over y, dy;
//   1: t is KNOWN before the launch, as a dot product of two small operands (ex.ta / tb2 / tc): when dy = g W comes out
//      of a linear layer on y, <y, dy> = sum_k g_k (logit_k - bias_k).  Every workgroup forms t in its prologue (wave 0,
//      fixed order).  BCNN: it subtracts its rows' share of the rank-1 term while the block is copied out - no second
//      pass, no partial sums of y dy in the K loop (7 VALU ops per fragment quad less next to the MFMAs); signed sqrt
//      (t sits inside the operand P): no pass over y and dy in front of the kernel.
// (Measured and removed, round 5: TK 2, t unknown - the workgroups of an image take a ticket when their block and partial
//  sum are out and the last to arrive makes the pass over the image's dX from L2.  Correct by the device-scope release /
//  acquire recipe, bit-identical to the two-launch route - and slower than it: 85.2 vs 81.4 us at B = 64 (the release
//  writes back 25 MB of freshly dirtied L2 lines and one workgroup streams 2 x 401 KB alone at the end; with
//  __threadfence() in every thread it was 132 us).)
template <int HW, int MODE, int RB, bool REMV, bool EPI, bool ROWW = false, bool COEFL = false, int TK = 0>
__global__ __launch_bounds__(512, 2) void gram_bwd3_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                           float* __restrict__ tpart, int C, int nI, int B, BwdExtra ex) {
    static_assert(MODE == 0 || MODE == 1 || MODE == 3, "operand (dy + dy^T) * w only");
    static_assert(!REMV || HW % 16 == 4, "VALU remainder: four columns");
    static_assert(RB == 1 || RB == 2, "64- or 128-row blocks");
    constexpr bool HAS_Y = MODE == 0 || MODE == 3;
    constexpr bool MUCOL = MODE == 1;
    static_assert(!ROWW || RB == 2, "row-per-wave split: 128-row blocks");
    static_assert(TK == 0 || ((MODE == 0 || MODE == 3) && EPI && HW % 4 == 0), "t handed over: BCNN / signed sqrt, LDS-staged epilogue");
    constexpr int NT = REMV ? HW / 16 : (HW + 15) / 16;   // 16-column MFMA tiles
    constexpr int NH = ROWW ? NT : (NT + 1) / 2;          // tiles of a wave (of the first column half)
    constexpr int RW = ROWW ? 1 : RB;                     // 16-row blocks of a wave
    constexpr int NR = ROWW ? 4 : 2;                      // remainder columns of a wave
    typedef float remv_t __attribute__((ext_vector_type(NR)));
    constexpr int KB = 32;
    constexpr int IB = 64 * RB;                           // rows of a workgroup block
    constexpr int T_SZ = IB * KB;                         // floats of one dy / y tile
    constexpr int XN4 = KB * HW / 4;
    constexpr int NXP = (XN4 + 63) / 64;                  // X pieces of 1 KB (the last may be partial: clamped source)
    constexpr int X_SZ = NXP * 256;
    constexpr int NTILE = HAS_Y ? 3 : 2;
    constexpr int STAGE = NTILE * T_SZ + X_SZ;
    constexpr int O4 = IB * HW / 4;                       // float4 of the output block
    static_assert(NXP <= 32, "X pieces are dealt to the 8 waves four deep");
    static_assert(IB * HW + IB + HW <= 2 * STAGE, "the output image + the mu column + the column sums fit the two stages");
    HK_DYN_LDS16(lds);

    int b, I;
    if (!xcd_map(blockIdx.x, B, nI, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wrow = ROWW ? wave * 16 : (wave & 3) * (16 * RB);
    const int half = ROWW ? 0 : wave >> 2;                      // wave-uniform
    // this wave's MFMA column tiles nt0 .. nt0 + nloc - 1 (an even tile count: the same number in both halves, compile-time)
    const int nt0 = half * NH, nloc = (ROWW || NT % 2 == 0) ? NH : (half ? NT - NH : NH);
    const bool do_t = ROWW || half == 0;                        // the waves that add up t = <y, dy> (each row once)
    const long long cc = (long long)b * C * C;
    const float* xb = x + (long long)b * C * HW;
    const int nkb = C / KB;                                     // even (C % 64 == 0)
    float coef = 1.0f / (float)HW;
    if (HAS_Y) {
        const float in = inv_norm[b];
        coef = ((MODE == 3 && TK == 1 && ex.t_inv2) ? in : in * in) / (2.0f * (float)HW);
    }
    const float cfrag = COEFL ? 1.0f : coef;                    // factor applied per fragment element
    // signed sqrt: t = <y, dy> from its partial sums - or (TK 1) from the dot product wave 0 forms below
    float t2 = (MODE == 3 && TK != 1) ? 2.0f * bwd_t_of(ex, b) : 0.f;

    f32x4 acc[RW][NH];
    float rem[RW][NR], mcol[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NR; ++c) rem[i][c] = 0.f;
        mcol[i] = 0.f;
    }
    float tacc = 0.f;

    // the sample's channel means behind the two stages (covariance): read as mu[k .. k + 3] per MFMA group
    float* mus = lds + 2 * STAGE;
    if (MUCOL) {
        for (int e = tid; e < C; e += 512) mus[e] = ex.mu[(long long)b * C + e];
    }
    // TK 1: t = sum_k ta[b][k] (tb2[b][k] - tc[k]) by wave 0 - lane l takes k = l, l + 64, .. in order, then a fixed
    // butterfly - into the word behind the two stages (published by the prologue's barrier)
    float csr = 0.f;                                            // TK 1: this thread's column sum, fetched here, used in the epilogue
    if (TK == 1) {
        if (wave == 0) {
            float p = 0.f;
            for (int k = lane; k < ex.tK; k += 64) {
                const long long o = (long long)b * ex.tK + k;
                p = fmaf(ex.ta[o], ex.tb2[o] - (ex.tc ? ex.tc[k] : 0.f), p);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m, 64);
            if (lane == 0) mus[0] = p;
        }
        if (MODE == 0 && tid < HW) csr = ex.colsum[(long long)b * HW + tid];
    }

    // ---- this lane's 16 bytes in the pieces its wave issues (32-bit offsets; bases advance with kb)
    // S1 / Y [IB][32]: piece p = rows 8 p .. 8 p + 7; RB = 2: pieces 2 wave, 2 wave + 1; RB = 1: piece wave
    // (BYTE offsets, unsigned: wave-uniform base pointer + zero-extended 32-bit lane offset is the saddr form of
    //  global_load_lds - no 64-bit address arithmetic or register pairs per piece)
    const unsigned prow = (RB == 2 ? 16 : 8) * wave + (lane >> 3);
    const unsigned o1 = 4u * (prow * C + 4 * ((lane & 7) ^ ((lane >> 3) & 7)));
    // S2 [32][IB]: RB = 2: piece p = k rows 2 p, 2 p + 1, pieces 2 wave, 2 wave + 1 (k rows 4 wave + (lane >> 5) (+ 2));
    //              RB = 1: piece wave = k rows 4 wave + (lane >> 4); slot j holds i4 = j ^ 4 ((k >> 2) & 1) = j ^ 4 (wave & 1)
    const unsigned o2 = 4u * (RB == 2 ? (4 * wave + (lane >> 5)) * C + 4 * ((lane & 31) ^ ((wave & 1) << 2))
                                      : (4 * wave + (lane >> 4)) * C + 4 * ((lane & 15) ^ ((wave & 1) << 2)));
    unsigned ox[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = 64 * (wave + 8 * u) + lane;
        ox[u] = 16u * (f < XN4 ? f : XN4 - 1);
    }
    const char* ybase = HAS_Y ? reinterpret_cast<const char*>(y + cc + (long long)I * IB * C) : nullptr;   // + kb * KB floats
    const char* dbase = reinterpret_cast<const char*>(dy + cc + (long long)I * IB * C);                    // + kb * KB floats
    const char* tbase = reinterpret_cast<const char*>(dy + cc + I * IB);                                   // + kb * KB * C floats
    const char* xbase = reinterpret_cast<const char*>(xb);                                                 // + kb * KB * HW floats
    constexpr int PW = RB == 2 ? 512 : 256;                                  // floats of a tile a wave fills
    constexpr int OY = T_SZ, OS2 = (NTILE - 1) * T_SZ, OX = NTILE * T_SZ;
    const long long row8 = 32ll * C, krow2 = 8ll * C;                        // bytes: eight rows of dy / two channel rows
#define HK_B3_G(base_, off_, l_) glds16(reinterpret_cast<const float*>((base_) + (off_)), l_)
#define HK_B3_DMA(kb, st_, part)                                                                               \
    do {                                                                                                       \
        float* S_ = lds + (st_) + PW * wave;                                                                   \
        if ((part) == 0) { const char* p_ = dbase + (long long)(kb) * (KB * 4);                                \
                           HK_B3_G(p_, o1, S_);                                                                \
                           if (RB == 2) HK_B3_G(p_ + row8, o1, S_ + 256); }                                    \
        if ((part) == 1 && HAS_Y) { const char* p_ = ybase + (long long)(kb) * (KB * 4);                       \
                           HK_B3_G(p_, o1, S_ + OY);                                                           \
                           if (RB == 2) HK_B3_G(p_ + row8, o1, S_ + OY + 256); }                               \
        if ((part) == 2) { const char* p_ = tbase + (long long)(kb) * KB * C * 4;                              \
                           HK_B3_G(p_, o2, S_ + OS2);                                                          \
                           if (RB == 2) HK_B3_G(p_ + krow2, o2, S_ + OS2 + 256); }                             \
        if ((part) == 3) { const char* xk_ = xbase + (long long)(kb) * (KB * HW * 4);                          \
                           float* X_ = lds + (st_) + OX + 256 * wave;                                          \
                           HK_B3_G(xk_, ox[0], X_);                                                            \
                           if (NXP > 8 && wave + 8 < NXP) HK_B3_G(xk_, ox[1], X_ + 2048); }                    \
        if ((part) == 4) { const char* xk_ = xbase + (long long)(kb) * (KB * HW * 4);                          \
                           float* X_ = lds + (st_) + OX + 256 * wave;                                          \
                           if (NXP > 16 && wave + 16 < NXP) HK_B3_G(xk_, ox[2], X_ + 4096);                    \
                           if (NXP > 24 && wave + 24 < NXP) HK_B3_G(xk_, ox[3], X_ + 6144); }                  \
    } while (0)

    // A fragments of the wave's 16-row blocks for k = 16 s + 4 lq + t, formed from the raw tiles; the mu column rides along
#define HK_B3_AFRAG(A_, s_, kb_)                                                                               \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RW; ++i) {                                                       \
            const int row_ = wrow + i * 16 + l15;                                                              \
            const int sl_ = row_ * 32 + (((4 * (s_) + lq) ^ (row_ & 7)) << 2);                                 \
            f32x4 d1_ = *reinterpret_cast<const f32x4*>(S1 + sl_);                                             \
            f32x4 yv_ = (f32x4){1.f, 1.f, 1.f, 1.f};                                                           \
            if (HAS_Y) yv_ = *reinterpret_cast<const f32x4*>(Yt + sl_);                                        \
            if (MODE == 0 && do_t && TK != 1)                                                                  \
                tacc += (yv_[0] * d1_[0] + yv_[1] * d1_[1]) + (yv_[2] * d1_[2] + yv_[3] * d1_[3]);             \
            if (MODE == 3) d1_ -= t2 * yv_;                                                                    \
            const float* s2p_ = S2 + (16 * (s_) + 4 * lq) * IB + ((((row_ >> 2) ^ ((lq & 1) << 2))) << 2) + (row_ & 3); \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
                float w_ = cfrag;                                                                              \
                if (MODE == 3) w_ = yv_[t] == 0.f ? 0.f : (COEFL ? __builtin_amdgcn_rcpf(fabsf(yv_[t])) : __builtin_amdgcn_rcpf(fabsf(yv_[t])) * coef); \
                if (MODE == 0) w_ = COEFL ? __builtin_amdgcn_rcpf(yv_[t]) : __builtin_amdgcn_rcpf(yv_[t]) * coef; \
                A_[i][t] = (d1_[t] + s2p_[t * IB]) * w_;                                                       \
            }                                                                                                  \
        }                                                                                                      \
        if (MUCOL && do_t) {                                                                                   \
            const f32x4 mu_ = *reinterpret_cast<const f32x4*>(mus + (kb_) * KB + 16 * (s_) + 4 * lq);          \
            _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                     \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) HK_FMAC_PINNED(mcol[i], A_[i][t], mu_[t]);       \
        }                                                                                                      \
    } while (0)
    // B fragments of MFMA group (s, t): X[k][16 (nt0 + n) + l15], and the remainder columns X[k][16 NT + 2 half + 0 / 1]
#define HK_B3_BFRAG(B_, R_, s_, t_)                                                                            \
    do {                                                                                                       \
        const float* xr_ = X + (16 * (s_) + 4 * lq + (t_)) * HW;                                               \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) B_[n] = (n < nloc) ? xr_[16 * (nt0 + n) + l15] : 0.f;   \
        if (REMV) R_ = *reinterpret_cast<const remv_t*>(xr_ + 16 * NT + 2 * half);                             \
    } while (0)
#define HK_B3_MFMA(A_, B_, R_, t_)                                                                             \
    do {                                                                                                       \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) {                                                       \
            if (n < NH - 1 || n < nloc) {                                                                      \
                _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                 \
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][t_], B_[n], acc[i][n], 0, 0, 0);    \
            }                                                                                                  \
        }                                                                                                      \
        if (REMV) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                     \
                _Pragma("unroll") for (int c = 0; c < NR; ++c) HK_FMAC_PINNED(rem[i][c], A_[i][t_], R_[c]);    \
        }                                                                                                      \
    } while (0)
    // One K-block out of stage CUR_ (0 / STAGE): eight MFMA groups (s = 0, 1; t = 0..3), the fragments of a group read
    // while the previous group's MFMAs run, the pieces of K-block kb_ + 1 issued behind groups 0-4 into the other stage.
#define HK_B3_KBLOCK(kb_, CUR_, LOAD_)                                                                         \
    do {                                                                                                       \
        const float* S1 = lds + (CUR_);                                                                        \
        const float* Yt = S1 + OY;                                                                             \
        const float* S2 = S1 + OS2;                                                                            \
        const float* X = S1 + OX;                                                                              \
        constexpr int NXT_ = STAGE - (CUR_);                                                                   \
        float a0[RW][4], a1[RW][4], bA[NH], bB[NH];                                                            \
        remv_t rA = (remv_t)(0.f), rB = (remv_t)(0.f);                                                         \
        HK_B3_AFRAG(a0, 0, kb_);                                                                               \
        HK_B3_BFRAG(bA, rA, 0, 0);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const int kn_ = (kb_) + 1;                                                                             \
        HK_B3_BFRAG(bB, rB, 0, 1); HK_B3_MFMA(a0, bA, rA, 0); if (LOAD_) HK_B3_DMA(kn_, NXT_, 0);              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_BFRAG(bA, rA, 0, 2); HK_B3_MFMA(a0, bB, rB, 1); if (LOAD_) HK_B3_DMA(kn_, NXT_, 1);              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_BFRAG(bB, rB, 0, 3); HK_B3_MFMA(a0, bA, rA, 2); if (LOAD_) HK_B3_DMA(kn_, NXT_, 2);              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_AFRAG(a1, 1, kb_);                                                                               \
        HK_B3_BFRAG(bA, rA, 1, 0); HK_B3_MFMA(a0, bB, rB, 3); if (LOAD_) HK_B3_DMA(kn_, NXT_, 3);              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_BFRAG(bB, rB, 1, 1); HK_B3_MFMA(a1, bA, rA, 0); if (LOAD_) HK_B3_DMA(kn_, NXT_, 4);              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_BFRAG(bA, rA, 1, 2); HK_B3_MFMA(a1, bB, rB, 1);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_BFRAG(bB, rB, 1, 3); HK_B3_MFMA(a1, bA, rA, 2);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_B3_MFMA(a1, bB, rB, 3);                                                                             \
        __syncthreads();                                                                                       \
    } while (0)

    // prologue: K-block 0 into stage 0 (the barrier also publishes mus)
#pragma unroll
    for (int part = 0; part < 5; ++part) HK_B3_DMA(0, 0, part);
    __syncthreads();
    const float kfix1 = (TK == 1 && MODE == 0) ? mus[0] * inv_norm[b] * inv_norm[b] / (float)HW : 0.f;      // (bcnn_rank1_fix_kernel's k)
    if (TK == 1 && MODE == 3) t2 = 2.0f * mus[0] * (ex.t_inv2 ? inv_norm[b] : 1.0f);
    int kb = 0;
    for (; kb + 2 < nkb; kb += 2) {                                     // steady state, two K-blocks per trip
        HK_B3_KBLOCK(kb, 0, true);
        HK_B3_KBLOCK(kb + 1, STAGE, true);
    }
    HK_B3_KBLOCK(kb, 0, true);
    HK_B3_KBLOCK(kb + 1, STAGE, false);                                 // last block: nothing left to stage
#undef HK_B3_KBLOCK
#undef HK_B3_MFMA
#undef HK_B3_BFRAG
#undef HK_B3_AFRAG
#undef HK_B3_DMA
#undef HK_B3_G

    // the four lq-partials of a row's side columns: p0 + p1, then + (p2 + p3), the same value in every lane of the row
    if (REMV) {
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                float v = rem[i][c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                rem[i][c] = COEFL ? v * coef : v;
            }
    }
    if (COEFL) {
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int n = 0; n < NH; ++n) acc[i][n] *= coef;
    }
    if (MUCOL) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            float v = mcol[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            mcol[i] = COEFL ? v * coef : v;
        }
    }

    float* dxb = dx + (long long)b * C * HW + (long long)I * IB * HW;       // the block: IB x HW contiguous floats
    if (EPI) {
        // (the K loop ended on a barrier: both stages are free) the block as it lies in HBM, then 16-byte stores
        float* O = lds;
        float* CM = lds + IB * HW;                                          // [IB] (P mu) of the block's rows
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            float* orow = O + (wrow + i * 16 + lq * 4) * HW + 16 * nt0 + l15;
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                if (n < nloc && (REMV || 16 * (nt0 + n) + l15 < HW)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) orow[r * HW + 16 * n] = acc[i][n][r];
                }
            }
            if (REMV && lq == 0) {
                remv_t rv_;
#pragma unroll
                for (int c = 0; c < NR; ++c) rv_[c] = rem[i][c];
                *reinterpret_cast<remv_t*>(O + (wrow + i * 16 + l15) * HW + 16 * NT + 2 * half) = rv_;
            }
            if (MUCOL && do_t && lq == 0) CM[wrow + i * 16 + l15] = mcol[i];
        }
        float* CS = CM + IB;                                                // [HW] k colsum (TK 1)
        if (TK == 1 && MODE == 0 && tid < HW) CS[tid] = kfix1 * csr;
        __syncthreads();
        const f32x4* o4 = reinterpret_cast<const f32x4*>(O);
        f32x4* g4 = reinterpret_cast<f32x4*>(dxb);
#pragma unroll
        for (int u = 0; u < (O4 + 511) / 512; ++u) {
            const int f = tid + 512 * u;
            if (f < O4) {
                f32x4 v = o4[f];
                if (MUCOL) v -= CM[(4 * f) / HW];
                if (TK == 1 && MODE == 0) v -= reinterpret_cast<const f32x4*>(CS)[f % (HW / 4)];
                g4[f] = v;
            }
        }
    } else {
        if (MUCOL) {                                                        // every lane needs (P mu) of ITS accumulator rows
            float* CM = lds;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RW; ++i)
                if (do_t && lq == 0) CM[wrow + i * 16 + l15] = mcol[i];
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            float* drow = dxb + (long long)(wrow + i * 16 + lq * 4) * HW;
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                const int col = 16 * (nt0 + n) + l15;
                if (n < nloc && col < HW) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        drow[(long long)r * HW + col] = acc[i][n][r] - (MUCOL ? lds[wrow + i * 16 + lq * 4 + r] : 0.f);
                }
            }
            if (REMV && lq == 0) {
                const float m_ = MUCOL ? lds[wrow + i * 16 + l15] : 0.f;
                remv_t rv_;
#pragma unroll
                for (int c = 0; c < NR; ++c) rv_[c] = rem[i][c] - m_;
                *reinterpret_cast<remv_t*>(dxb + (long long)(wrow + i * 16 + l15) * HW + 16 * NT + 2 * half) = rv_;
            }
        }
    }
    if (MODE == 0 && TK != 1) {                                // t partials: C / 64 slots per image (zero-filled for RB = 2)
        __syncthreads();
        const float tsum = block_sum<8>(tacc, lds);
        if (tid == 0) {
            if (RB == 2) {
                tpart[(long long)b * (2 * nI) + 2 * I] = tsum;
                tpart[(long long)b * (2 * nI) + 2 * I + 1] = 0.f;
            } else {
                tpart[(long long)b * nI + I] = tsum;
            }
        }
    }
}

template <int HW, int MODE, int RB>
static inline size_t bwd3_lds_bytes(int C) {
    constexpr int nxp = (32 * HW / 4 + 63) / 64;
    constexpr int ntile = (MODE == 0 || MODE == 3) ? 3 : 2;
    return ((size_t)2 * (ntile * 64 * RB * 32 + nxp * 256) + (MODE == 1 ? (size_t)C : 4)) * sizeof(float);
}

// HK_ERR_UNSUPPORTED unless C % (64 RB) == 0 and the operands are 16-byte aligned (the caller then takes another kernel).
template <int HW, int MODE, int RB, int TK = 0>
static int bwd3_launch(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart,
                       int B, int C, const BwdExtra& ex, hipStream_t st) {
    if (C % (64 * RB) != 0 || (long long)C * C >= (1ll << 31) || !aligned16(x) || !aligned16(dy) || !aligned16(dx) ||
        ((MODE == 0 || MODE == 3) && !aligned16(y)))
        return HK_ERR_UNSUPPORTED;
    const size_t lds = bwd3_lds_bytes<HW, MODE, RB>(C);
    if (lds > 160 * 1024) return HK_ERR_UNSUPPORTED;
    const int nI = C / (64 * RB);
    const dim3 grid(xcd_grid(B, nI));
    // the shipped form: VALU remainder columns where HW % 16 == 4, LDS-staged epilogue; with 128-row blocks a wave owns 16
    // rows and all column tiles and the coefficient multiplies the accumulators once (each step
    // measured - 71.5 -> 66.9 -> 65.3 -> 64.1 us at B = 64, C = 512, 14 x 14)
    constexpr bool REM = HW % 16 == 4;
    constexpr bool ROWW = REM && RB == 2;
    HK_ALLOW_BIG_LDS((&gram_bwd3_kernel<HW, MODE, RB, REM, true, ROWW, ROWW, TK>), lds);
    hipLaunchKernelGGL((gram_bwd3_kernel<HW, MODE, RB, REM, true, ROWW, ROWW, TK>), grid, dim3(512), lds, st, x, y, dy, inv_norm, dx,
                       tpart, C, nI, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
