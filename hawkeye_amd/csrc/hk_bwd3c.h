// Compact-bilinear backward GEMM:  dX[I rows] = sum_K P(I,K) X(K),
//     P_ik = s1_i s2_k dc[(h1_i + h2_k) mod D] + s1_k s2_i dc[(h1_k + h2_i) mod D]        (dG + dG^T, CBCNN.py backward)
// on the K-block pipeline of hk_bwd3.h (128- or 64-row blocks, eight waves, two LDS stages, X staged by LDS-DMA,
// six MFMA column tiles per wave + the HW % 16 == 4 remainder columns on the VALU, LDS-staged 16-byte epilogue).
// Round 2 ran this mode on the register-staged kernels (hk_bwd128.h: X through registers, 88 us at B = 64) and, at the
// yaml batch of 16, on the four-wave panel kernel with 128 workgroups on 256 CUs (66 us).
//
// P is not in memory: the sample's dc vector (24 KB) and the hash / sign tables live in LDS and every thread GENERATES
// its 16 bytes of the next K-block's P tile ([rows][32 k], the swizzled layout the A fragments are read from) behind the
// MFMA groups of the current one: per element two gathers from dc, two additions mod D (x + y, then min(z, z - D) on
// unsigned byte offsets: no compare / select), two three-way XORs for the signs (sign bits of s1, s2 as masks) and one
// add.  The A fragment is then a plain ds_read_b128 - no arithmetic in the MFMA stream's own fragment path.
//
#pragma once
#include "hk_bwd3.h"

namespace hk {

// NSPLIT = 2: the column tiles of a row block are divided between two workgroups (each generates the whole P tile, each
// stores its own columns: nothing to add up) - how a batch of 16 fills the chip without atomics.
template <int HW, int RB, bool REMV, int NSPLIT>
__global__ __launch_bounds__(512, 2) void cbp_bwd3_kernel(const float* __restrict__ x, float* __restrict__ dx, int C, int nI,
                                                          int B, BwdExtra ex) {
    static_assert(!REMV || HW % 16 == 4, "VALU remainder: four columns");
    constexpr int NTA = REMV ? HW / 16 : (HW + 15) / 16;  // 16-column MFMA tiles of the row block
    static_assert(NTA % NSPLIT == 0, "column split: an even number of tiles");
    constexpr int NT = NTA / NSPLIT;                      // ... of this workgroup
    constexpr int NH = (NT + 1) / 2;
    constexpr int KB = 32;
    constexpr int IB = 64 * RB;
    constexpr int T_SZ = IB * KB;                         // floats of the P tile
    constexpr int XN4 = KB * HW / 4;
    constexpr int NXP = (XN4 + 63) / 64;
    constexpr int X_SZ = NXP * 256;
    constexpr int STAGE = T_SZ + X_SZ;
    constexpr int O4 = IB * HW / 4;
    static_assert(NXP <= 32, "X pieces are dealt to the 8 waves four deep");
    HK_DYN_LDS16(lds);

    int b, w;
    if (!xcd_map(blockIdx.x, B, nI * NSPLIT, b, w)) return;
    const int I = w / NSPLIT, nh = w % NSPLIT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wrow = (wave & 3) * (16 * RB);
    const int half = wave >> 2;
    const int nt0 = nh * NT + half * NH, nloc = (NT % 2 == 0) ? NH : (half ? NT - NH : NH);
    const bool do_rem = REMV && nh == NSPLIT - 1;               // the remainder columns belong to the last column split
    const float* xb = x + (long long)b * C * HW;
    const int nkbl = C / KB;                                    // K-blocks (even)
    constexpr int kb0 = 0;
    const unsigned D4 = 4u * (unsigned)ex.D;

    f32x4 acc[RB][NH];
    float rem[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rem[i][0] = rem[i][1] = 0.f;
    }

    // behind the two stages: dc of the sample, then 4 h1, 4 h2 (byte offsets into dc), sign masks of s1, s2
    float* dcl = lds + 2 * STAGE;
    unsigned* th1 = reinterpret_cast<unsigned*>(dcl + ((ex.D + 3) / 4) * 4);
    unsigned* th2 = th1 + C;
    unsigned* sb1 = th2 + C;
    unsigned* sb2 = sb1 + C;
    if (ex.dc) {
        const float* dcb = ex.dc + (long long)b * ex.D;
        for (int e = tid; e < ex.D; e += 512) dcl[e] = dcb[e];
    } else {
        // dc of the sample from the forward's saved state, every workgroup for itself (CBCNN.py:132-133 backwards):
        //     t = <y, dy> ;  du = (dy - y t) / n ;  dc = du / (2 sqrt(|c| + 1e-10)), 0 where c == 0 (torch: sign' = 0)
        // - 6000 bins: cheaper than a separate launch and its round trip.  Fixed summation order: all workgroups of a
        // sample get the same t, bit for bit.
        const long long o = (long long)b * ex.D;
        constexpr int NE = 16;                             // bins per thread held in registers (D <= 512 * NE: launch check)
        float yv[NE], dv[NE], cv[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {                     // all loads in flight at once: every workgroup of the grid is
            const int e = tid + 512 * j;                   // in this prologue at the same time, nothing else covers it
            const bool ok = e < ex.D;
            yv[j] = ok ? ex.cy[o + e] : 0.f;
            dv[j] = ok ? ex.cdy[o + e] : 0.f;
            cv[j] = ok ? ex.ccraw[o + e] : 0.f;
        }
        float tt = 0.f;
#pragma unroll
        for (int j = 0; j < NE; ++j) tt += yv[j] * dv[j];
        const float t = block_sum<8>(tt, lds);             // (scratch: the stages are not in use yet)
        const float in = ex.cinv[b];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = tid + 512 * j;
            const float du = (dv[j] - yv[j] * t) * in;
            if (e < ex.D) dcl[e] = (cv[j] != 0.f) ? du / (2.0f * sqrtf(fabsf(cv[j]) + 1e-10f)) : 0.f;
        }
    }
    {
        for (int e = tid; e < C; e += 512) {
            th1[e] = 4u * (unsigned)ex.h1[e];
            th2[e] = 4u * (unsigned)ex.h2[e];
            sb1[e] = ex.s1[e] < 0.f ? 0x80000000u : 0u;
            sb2[e] = ex.s2[e] < 0.f ? 0x80000000u : 0u;
        }
    }
    // this thread's rows of the P tile (r, and r + 64 for 128-row blocks) and its four channels 4 k4 .. 4 k4 + 3 of a K-block
    const int pr = tid >> 3, k4 = tid & 7;
    unsigned h1i[RB], h2i[RB], s1i[RB], s2i[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int i = I * IB + pr + 64 * u;
        h1i[u] = 4u * (unsigned)ex.h1[i];
        h2i[u] = 4u * (unsigned)ex.h2[i];
        s1i[u] = ex.s1[i] < 0.f ? 0x80000000u : 0u;
        s2i[u] = ex.s2[i] < 0.f ? 0x80000000u : 0u;
    }
    unsigned ox[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = 64 * (wave + 8 * u) + lane;
        ox[u] = 16u * (f < XN4 ? f : XN4 - 1);
    }
    const char* xbase = reinterpret_cast<const char*>(xb);
    const char* dcb8 = reinterpret_cast<const char*>(dcl);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // P tile of K-block kb into stage st_, in parts: 0 = the channel tables of this thread's four k, 1 / 3 = addresses and
    // gathers of row u = 0 / 1, 2 / 4 = combine and store.  X block by LDS-DMA in parts 3 and 4 (as in hk_bwd3.h).
    u32x4 tk1, tk2, sk1, sk2;
    float ga[RB][4], gb[RB][4];
#define HK_BC_TAB(kb)                                                                                          \
    do {                                                                                                       \
        const int k_ = (kb) * KB + 4 * k4;                                                                     \
        tk1 = *reinterpret_cast<const u32x4*>(th1 + k_);                                                       \
        tk2 = *reinterpret_cast<const u32x4*>(th2 + k_);                                                       \
        sk1 = *reinterpret_cast<const u32x4*>(sb1 + k_);                                                       \
        sk2 = *reinterpret_cast<const u32x4*>(sb2 + k_);                                                       \
    } while (0)
#define HK_BC_GATHER(u_)                                                                                       \
    do {                                                                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                        \
            unsigned a_ = h1i[u_] + tk2[t];                                                                    \
            a_ = a_ < a_ - D4 ? a_ : a_ - D4;              /* (h1_i + h2_k) mod D on byte offsets: v_min_u32 */   \
            unsigned b_ = tk1[t] + h2i[u_];                                                                    \
            b_ = b_ < b_ - D4 ? b_ : b_ - D4;                                                                  \
            ga[u_][t] = *HK_LDS_CONST(dcb8 + a_);                                                              \
            gb[u_][t] = *HK_LDS_CONST(dcb8 + b_);                                                              \
        }                                                                                                      \
    } while (0)
#define HK_BC_STORE(st_, u_)                                                                                   \
    do {                                                                                                       \
        f32x4 p_;                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                          \
            p_[t] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ga[u_][t]) ^ s1i[u_] ^ sk2[t]) +    \
                    __builtin_bit_cast(float, __builtin_bit_cast(unsigned, gb[u_][t]) ^ sk1[t] ^ s2i[u_]);     \
        const int row_ = pr + 64 * (u_);                                                                       \
        *reinterpret_cast<f32x4*>(lds + (st_) + row_ * 32 + ((k4 ^ (row_ & 7)) << 2)) = p_;                    \
    } while (0)
#define HK_BC_G(base_, off_, l_) glds16(reinterpret_cast<const float*>((base_) + (off_)), l_)
#define HK_BC_STAGE(kb, st_, part)                                                                             \
    do {                                                                                                       \
        if ((part) == 0) HK_BC_TAB(kb);                                                                        \
        if ((part) == 1) HK_BC_GATHER(0);                                                                      \
        if ((part) == 2) { HK_BC_STORE(st_, 0); if (RB == 2) HK_BC_GATHER(RB - 1); }                           \
        if ((part) == 3) { if (RB == 2) HK_BC_STORE(st_, RB - 1);                                              \
                           const char* xk_ = xbase + (long long)(kb) * (KB * HW * 4);                          \
                           float* X_ = lds + (st_) + T_SZ + 256 * wave;                                        \
                           HK_BC_G(xk_, ox[0], X_);                                                            \
                           if (NXP > 8 && wave + 8 < NXP) HK_BC_G(xk_, ox[1], X_ + 2048); }                    \
        if ((part) == 4) { const char* xk_ = xbase + (long long)(kb) * (KB * HW * 4);                          \
                           float* X_ = lds + (st_) + T_SZ + 256 * wave;                                        \
                           if (NXP > 16 && wave + 16 < NXP) HK_BC_G(xk_, ox[2], X_ + 4096);                    \
                           if (NXP > 24 && wave + 24 < NXP) HK_BC_G(xk_, ox[3], X_ + 6144); }                  \
    } while (0)

#define HK_BC_AFRAG(A_, s_)                                                                                    \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                       \
            const int row_ = wrow + i * 16 + l15;                                                              \
            const f32x4 d1_ = *reinterpret_cast<const f32x4*>(S1 + row_ * 32 + (((4 * (s_) + lq) ^ (row_ & 7)) << 2)); \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) A_[i][t] = d1_[t];                                   \
        }                                                                                                      \
    } while (0)
#define HK_BC_BFRAG(B_, R_, s_, t_)                                                                            \
    do {                                                                                                       \
        const float* xr_ = X + (16 * (s_) + 4 * lq + (t_)) * HW;                                               \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) B_[n] = (n < nloc) ? xr_[16 * (nt0 + n) + l15] : 0.f;   \
        if (REMV) R_ = *reinterpret_cast<const f32x2*>(xr_ + 16 * NTA + 2 * half);                             \
    } while (0)
#define HK_BC_MFMA(A_, B_, R_, t_)                                                                             \
    do {                                                                                                       \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) {                                                       \
            if (n < NH - 1 || n < nloc) {                                                                      \
                _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                 \
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][t_], B_[n], acc[i][n], 0, 0, 0);    \
            }                                                                                                  \
        }                                                                                                      \
        if (REMV) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                   \
                HK_FMAC_PINNED(rem[i][0], A_[i][t_], R_[0]);                                                   \
                HK_FMAC_PINNED(rem[i][1], A_[i][t_], R_[1]);                                                   \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#define HK_BC_KBLOCK(kb_, CUR_, LOAD_)                                                                         \
    do {                                                                                                       \
        const float* S1 = lds + (CUR_);                                                                        \
        const float* X = S1 + T_SZ;                                                                            \
        constexpr int NXT_ = STAGE - (CUR_);                                                                   \
        float a0[RB][4], a1[RB][4], bA[NH], bB[NH];                                                            \
        f32x2 rA = (f32x2){0.f, 0.f}, rB = (f32x2){0.f, 0.f};                                                  \
        HK_BC_AFRAG(a0, 0);                                                                                    \
        HK_BC_BFRAG(bA, rA, 0, 0);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const int kn_ = (kb_) + 1;                                                                             \
        HK_BC_BFRAG(bB, rB, 0, 1); HK_BC_MFMA(a0, bA, rA, 0); if (LOAD_) HK_BC_STAGE(kn_, NXT_, 0);            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_BFRAG(bA, rA, 0, 2); HK_BC_MFMA(a0, bB, rB, 1); if (LOAD_) HK_BC_STAGE(kn_, NXT_, 1);            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_BFRAG(bB, rB, 0, 3); HK_BC_MFMA(a0, bA, rA, 2); if (LOAD_) HK_BC_STAGE(kn_, NXT_, 2);            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_AFRAG(a1, 1);                                                                                    \
        HK_BC_BFRAG(bA, rA, 1, 0); HK_BC_MFMA(a0, bB, rB, 3); if (LOAD_) HK_BC_STAGE(kn_, NXT_, 3);            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_BFRAG(bB, rB, 1, 1); HK_BC_MFMA(a1, bA, rA, 0); if (LOAD_) HK_BC_STAGE(kn_, NXT_, 4);            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_BFRAG(bA, rA, 1, 2); HK_BC_MFMA(a1, bB, rB, 1);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_BFRAG(bB, rB, 1, 3); HK_BC_MFMA(a1, bA, rA, 2);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BC_MFMA(a1, bB, rB, 3);                                                                             \
        __syncthreads();                                                                                       \
    } while (0)

    __syncthreads();                                                    // dc and the tables are in LDS
#pragma unroll
    for (int part = 0; part < 5; ++part) HK_BC_STAGE(kb0, 0, part);     // prologue: K-block kb0 into stage 0
    __syncthreads();
    int kb = kb0;
    for (; kb + 2 < kb0 + nkbl; kb += 2) {
        HK_BC_KBLOCK(kb, 0, true);
        HK_BC_KBLOCK(kb + 1, STAGE, true);
    }
    HK_BC_KBLOCK(kb, 0, true);
    HK_BC_KBLOCK(kb + 1, STAGE, false);
#undef HK_BC_KBLOCK
#undef HK_BC_MFMA
#undef HK_BC_BFRAG
#undef HK_BC_AFRAG
#undef HK_BC_STAGE
#undef HK_BC_G
#undef HK_BC_STORE
#undef HK_BC_GATHER
#undef HK_BC_TAB

    if (REMV) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v = rem[i][c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                rem[i][c] = v;
            }
    }
    // the block as it lies in HBM (both stages are free: the K loop ended on a barrier), then 16-byte stores
    float* dxb = dx + (long long)b * C * HW + (long long)I * IB * HW;
    float* O = lds;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float* orow = O + (wrow + i * 16 + lq * 4) * HW + 16 * nt0 + l15;
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            if (n < nloc && (REMV || 16 * (nt0 + n) + l15 < HW)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) orow[r * HW + 16 * n] = acc[i][n][r];
            }
        }
        if (do_rem && lq == 0)
            *reinterpret_cast<f32x2*>(O + (wrow + i * 16 + l15) * HW + 16 * NTA + 2 * half) = (f32x2){rem[i][0], rem[i][1]};
    }
    __syncthreads();
    if (NSPLIT > 1) {                                   // this workgroup's columns [c0, c1) of every row (multiples of 4)
        const int c0 = 16 * nh * NT, c1 = nh == NSPLIT - 1 ? HW : 16 * (nh + 1) * NT;
        const int w4 = (c1 - c0) / 4;
        for (int f = tid; f < IB * w4; f += 512) {
            const int r = f / w4, c = c0 + 4 * (f % w4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(O + r * HW + c);
            *reinterpret_cast<f32x4*>(dxb + (long long)r * HW + c) = v;
        }
    } else {
        const f32x4* o4 = reinterpret_cast<const f32x4*>(O);
        f32x4* g4 = reinterpret_cast<f32x4*>(dxb);
#pragma unroll
        for (int u = 0; u < (O4 + 511) / 512; ++u) {
            const int f = tid + 512 * u;
            if (f < O4) g4[f] = o4[f];
        }
    }
}

template <int HW, int RB>
static inline size_t cbp_bwd3_lds_bytes(int C, int D) {
    constexpr int nxp = (32 * HW / 4 + 63) / 64;
    const size_t loop = (size_t)2 * (64 * RB * 32 + nxp * 256) + (size_t)((D + 3) / 4) * 4 + 4 * (size_t)C;
    const size_t image = (size_t)64 * RB * HW;              // the epilogue's output image reuses everything
    return (loop > image ? loop : image) * sizeof(float);
}

// HK_ERR_UNSUPPORTED when the shape is not covered.  nsplit 1 or 2 (2: only where the row block has an even number of
// MFMA column tiles)
template <int HW, int RB>
static int cbp_bwd3_launch(const float* x, float* dx, int B, int C, const BwdExtra& ex, int nsplit, hipStream_t st) {
    if (C % (64 * RB) != 0 || !aligned16(x) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    const size_t lds = cbp_bwd3_lds_bytes<HW, RB>(C, ex.D);
    if (lds > 160 * 1024 || (!ex.dc && ex.D > 512 * 16)) return HK_ERR_UNSUPPORTED;
    constexpr bool REMV = HW % 16 == 4;
    constexpr int NTA = REMV ? HW / 16 : (HW + 15) / 16;
    if (nsplit == 2 && NTA % 2 != 0) return HK_ERR_UNSUPPORTED;
    const int nI = C / (64 * RB);
    const dim3 grid(xcd_grid(B, nI * nsplit));
    if constexpr (NTA % 2 == 0) {
        if (nsplit == 2) {
            HK_ALLOW_BIG_LDS((&cbp_bwd3_kernel<HW, RB, REMV, 2>), lds);
            hipLaunchKernelGGL((cbp_bwd3_kernel<HW, RB, REMV, 2>), grid, dim3(512), lds, st, x, dx, C, nI, B, ex);
            HK_LAUNCH_CHECK();
            return HK_OK;
        }
    }
    HK_ALLOW_BIG_LDS((&cbp_bwd3_kernel<HW, RB, REMV, 1>), lds);
    hipLaunchKernelGGL((cbp_bwd3_kernel<HW, RB, REMV, 1>), grid, dim3(512), lds, st, x, dx, C, nI, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
