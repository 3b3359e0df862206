// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32 fma chain, 157 TF/s peak) with pluggable operand loaders and epilogues.
//
//   C[b] (M x N) = epilogue( sum_k A[b](i,k) * B[b](k,j) )
//
// Tiling: 64x64 (or 128x128 for large M,N) output tile per 256-thread workgroup
// (4 waves, 2x2, each TxT 32x32 MFMA accumulators), K in chunks of 32,
// register-staged global->LDS double buffer, one barrier per chunk.
//
// Operand storage is described by a compile-time flag per operand:
//   *_KC = true : memory is [mn][k], k contiguous  (A row-major / B given as N x K, "NT")
//   *_KC = false: memory is [k][mn], mn contiguous (A given transposed / B row-major K x N, "NN")
// LDS keeps the tile in memory order.  K-contiguous tiles are read with one
// ds_read_b128 per lane (row pitch 36 floats: pitch/4 odd -> conflict-free for
// the 16-lane groups of ds_read_b128); a lane's 4 values feed 4 consecutive
// MFMAs, so within an 8-wide k-step lanes 0-31 own k = 8s+t and lanes 32-63
// own k = 8s+4+t.  The same k assignment is used for mn-contiguous tiles
// (ds_read_b32, 32 consecutive floats per half-wave).  The summation order over
// k is therefore a fixed permutation of 0..K-1: deterministic, fp32 exact fma.
#pragma once
#include "hk_common.h"

namespace hk {

// ---------------------------------------------------------------- loaders
// Loader concept:  float4 ld4(int b, int r, int c)  -- 4 consecutive elements
// along the contiguous dimension, starting at (slow index r, fast index c),
// c % 4 == 0, zero-filled outside the operand.  begin()/finish() are hooks.
struct LdPlain {
    const float* p;
    long long bs;  // batch stride (elements)
    int ld, R, C;  // leading dim, #rows (slow), #cols (fast)
    int vec;       // 1: base/ld/bs allow aligned float4 loads
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) {
            const float* q = p + (long long)b * bs + (long long)r * ld + c;
            if (vec && c + 3 < C) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                v.x = q[0];
                if (c + 1 < C) v.y = q[1];
                if (c + 2 < C) v.z = q[2];
                if (c + 3 < C) v.w = q[3];
            }
        }
        return v;
    }
};

// Plain operand minus a per-(batch,row) scalar (spatial centring for the
// covariance: rows are channels, mu[b][row] the channel mean).  Padding stays 0.
struct LdRowSub {
    LdPlain base;
    const float* mu;
    long long mubs;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float4 v = base.ld4(b, r, c);
        if (r < base.R && c < base.C) {
            const float m = mu[(long long)b * mubs + r];
            v.x -= m;
            if (c + 1 < base.C) v.y -= m;
            if (c + 2 < base.C) v.z -= m;
            if (c + 3 < base.C) v.w -= m;
        }
        return v;
    }
};

// Square operand symmetrised on the fly: S = G + G^T (covariance backward,
// MPNCOV.py:131).  The transposed read is strided but L2-resident (d x d x 4 B).
struct LdSym {
    const float* p;
    long long bs;
    int d;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < d) {
            const float* q = p + (long long)b * bs;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (c + t < d) v[t] = q[(long long)r * d + c + t] + q[(long long)(c + t) * d + r];
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// ---------------------------------------------------------------- epilogues
// C = alpha * bscale[b] * acc + beta * C + diag * I   (optionally stored transposed)
struct EpAffine {
    float* c;
    long long bs;
    int ld;
    float alpha;
    const float* bscale;  // nullable, per batch
    float beta;
    float diag;
    int trans;
    __device__ __forceinline__ void operator()(int b, int i, int j, float v) const {
        const float s = bscale ? alpha * bscale[b] : alpha;
        float* q = c + (long long)b * bs + (trans ? (long long)j * ld + i : (long long)i * ld + j);
        float r = s * v;
        if (i == j) r += diag;
        if (beta != 0.f) r += beta * (*q);
        *q = r;
    }
};

// ---------------------------------------------------------------- kernel
// T = MFMA tiles per wave in each direction: T=1 -> 64x64 workgroup tile (16 acc VGPRs, 37 KB LDS, 4 WGs/CU),
// T=2 -> 128x128 (64 acc VGPRs, 71 KB LDS, 2 WGs/CU; 16 MFMAs per pair of operand reads).  Measured on the
// Newton-Schulz shape (256^3 x 64 samples): T=2 gives 256 workgroups = 1 per CU with nothing to overlap its
// load/barrier phases and is 1.3x SLOWER than T=1 (1024 workgroups, 4 per CU), so T=1 is the default;
// allow_big=1 opts in (useful only when tiles >> CUs).
//
// SYM = true (square problems whose RESULT is symmetric, e.g. products of commuting symmetric matrices in the
// Newton-Schulz chain): only the tilesM (tilesM + 1) / 2 tiles on or above the diagonal are computed and every
// off-diagonal tile is also stored mirrored - 10 instead of 16 tiles at d = 256.  The epilogue's beta * C term then
// reads C[j][i], so C must be symmetric as well.
template <int T, int BKT, bool A_KC, bool B_KC, class AL, class BL, class EP, bool SYM = false>
__global__ __launch_bounds__(256) void bgemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int nb,
                                                     int tilesM, int tilesN) {
    constexpr int BM = 64 * T, BN = 64 * T, BK = BKT;
    constexpr int PA = A_KC ? BK + 4 : BM + 4;
    constexpr int PB = B_KC ? BK + 4 : BN + 4;
    constexpr int SA = (A_KC ? BM : BK) * PA;
    constexpr int SB = (B_KC ? BN : BK) * PB;
    constexpr int NLA = BM * BK / 4 / 256;      // float4 per thread per operand tile (2 or 4)
    constexpr int NLB = BN * BK / 4 / 256;
    constexpr int A4 = A_KC ? BK / 4 : BM / 4;  // float4 per staged row
    constexpr int B4 = B_KC ? BK / 4 : BN / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];

    int b, tile;
    if (!xcd_map(blockIdx.x, nb, SYM ? tilesM * (tilesM + 1) / 2 : tilesM * tilesN, b, tile)) return;
    int tm = tile / tilesN, tn = tile % tilesN;
    if (SYM) {                                    // row-major enumeration of the upper triangle
        tm = 0;
        int rem = tile;
        while (rem >= tilesM - tm) { rem -= tilesM - tm; ++tm; }
        tn = tm + rem;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    al.begin(b, tm, tn);
    bl.begin(b, tm, tn);

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 acc2;                                  // second k-chain (T == 1 only)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

    float4 ra[NLA], rb[NLB];

// (macro-local names carry a trailing underscore: the argument expressions mention the caller's `c`)
#define HK_GLOAD(k0)                                                                                   \
    do {                                                                                               \
        _Pragma("unroll") for (int u = 0; u < NLA; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / A4, c_ = f_ % A4;                                       \
            ra[u] = A_KC ? al.ld4(b, m0 + r_, (k0) + 4 * c_) : al.ld4(b, (k0) + r_, m0 + 4 * c_);          \
        }                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NLB; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / B4, c_ = f_ % B4;                                       \
            rb[u] = B_KC ? bl.ld4(b, n0 + r_, (k0) + 4 * c_) : bl.ld4(b, (k0) + r_, n0 + 4 * c_);          \
        }                                                                                              \
    } while (0)

#define HK_SSTORE(buf)                                                                                 \
    do {                                                                                               \
        float* As_ = lds + (buf) * (SA + SB);                                                          \
        float* Bs_ = As_ + SA;                                                                         \
        _Pragma("unroll") for (int u = 0; u < NLA; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / A4, c_ = f_ % A4;                                       \
            *reinterpret_cast<float4*>(&As_[r_ * PA + 4 * c_]) = ra[u];                                  \
        }                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NLB; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / B4, c_ = f_ % B4;                                       \
            *reinterpret_cast<float4*>(&Bs_[r_ * PB + 4 * c_]) = rb[u];                                  \
        }                                                                                              \
    } while (0)

    const int nk = (K + BK - 1) / BK;
    HK_GLOAD(0);
    HK_SSTORE(0);
    __syncthreads();

    for (int c = 0; c < nk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nk) HK_GLOAD((c + 1) * BK);
        const float* As = lds + cur * (SA + SB);
        const float* Bs = As + SA;
        // operand fragments of step s+1 are fetched before the MFMAs of step s are issued (the dependent MFMAs block
        // the wave's issue for >100 cycles); for T == 1 the single 32x32 accumulator is split into two independent
        // k-chains (summed in the epilogue) so that the matrix pipe is not paced by one dependency chain
        float a[T][4], bb[T][4];
#define HK_FRAG(s_, A_, B_)                                                                                    \
        do {                                                                                                       \
            _Pragma("unroll") for (int i = 0; i < T; ++i) {                                                        \
                const int row_ = (wm * T + i) * 32 + l31;                                                          \
                if (A_KC) {                                                                                        \
                    const float4 v_ = *reinterpret_cast<const float4*>(&As[row_ * PA + 8 * (s_) + 4 * lh]);        \
                    A_[i][0] = v_.x; A_[i][1] = v_.y; A_[i][2] = v_.z; A_[i][3] = v_.w;                            \
                } else {                                                                                           \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) A_[i][t] = As[(8 * (s_) + 4 * lh + t) * PA + row_]; \
                }                                                                                                  \
            }                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < T; ++j) {                                                        \
                const int col_ = (wn * T + j) * 32 + l31;                                                          \
                if (B_KC) {                                                                                        \
                    const float4 v_ = *reinterpret_cast<const float4*>(&Bs[col_ * PB + 8 * (s_) + 4 * lh]);        \
                    B_[j][0] = v_.x; B_[j][1] = v_.y; B_[j][2] = v_.z; B_[j][3] = v_.w;                            \
                } else {                                                                                           \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) B_[j][t] = Bs[(8 * (s_) + 4 * lh + t) * PB + col_]; \
                }                                                                                                  \
            }                                                                                                      \
        } while (0)
        HK_FRAG(0, a, bb);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float an[T][4], bn[T][4];
            if (s + 1 < BK / 8) HK_FRAG(s + 1, an, bn);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        if (T == 1 && (t & 1))
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], bb[j][t], acc2, 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], bb[j][t], acc[i][j], 0, 0, 0);
                    }
            if (s + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { a[i][t] = an[i][t]; bb[i][t] = bn[i][t]; }
            }
        }
#undef HK_FRAG
        if (c + 1 < nk) HK_SSTORE(cur ^ 1);
        __syncthreads();
    }
#undef HK_GLOAD
#undef HK_SSTORE

    if (T == 1) acc[0][0] += acc2;
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int jj = n0 + (wn * T + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ii = m0 + (wm * T + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ii < M && jj < N) {
                    ep(b, ii, jj, acc[i][j][r]);
                    if (SYM && tm != tn) ep(b, jj, ii, acc[i][j][r]);
                }
            }
        }
    al.finish(b, tm, tn, tilesM, lds);
}

// ---------------------------------------------------------------- 128x128 tile, 8 waves
// Why: on the Newton-Schulz shape (256^3 x 64 samples) the 64x64 kernel above reads 128 KB of operands per 2.1 MFLOP
// tile = 16 FLOP/B, i.e. ~10 TB/s of L2->LDS traffic at the fp32 MFMA peak - it is L2-bandwidth/latency bound (45 % of
// wave time parked at the chunk barrier, profiles/r1c_sq_wait_counters.csv).  The first 128x128 attempt (T = 2 above:
// 4 waves x 64x64) halves that traffic but leaves one 4-wave workgroup per CU with a one-chunk prefetch, and was
// 1.3x slower.  This variant keeps the 32 FLOP/B of the big tile and fixes the latency side: 8 waves per workgroup
// (wave = 32 rows x 64 columns: two independent accumulators, so consecutive MFMAs never depend on each other),
// and a TWO-chunk prefetch through two register sets - the loads for chunk c+2 are issued before chunk c is computed
// and are stored to LDS a whole chunk later.  256 tiles at B = 64: exactly one workgroup per CU.
// Same k-permutation and summation order as bgemm_kernel.  Opt-in (HK_NS_GEMM=4) until it has a measured number.
// BMN = 128: the kernel described above.  BMN = 64 (HK_NS_GEMM=5): the same two-chunk prefetch on the 64x64 tile of
// bgemm_kernel (4 waves of 32x32, two k-chains per wave, 4 workgroups per CU) - isolates the effect of the deeper
// prefetch from the effect of the bigger tile.
template <int BMN, bool A_KC, bool B_KC, class AL, class BL, class EP>
__global__ __launch_bounds__(BMN == 128 ? 512 : 256) void bgemm_p2_kernel(AL al, BL bl, EP ep, int M, int N, int K,
                                                                          int nb, int tilesM, int tilesN) {
    static_assert(BMN == 128 || BMN == 64, "tile is 128x128 (8 waves) or 64x64 (4 waves)");
    constexpr int BM = BMN, BN = BMN, BK = 32;
    constexpr int NTH = BMN == 128 ? 512 : 256;
    constexpr int WCOLS = BMN == 128 ? 64 : 32;          // columns per wave
    constexpr int PA = A_KC ? BK + 4 : BM + 4;
    constexpr int PB = B_KC ? BK + 4 : BN + 4;
    constexpr int SA = (A_KC ? BM : BK) * PA;
    constexpr int SB = (B_KC ? BN : BK) * PB;
    constexpr int NL = BM * BK / 4 / NTH;       // float4 per thread per operand chunk (2)
    constexpr int A4 = A_KC ? BK / 4 : BM / 4;
    constexpr int B4 = B_KC ? BK / 4 : BN / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];

    int b, tile;
    if (!xcd_map(blockIdx.x, nb, tilesM * tilesN, b, tile)) return;
    const int tm = tile / tilesN, tn = tile % tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;    // (BMN/32) x 2 waves: rows wm*32.., columns wn*WCOLS..
    const int l31 = lane & 31, lh = lane >> 5;

    al.begin(b, tm, tn);
    bl.begin(b, tm, tn);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float4 ra0[NL], rb0[NL], ra1[NL], rb1[NL];  // two register sets: chunks in flight

#define HK_GLOAD2(RA, RB, k0)                                                                          \
    do {                                                                                               \
        _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                               \
            const int f_ = tid + NTH * u, ar_ = f_ / A4, ac_ = f_ % A4, br_ = f_ / B4, bc_ = f_ % B4;  \
            RA[u] = A_KC ? al.ld4(b, m0 + ar_, (k0) + 4 * ac_) : al.ld4(b, (k0) + ar_, m0 + 4 * ac_);  \
            RB[u] = B_KC ? bl.ld4(b, n0 + br_, (k0) + 4 * bc_) : bl.ld4(b, (k0) + br_, n0 + 4 * bc_);  \
        }                                                                                              \
    } while (0)
#define HK_SSTORE2(RA, RB, buf)                                                                        \
    do {                                                                                               \
        float* As_ = lds + (buf) * (SA + SB);                                                          \
        float* Bs_ = As_ + SA;                                                                         \
        _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                               \
            const int f_ = tid + NTH * u, ar_ = f_ / A4, ac_ = f_ % A4, br_ = f_ / B4, bc_ = f_ % B4;  \
            *reinterpret_cast<float4*>(&As_[ar_ * PA + 4 * ac_]) = RA[u];                              \
            *reinterpret_cast<float4*>(&Bs_[br_ * PB + 4 * bc_]) = RB[u];                              \
        }                                                                                              \
    } while (0)
#define HK_COMPUTE2(buf)                                                                               \
    do {                                                                                               \
        const float* As = lds + (buf) * (SA + SB);                                                     \
        const float* Bs = As + SA;                                                                     \
        const int row_ = wm * 32 + l31, c0_ = wn * WCOLS + l31, c1_ = WCOLS == 64 ? c0_ + 32 : c0_;          \
        _Pragma("unroll") for (int s = 0; s < BK / 8; ++s) {                                           \
            float a_[4], p_[4], q_[4];                                                                 \
            if (A_KC) {                                                                                \
                const float4 v_ = *reinterpret_cast<const float4*>(&As[row_ * PA + 8 * s + 4 * lh]);   \
                a_[0] = v_.x; a_[1] = v_.y; a_[2] = v_.z; a_[3] = v_.w;                                \
            } else {                                                                                   \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) a_[t] = As[(8 * s + 4 * lh + t) * PA + row_]; \
            }                                                                                          \
            if (B_KC) {                                                                                \
                const float4 v_ = *reinterpret_cast<const float4*>(&Bs[c0_ * PB + 8 * s + 4 * lh]);    \
                const float4 w_ = *reinterpret_cast<const float4*>(&Bs[c1_ * PB + 8 * s + 4 * lh]);    \
                p_[0] = v_.x; p_[1] = v_.y; p_[2] = v_.z; p_[3] = v_.w;                                \
                q_[0] = w_.x; q_[1] = w_.y; q_[2] = w_.z; q_[3] = w_.w;                                \
            } else {                                                                                   \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                        \
                    p_[t] = Bs[(8 * s + 4 * lh + t) * PB + c0_];                                       \
                    q_[t] = Bs[(8 * s + 4 * lh + t) * PB + c1_];                                       \
                }                                                                                      \
            }                                                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
                if (WCOLS == 64) {      /* two column tiles */                                         \
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t], p_[t], acc0, 0, 0, 0);          \
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t], q_[t], acc1, 0, 0, 0);          \
                } else if (t & 1) {     /* one tile, two k-chains (as bgemm_kernel) */                  \
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t], p_[t], acc1, 0, 0, 0);          \
                } else {                                                                               \
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t], p_[t], acc0, 0, 0, 0);          \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    } while (0)

    const int nk = (K + BK - 1) / BK;
    HK_GLOAD2(ra0, rb0, 0);
    HK_SSTORE2(ra0, rb0, 0);
    if (nk > 1) HK_GLOAD2(ra1, rb1, BK);
    __syncthreads();
    // iteration c computes chunk c from stage c & 1; register set (c + 1) & 1 holds chunk c + 1, set c & 1 is free
    for (int c = 0; c < nk; c += 2) {
        if (c + 2 < nk) HK_GLOAD2(ra0, rb0, (c + 2) * BK);
        HK_COMPUTE2(0);
        if (c + 1 < nk) HK_SSTORE2(ra1, rb1, 1);
        __syncthreads();
        if (c + 1 < nk) {
            if (c + 3 < nk) HK_GLOAD2(ra1, rb1, (c + 3) * BK);
            HK_COMPUTE2(1);
            if (c + 2 < nk) HK_SSTORE2(ra0, rb0, 0);
            __syncthreads();
        }
    }
#undef HK_GLOAD2
#undef HK_SSTORE2
#undef HK_COMPUTE2

    if (WCOLS == 32) acc0 += acc1;
    const int ib = m0 + wm * 32 + 4 * lh;
    const int j0 = n0 + wn * WCOLS + l31, j1 = j0 + 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ii = ib + (r & 3) + 8 * (r >> 2);
        if (ii < M) {
            if (j0 < N) ep(b, ii, j0, acc0[r]);
            if (WCOLS == 64 && j1 < N) ep(b, ii, j1, acc1[r]);
        }
    }
    al.finish(b, tm, tn, tilesM, lds);
}

// ---------------------------------------------------------------- fp32 product on the bf16 matrix pipe (split operands)
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA.  An fp32 value splits EXACTLY into three
// bf16 pieces a = a1 + a2 + a3 (8 mantissa bits each: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); the
// subtractions are exact in fp32), every piece product is exact in the fp32 accumulator, and
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-32 |a b|)
// so SIX bf16 MFMAs per 16-deep k step reproduce the fp32 product to the accumulator's own rounding (NT = 6: 16/6 =
// 2.7x the f32-MFMA rate), and THREE keep everything above 2^-16 |a b| (NT = 3: 5.3x, relative error ~1e-5 per product
// before averaging).  Terms are added smallest first.  The operands are split once, when the tile is staged into
// LDS (three bf16 planes per operand; B is transposed on the way in so that both fragments are 16-byte reads).
// Opt-in for the Newton-Schulz products (HK_NS_GEMM=6 / 7): written after round 1's GPU budget was spent; the
// emulation models the operand layout as "lane l holds A[i = l % 32][k = 8 (l / 32) .. + 7]" (B alike), which still
// has to be confirmed on the device - a wrong layout fails the parity tests loudly, it cannot pass by accident.
typedef __bf16 hk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short hk_u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short hk_u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {           // round to nearest even (finite inputs)
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void bf16_split3(float a, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
    p1 = bf16_rne(a);
    const float r1 = a - bf16_f32(p1);
    p2 = bf16_rne(r1);
    p3 = bf16_rne(r1 - bf16_f32(p2));
}

template <int NT, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void bgemm_bf16split_kernel(AL al, BL bl, EP ep, int M, int N, int K, int nb, int tilesM,
                                                              int tilesN) {
    static_assert(NT == 3 || NT == 6, "three or six piece products");
    constexpr int BM = 64, BN = 64, BK = 32, PK = BK + 8;      // 80-byte rows: 16-byte aligned fragments
    constexpr int PLANE = 64 * PK;                               // one bf16 plane of one operand
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 2 * 3 * PLANE];   // [stage][A|B][piece]

    int b, tile;
    if (!xcd_map(blockIdx.x, nb, tilesM * tilesN, b, tile)) return;
    const int tm = tile / tilesN, tn = tile % tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 ra[2], rb[2];                                         // A: [m][k] k-contiguous ; B: [k][n] n-contiguous
#define HK_GLOADX(k0)                                                                   \
    do {                                                                                \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                 \
            const int f_ = tid + 256 * u;                                               \
            ra[u] = al.ld4(b, m0 + f_ / 8, (k0) + 4 * (f_ % 8));                        \
            rb[u] = bl.ld4(b, (k0) + f_ / 16, n0 + 4 * (f_ % 16));                      \
        }                                                                               \
    } while (0)
#define HK_SSTOREX(buf)                                                                 \
    do {                                                                                \
        unsigned short* As_ = lds + (buf) * 6 * PLANE;                                  \
        unsigned short* Bs_ = As_ + 3 * PLANE;                                          \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                 \
            const int f_ = tid + 256 * u;                                               \
            const float av_[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};                  \
            const float bv_[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};                  \
            hk_u16x4 p1_, p2_, p3_;                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                             \
                unsigned short x1_, x2_, x3_;                                           \
                bf16_split3(av_[t], x1_, x2_, x3_);                                     \
                p1_[t] = x1_; p2_[t] = x2_; p3_[t] = x3_;                               \
            }                                                                           \
            const int ao_ = (f_ / 8) * PK + 4 * (f_ % 8);           /* row m, 4 consecutive k */ \
            *reinterpret_cast<hk_u16x4*>(&As_[ao_]) = p1_;                              \
            *reinterpret_cast<hk_u16x4*>(&As_[PLANE + ao_]) = p2_;                      \
            *reinterpret_cast<hk_u16x4*>(&As_[2 * PLANE + ao_]) = p3_;                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {         /* B transposed: [n][k] */ \
                unsigned short x1_, x2_, x3_;                                           \
                bf16_split3(bv_[t], x1_, x2_, x3_);                                     \
                const int bo_ = (4 * (f_ % 16) + t) * PK + f_ / 16;                     \
                Bs_[bo_] = x1_; Bs_[PLANE + bo_] = x2_; Bs_[2 * PLANE + bo_] = x3_;     \
            }                                                                           \
        }                                                                               \
    } while (0)

    const int nk = (K + BK - 1) / BK;
    HK_GLOADX(0);
    HK_SSTOREX(0);
    __syncthreads();
    for (int c = 0; c < nk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nk) HK_GLOADX((c + 1) * BK);
        const unsigned short* As = lds + cur * 6 * PLANE;
        const unsigned short* Bs = As + 3 * PLANE;
        const int arow = (wm * 32 + l31) * PK, bcol = (wn * 32 + l31) * PK;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int ko = 16 * s + 8 * lh;
            hk_bf16x8 a[3], q[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = __builtin_bit_cast(hk_bf16x8, *reinterpret_cast<const hk_u16x8*>(&As[p * PLANE + arow + ko]));
                q[p] = __builtin_bit_cast(hk_bf16x8, *reinterpret_cast<const hk_u16x8*>(&Bs[p * PLANE + bcol + ko]));
            }
            if (NT == 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], q[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], q[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], q[2], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], q[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], q[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], q[0], acc, 0, 0, 0);
        }
        if (c + 1 < nk) HK_SSTOREX(cur ^ 1);
        __syncthreads();
    }
#undef HK_GLOADX
#undef HK_SSTOREX

    const int jj = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ii = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ii < M && jj < N) ep(b, ii, jj, acc[r]);
    }
}

// The same split products on the 128x128 / 8-wave / two-chunk-prefetch structure of bgemm_p2_kernel<128>: half the
// split work per multiply-add of the 64x64 tile (the split costs ~10 VALU operations per staged element, more than the
// six MFMAs of a 64x64x32 chunk take), 32 FLOP/B of L2 traffic, 120 KB of LDS (one 8-wave workgroup per CU).
template <int NT, class AL, class BL, class EP>
__global__ __launch_bounds__(512) void bgemm_bf16split128_kernel(AL al, BL bl, EP ep, int M, int N, int K, int nb,
                                                                 int tilesM, int tilesN) {
    static_assert(NT == 3 || NT == 6, "three or six piece products");
    constexpr int BM = 128, BN = 128, BK = 32, PK = BK + 8;
    constexpr int PLANE = 128 * PK;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 2 * 3 * PLANE];   // [stage][A|B][piece] = 120 KB

    int b, tile;
    if (!xcd_map(blockIdx.x, nb, tilesM * tilesN, b, tile)) return;
    const int tm = tile / tilesN, tn = tile % tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves: 32 rows x 64 columns each
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float4 ra0[2], rb0[2], ra1[2], rb1[2];                       // two register sets: chunks in flight
#define HK_GLOADY(RA, RB, k0)                                                           \
    do {                                                                                \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                 \
            const int f_ = tid + 512 * u;                                               \
            RA[u] = al.ld4(b, m0 + f_ / 8, (k0) + 4 * (f_ % 8));                        \
            RB[u] = bl.ld4(b, (k0) + f_ / 32, n0 + 4 * (f_ % 32));                      \
        }                                                                               \
    } while (0)
#define HK_SSTOREY(RA, RB, buf)                                                         \
    do {                                                                                \
        unsigned short* As_ = lds + (buf) * 6 * PLANE;                                  \
        unsigned short* Bs_ = As_ + 3 * PLANE;                                          \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                 \
            const int f_ = tid + 512 * u;                                               \
            const float av_[4] = {RA[u].x, RA[u].y, RA[u].z, RA[u].w};                  \
            const float bv_[4] = {RB[u].x, RB[u].y, RB[u].z, RB[u].w};                  \
            hk_u16x4 p1_, p2_, p3_;                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                             \
                unsigned short x1_, x2_, x3_;                                           \
                bf16_split3(av_[t], x1_, x2_, x3_);                                     \
                p1_[t] = x1_; p2_[t] = x2_; p3_[t] = x3_;                               \
            }                                                                           \
            const int ao_ = (f_ / 8) * PK + 4 * (f_ % 8);                               \
            *reinterpret_cast<hk_u16x4*>(&As_[ao_]) = p1_;                              \
            *reinterpret_cast<hk_u16x4*>(&As_[PLANE + ao_]) = p2_;                      \
            *reinterpret_cast<hk_u16x4*>(&As_[2 * PLANE + ao_]) = p3_;                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {         /* B transposed: [n][k] */ \
                unsigned short x1_, x2_, x3_;                                           \
                bf16_split3(bv_[t], x1_, x2_, x3_);                                     \
                const int bo_ = (4 * (f_ % 32) + t) * PK + f_ / 32;                     \
                Bs_[bo_] = x1_; Bs_[PLANE + bo_] = x2_; Bs_[2 * PLANE + bo_] = x3_;     \
            }                                                                           \
        }                                                                               \
    } while (0)
#define HK_COMPUTEY(buf)                                                                \
    do {                                                                                \
        const unsigned short* As = lds + (buf) * 6 * PLANE;                             \
        const unsigned short* Bs = As + 3 * PLANE;                                      \
        const int arow_ = (wm * 32 + l31) * PK, c0_ = (wn * 64 + l31) * PK, c1_ = c0_ + 32 * PK; \
        _Pragma("unroll") for (int s = 0; s < BK / 16; ++s) {                           \
            const int ko_ = 16 * s + 8 * lh;                                            \
            hk_bf16x8 a_[3], p_[3], q_[3];                                              \
            _Pragma("unroll") for (int e = 0; e < 3; ++e) {                             \
                a_[e] = __builtin_bit_cast(hk_bf16x8, *reinterpret_cast<const hk_u16x8*>(&As[e * PLANE + arow_ + ko_])); \
                p_[e] = __builtin_bit_cast(hk_bf16x8, *reinterpret_cast<const hk_u16x8*>(&Bs[e * PLANE + c0_ + ko_]));   \
                q_[e] = __builtin_bit_cast(hk_bf16x8, *reinterpret_cast<const hk_u16x8*>(&Bs[e * PLANE + c1_ + ko_]));   \
            }                                                                           \
            if (NT == 6) {                                                              \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], p_[0], acc0, 0, 0, 0); \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], q_[0], acc1, 0, 0, 0); \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], p_[1], acc0, 0, 0, 0); \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], q_[1], acc1, 0, 0, 0); \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], p_[2], acc0, 0, 0, 0); \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], q_[2], acc1, 0, 0, 0); \
            }                                                                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], p_[0], acc0, 0, 0, 0);     \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], q_[0], acc1, 0, 0, 0);     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], p_[1], acc0, 0, 0, 0);     \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], q_[1], acc1, 0, 0, 0);     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], p_[0], acc0, 0, 0, 0);     \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], q_[0], acc1, 0, 0, 0);     \
        }                                                                               \
    } while (0)

    const int nk = (K + BK - 1) / BK;
    HK_GLOADY(ra0, rb0, 0);
    HK_SSTOREY(ra0, rb0, 0);
    if (nk > 1) HK_GLOADY(ra1, rb1, BK);
    __syncthreads();
    for (int c = 0; c < nk; c += 2) {                            // same two-register-set schedule as bgemm_p2_kernel
        if (c + 2 < nk) HK_GLOADY(ra0, rb0, (c + 2) * BK);
        HK_COMPUTEY(0);
        if (c + 1 < nk) HK_SSTOREY(ra1, rb1, 1);
        __syncthreads();
        if (c + 1 < nk) {
            if (c + 3 < nk) HK_GLOADY(ra1, rb1, (c + 3) * BK);
            HK_COMPUTEY(1);
            if (c + 2 < nk) HK_SSTOREY(ra0, rb0, 0);
            __syncthreads();
        }
    }
#undef HK_GLOADY
#undef HK_SSTOREY
#undef HK_COMPUTEY

    const int ib = m0 + wm * 32 + 4 * lh;
    const int j0 = n0 + wn * 64 + l31, j1 = j0 + 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ii = ib + (r & 3) + 8 * (r >> 2);
        if (ii < M) {
            if (j0 < N) ep(b, ii, j0, acc0[r]);
            if (j1 < N) ep(b, ii, j1, acc1[r]);
        }
    }
}

template <int NT, class AL, class BL, class EP>
static inline int bgemm_bf16split128_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb,
                                            hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    hipLaunchKernelGGL((bgemm_bf16split128_kernel<NT, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(512), 0, st, al, bl, ep,
                       M, N, K, nb, tm, tn);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// A row-major [M][K], B row-major [K][N] (the Newton-Schulz `mm` form)
template <int NT, class AL, class BL, class EP>
static inline int bgemm_bf16split_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb,
                                         hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    hipLaunchKernelGGL((bgemm_bf16split_kernel<NT, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st, al, bl, ep, M,
                       N, K, nb, tm, tn);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

template <bool A_KC, bool B_KC, class AL, class BL, class EP>
static inline int bgemm_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb,
                               hipStream_t st, int allow_big = 0) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    if (allow_big == 1 && M >= 128 && N >= 128) {
        const int tm = (M + 127) / 128, tn = (N + 127) / 128;
        hipLaunchKernelGGL((bgemm_kernel<2, 32, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st,
                           al, bl, ep, M, N, K, nb, tm, tn);
    } else if (allow_big == 2 && K >= 128) {   // 64-deep K chunks: half the barriers, 70 KB LDS (2 WGs/CU)
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((bgemm_kernel<1, 64, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st,
                           al, bl, ep, M, N, K, nb, tm, tn);
    } else if (allow_big == 3) {               // 16-deep K chunks: 19 KB LDS -> up to 8 WGs/CU
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((bgemm_kernel<1, 16, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st,
                           al, bl, ep, M, N, K, nb, tm, tn);
    } else {
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((bgemm_kernel<1, 32, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st,
                           al, bl, ep, M, N, K, nb, tm, tn);
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// 128x128 / 8-wave variant (bgemm_p2_kernel<128>); only instantiated where it is asked for
template <bool A_KC, bool B_KC, class AL, class BL, class EP>
static inline int bgemm128_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    hipLaunchKernelGGL((bgemm_p2_kernel<128, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(512), 0, st, al,
                       bl, ep, M, N, K, nb, tm, tn);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// 64x64 tile with the two-chunk prefetch of bgemm_p2_kernel
template <bool A_KC, bool B_KC, class AL, class BL, class EP>
static inline int bgemm64p2_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    hipLaunchKernelGGL((bgemm_p2_kernel<64, A_KC, B_KC, AL, BL, EP>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st, al, bl,
                       ep, M, N, K, nb, tm, tn);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// symmetric-result variant (M == N, 64x64x32 tiles): see SYM above
template <bool A_KC, bool B_KC, class AL, class BL, class EP>
static inline int bgemm_launch_sym(const AL& al, const BL& bl, const EP& ep, int M, int K, int nb, hipStream_t st) {
    if (M <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 63) / 64;
    hipLaunchKernelGGL((bgemm_kernel<1, 32, A_KC, B_KC, AL, BL, EP, true>), dim3(xcd_grid(nb, tm * (tm + 1) / 2)),
                       dim3(256), 0, st, al, bl, ep, M, M, K, nb, tm, tm);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

static inline LdPlain make_plain(const float* p, long long bs, int ld, int R, int C) {
    LdPlain l;
    l.p = p; l.bs = bs; l.ld = ld; l.R = R; l.C = C;
    l.vec = (aligned16(p) && (ld % 4 == 0) && (bs % 4 == 0)) ? 1 : 0;
    return l;
}

static inline EpAffine make_affine(float* c, long long bs, int ld, float alpha, const float* bscale, float beta,
                                   float diag, int trans = 0) {
    EpAffine e;
    e.c = c; e.bs = bs; e.ld = ld; e.alpha = alpha; e.bscale = bscale; e.beta = beta; e.diag = diag; e.trans = trans;
    return e;
}

}  // namespace hk
