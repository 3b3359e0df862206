// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32 fma chain, 157 TF/s peak) with pluggable operand loaders and epilogues.
//
//   C[b] (M x N) = epilogue( sum_k A[b](i,k) * B[b](k,j) )
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves, 2x2, one 32x32 MFMA accumulator each), K in chunks of 32,
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
#include <type_traits>

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

// Branch-free variants for operands whose requested tiles lie INSIDE the matrix (CIN: C % 64 == 0, K = C).  A load
// behind a bounds branch costs more than the branch: the compiler cannot count how many loads a later, conditional
// request has put in flight, so every wait on an earlier request becomes s_waitcnt vmcnt(0) - with the loaders above the
// DEEP kernel waited for chunk c + 2 before it stored chunk c + 1 (one chunk of real prefetch, ~3 us per 32-deep chunk).
// LdPlainV: every float4 is inside the operand and 16-byte aligned.  LdPlainC: rows inside, columns clamped to C - 1 and
// zero-filled afterwards - four unconditional 4-byte loads (C x 49 maps: pitch 49).
struct LdPlainV {
    const float* p;
    long long bs;
    int ld;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        return *reinterpret_cast<const float4*>(p + (long long)b * bs + (long long)r * ld + c);
    }
};
struct LdPlainC {
    struct Raw { float v[4]; int n; };               // n: how many of the four columns exist
    const float* p;
    long long bs;
    int ld, C;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ Raw ldraw(int b, int r, int c) const {
        const float* q = p + (long long)b * bs + (long long)r * ld;
        Raw x;
#pragma unroll
        for (int t = 0; t < 4; ++t) x.v[t] = q[c + t < C ? c + t : C - 1];
        x.n = C - c;
        return x;
    }
    __device__ __forceinline__ float4 cook(const Raw& x) const {
        return make_float4(x.n > 0 ? x.v[0] : 0.f, x.n > 1 ? x.v[1] : 0.f, x.n > 2 ? x.v[2] : 0.f, x.n > 3 ? x.v[3] : 0.f);
    }
};

// LdPlainN: as LdPlainC without the zero fill - for an operand whose clamped columns are OUTPUT columns beyond N (B given
// as K x N): what lands there is never stored, and with no arithmetic on the loaded registers nothing waits for them before
// the tile goes to LDS.
struct LdPlainN {
    const float* p;
    long long bs;
    int ld, C;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        const float* q = p + (long long)b * bs + (long long)r * ld;
        const int l = C - 1;
        return make_float4(q[c < l ? c : l], q[c + 1 < l ? c + 1 : l], q[c + 2 < l ? c + 2 : l], q[c + 3 < l ? c + 3 : l]);
    }
};

// A loader may split ld4 into the memory request (`Raw ldraw(b, r, c)`, kept in registers while in flight) and the
// arithmetic on it (`float4 cook(Raw)`, applied when the tile goes to LDS) - otherwise arithmetic inside ld4 waits for
// the data where it was requested.  Loaders without a `Raw` member are used as they are.
template <class L, class = void>
struct LdTraits {
    using Raw = float4;
    static __device__ __forceinline__ Raw ld(L& l, int b, int r, int c) { return l.ld4(b, r, c); }
    static __device__ __forceinline__ float4 cook(const L&, const Raw& x) { return x; }
};
template <class L>
struct LdTraits<L, std::void_t<typename L::Raw>> {
    using Raw = typename L::Raw;
    static __device__ __forceinline__ Raw ld(L& l, int b, int r, int c) { return l.ldraw(b, r, c); }
    static __device__ __forceinline__ float4 cook(const L& l, const Raw& x) { return l.cook(x); }
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
    // the same for four consecutive columns j .. j + 3 of row i (not transposed; 16-byte aligned: see bgemm_kernel VEPI)
    __device__ __forceinline__ void store4(int b, int i, int j, float4 v) const {
        const float s = bscale ? alpha * bscale[b] : alpha;
        float* q = c + (long long)b * bs + (long long)i * ld + j;
        float r[4] = {s * v.x, s * v.y, s * v.z, s * v.w};
        if (i >= j && i < j + 4) r[i - j] += diag;
        if (beta != 0.f) {
            const float4 o = *reinterpret_cast<const float4*>(q);
            r[0] += beta * o.x; r[1] += beta * o.y; r[2] += beta * o.z; r[3] += beta * o.w;
        }
        *reinterpret_cast<float4*>(q) = make_float4(r[0], r[1], r[2], r[3]);
    }
};

// ---------------------------------------------------------------- kernel
// 64x64 workgroup tile (4 waves x one 32x32 accumulator split into two k-chains, 37 KB LDS, 4 workgroups / CU), K in
// chunks of 32.  Round 1 also carried a 128x128 tile, 64- and 16-deep chunks, a two-chunk prefetch, a symmetric-tile
// mode and bf16-split products here; all measured slower or equal on the MI355X (BENCH_r01: 417-563 us against
// 436 us on the Newton-Schulz forward) and were removed.  The Newton-Schulz chain now has its own grouped kernel
// (hk_nsmm.h); this one serves the ragged / small shapes: classifier slabs, CIN, n-pairs, generic fallbacks.
// DEEP = true (long-K, memory-streaming shapes: CIN's C x C by C x 49 products, K = 2048): chunk c + 2 is requested
// while chunk c is computed (two register sets, loop unrolled by two, no branch around a load or an LDS store: the
// chunk index is clamped instead, a redundant last load / store is harmless) - with one chunk of prefetch a workgroup
// waited ~3.5 us per 32-deep chunk and 2.5 resident workgroups per CU did not cover it; measured on CIN's four
// products: 242 -> 200, 440 -> 407, 379 -> 352, 376 -> 355 us.  What was left (~1.8 TB/s) was NOT the access pattern: the
// loaders' bounds branches made every wait a vmcnt(0) (see LdPlainV above) - with branch-free loaders the same kernel
// streams |W - w W'| X at 4.1 TB/s (344 -> 163 us).  The loop walks chunk PAIRS without a branch inside (the launcher takes
// the plain kernel for an odd chunk count).
// VEPI = true (EP = EpAffine, not transposed, ld % 4 == 0, 16-byte aligned result, N % 4 == 0 - large results with a
// short K: CIN's dW = dY X^T writes 335 MB for 49-deep products): the tile leaves through LDS as 16-byte stores of
// whole 256-byte rows instead of one 4-byte store per accumulator register (128-byte runs): same values, same order of
// operations per element.
template <bool A_KC, bool B_KC, class AL, class BL, class EP, int DEEP = 0, bool VEPI = false>
__global__ __launch_bounds__(256) void bgemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int nb,
                                                     int tilesM, int tilesN) {
    constexpr int T = 1, BKT = 32;                       // (64-deep chunks with DEEP: 1.3-1.7x SLOWER on the CIN products)
    constexpr bool SYM = false;
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

    using TA = LdTraits<AL>;
    using TB = LdTraits<BL>;
    constexpr int NSET = DEEP ? DEEP : 1;                // register sets of requested chunks (DEEP: 2 or 4)
    static_assert(DEEP == 0 || DEEP == 2 || DEEP == 4, "an even number of sets: the LDS stage of a chunk is then compile-time");
    typename TA::Raw ras[NSET][NLA];
    typename TB::Raw rbs[NSET][NLB];
    auto& ra = ras[0];
    auto& rb = rbs[0];

// (macro-local names carry a trailing underscore: the argument expressions mention the caller's `c`)
#define HK_GLOAD_TO(RA_, RB_, k0)                                                                      \
    do {                                                                                               \
        _Pragma("unroll") for (int u = 0; u < NLA; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / A4, c_ = f_ % A4;                                       \
            RA_[u] = A_KC ? TA::ld(al, b, m0 + r_, (k0) + 4 * c_) : TA::ld(al, b, (k0) + r_, m0 + 4 * c_); \
        }                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NLB; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / B4, c_ = f_ % B4;                                       \
            RB_[u] = B_KC ? TB::ld(bl, b, n0 + r_, (k0) + 4 * c_) : TB::ld(bl, b, (k0) + r_, n0 + 4 * c_); \
        }                                                                                              \
    } while (0)
#define HK_GLOAD(k0) HK_GLOAD_TO(ra, rb, k0)

#define HK_SSTORE_FROM(RA_, RB_, buf)                                                                  \
    do {                                                                                               \
        float* As_ = lds + (buf) * (SA + SB);                                                          \
        float* Bs_ = As_ + SA;                                                                         \
        _Pragma("unroll") for (int u = 0; u < NLA; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / A4, c_ = f_ % A4;                                       \
            *reinterpret_cast<float4*>(&As_[r_ * PA + 4 * c_]) = TA::cook(al, RA_[u]);                   \
        }                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NLB; ++u) {                                              \
            const int f_ = tid + 256 * u, r_ = f_ / B4, c_ = f_ % B4;                                       \
            *reinterpret_cast<float4*>(&Bs_[r_ * PB + 4 * c_]) = TB::cook(bl, RB_[u]);                   \
        }                                                                                              \
    } while (0)
#define HK_SSTORE(buf) HK_SSTORE_FROM(ra, rb, buf)

    const int nk = (K + BK - 1) / BK;
    HK_GLOAD(0);
#pragma unroll
    for (int h = 1; h < NSET; ++h) HK_GLOAD_TO(ras[h], rbs[h], (h < nk ? h : nk - 1) * BK);   // chunks 0 .. NSET - 1 in flight
    HK_SSTORE(0);
    __syncthreads();

    // DEEP: the loop counts groups of NSET chunks; chunk c = NSET cc + half lives in register set `half` (compile-time)
    const int nloop = DEEP ? nk / NSET : nk;             // (DEEP: the launcher guarantees nk % NSET == 0)
    for (int cc = 0; cc < nloop; ++cc)
#pragma unroll
    for (int half = 0; half < NSET; ++half) {
        const int c = DEEP ? NSET * cc + half : cc;
        const int cur = c & 1;
        if (DEEP) {                                      // chunk c + NSET into the set whose content (chunk c) is in LDS
            const int cn = c + NSET < nk ? c + NSET : nk - 1;
            HK_GLOAD_TO(ras[half], rbs[half], cn * BK);
            __builtin_amdgcn_sched_barrier(0);           // the requests stay here, ahead of the chunk's MFMAs
        } else {
            if (c + 1 < nk) HK_GLOAD((c + 1) * BK);
        }
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
        if (DEEP) {                                      // chunk c + 1 (requested a chunk ago) to the other stage; behind the
            HK_SSTORE_FROM(ras[(half + 1) % NSET], rbs[(half + 1) % NSET], cur ^ 1);   // last chunk: a stage nobody reads again
        } else {
            if (c + 1 < nk) HK_SSTORE(cur ^ 1);
        }
        __syncthreads();
    }
#undef HK_GLOAD
#undef HK_SSTORE
#undef HK_GLOAD_TO
#undef HK_SSTORE_FROM

    if (T == 1) acc[0][0] += acc2;
    if constexpr (VEPI) {
        // (the last chunk's barrier has passed: the stages are free) tile image [64][68], then 16 float4 per row
        static_assert(T == 1 && 64 * 68 <= 2 * (SA + SB), "the output image fits the stages");
        float* img = lds;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            img[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + wn * 32 + l31] = acc[0][0][r];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = tid + 256 * u, row = f >> 4, c4 = (f & 15) << 2;
            const int ii = m0 + row, jj = n0 + c4;
            if (ii < M && jj < N) ep.store4(b, ii, jj, *reinterpret_cast<const float4*>(&img[row * 68 + c4]));
        }
        al.finish(b, tm, tn, tilesM, lds);
        return;
    }
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

template <bool A_KC, bool B_KC, int DEEP = 0, bool VEPI = false, class AL, class BL, class EP>
static inline int bgemm_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    if (DEEP && ((K + 31) / 32) % (DEEP ? DEEP : 1) != 0)   // the pipeline walks groups of DEEP chunks, no branch inside
        hipLaunchKernelGGL((bgemm_kernel<A_KC, B_KC, AL, BL, EP, 0, VEPI>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st, al, bl,
                           ep, M, N, K, nb, tm, tn);
    else
        hipLaunchKernelGGL((bgemm_kernel<A_KC, B_KC, AL, BL, EP, DEEP, VEPI>), dim3(xcd_grid(nb, tm * tn)), dim3(256), 0, st, al, bl,
                           ep, M, N, K, nb, tm, tn);
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
