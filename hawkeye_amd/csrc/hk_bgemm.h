// Batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32 fma chain, 157 TF/s peak) with pluggable operand loaders and epilogues.
//
//   C[b] (M x N) = epilogue( sum_k A[b](i,k) * B[b](k,j) )
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves, 2x2, one 32x32
// MFMA accumulator = 16 VGPRs each), K in chunks of 32, register-staged
// global->LDS double buffer, one barrier per chunk; 37 KB LDS -> 4 WGs/CU.
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
template <bool A_KC, bool B_KC, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void bgemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int nb,
                                                     int tilesM, int tilesN) {
    constexpr int BM = 64, BN = 64, BK = 32;
    constexpr int PA = A_KC ? BK + 4 : BM + 4;
    constexpr int PB = B_KC ? BK + 4 : BN + 4;
    constexpr int SA = (A_KC ? BM : BK) * PA;
    constexpr int SB = (B_KC ? BN : BK) * PB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];

    int b, tile;
    if (!xcd_map(blockIdx.x, nb, tilesM * tilesN, b, tile)) return;
    const int tm = tile / tilesN, tn = tile % tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    al.begin(b, tm, tn);
    bl.begin(b, tm, tn);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    float4 ra[2], rb[2];
    // staging coordinates of this thread's two float4 per operand
    const int f0 = tid, f1 = tid + 256;
    const int arow0 = A_KC ? (f0 >> 3) : (f0 >> 4), ac0 = A_KC ? (f0 & 7) : (f0 & 15);
    const int arow1 = A_KC ? (f1 >> 3) : (f1 >> 4), ac1 = A_KC ? (f1 & 7) : (f1 & 15);
    const int brow0 = B_KC ? (f0 >> 3) : (f0 >> 4), bc0 = B_KC ? (f0 & 7) : (f0 & 15);
    const int brow1 = B_KC ? (f1 >> 3) : (f1 >> 4), bc1 = B_KC ? (f1 & 7) : (f1 & 15);

#define HK_GLOAD(k0)                                                                       \
    do {                                                                                   \
        ra[0] = A_KC ? al.ld4(b, m0 + arow0, (k0) + 4 * ac0) : al.ld4(b, (k0) + arow0, m0 + 4 * ac0); \
        ra[1] = A_KC ? al.ld4(b, m0 + arow1, (k0) + 4 * ac1) : al.ld4(b, (k0) + arow1, m0 + 4 * ac1); \
        rb[0] = B_KC ? bl.ld4(b, n0 + brow0, (k0) + 4 * bc0) : bl.ld4(b, (k0) + brow0, n0 + 4 * bc0); \
        rb[1] = B_KC ? bl.ld4(b, n0 + brow1, (k0) + 4 * bc1) : bl.ld4(b, (k0) + brow1, n0 + 4 * bc1); \
    } while (0)

#define HK_SSTORE(buf)                                                                     \
    do {                                                                                   \
        float* As_ = lds + (buf) * (SA + SB);                                              \
        float* Bs_ = As_ + SA;                                                             \
        *reinterpret_cast<float4*>(&As_[arow0 * PA + 4 * ac0]) = ra[0];                    \
        *reinterpret_cast<float4*>(&As_[arow1 * PA + 4 * ac1]) = ra[1];                    \
        *reinterpret_cast<float4*>(&Bs_[brow0 * PB + 4 * bc0]) = rb[0];                    \
        *reinterpret_cast<float4*>(&Bs_[brow1 * PB + 4 * bc1]) = rb[1];                    \
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
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float a[4], bb[4];
            if (A_KC) {
                const float4 v = *reinterpret_cast<const float4*>(&As[(wm * 32 + l31) * PA + 8 * s + 4 * lh]);
                a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = As[(8 * s + 4 * lh + t) * PA + wm * 32 + l31];
            }
            if (B_KC) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[(wn * 32 + l31) * PB + 8 * s + 4 * lh]);
                bb[0] = v.x; bb[1] = v.y; bb[2] = v.z; bb[3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bb[t] = Bs[(8 * s + 4 * lh + t) * PB + wn * 32 + l31];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bb[t], acc, 0, 0, 0);
        }
        if (c + 1 < nk) HK_SSTORE(cur ^ 1);
        __syncthreads();
    }
#undef HK_GLOAD
#undef HK_SSTORE

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int j = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < M && j < N) ep(b, i, j, acc[r]);
    }
    al.finish(b, tm, tn, tilesM, lds);
}

template <bool A_KC, bool B_KC, class AL, class BL, class EP>
static inline int bgemm_launch(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int nb,
                               hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    const int grid = xcd_grid(nb, tm * tn);
    hipLaunchKernelGGL((bgemm_kernel<A_KC, B_KC, AL, BL, EP>), dim3(grid), dim3(256), 0, st, al, bl, ep, M, N, K,
                       nb, tm, tn);
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
