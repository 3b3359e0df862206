// Grouped square products for the Newton-Schulz chain (Sqrtm.forward / Sqrtm.backward, model/methods/MPNCOV.py:137-202).
//
// Why a second GEMM kernel next to hk_bgemm.h: the chain is 12 (forward) / 38 (backward) products of d x d matrices per
// sample, d = 256, and round 1 ran it as 50 launches of a 64x64x32 tile at 0.35-0.38 of the fp32 MFMA peak - 45 % of
// the wave time parked at the chunk barrier, 16 FLOP per byte of L2 -> LDS traffic, a launch boundary every 35 us
// (profiles/r1c_sq_wait_counters.csv).  What this kernel changes:
//   * ONE launch runs a GROUP of up to four independent problems (the reference's schedule has them: Y ZY and ZY Z of
//     one forward iteration; Y Z, Z dldZ and Y dldY of one backward iteration, MPNCOV.py:156-159,183-193), so a launch
//     holds 512-768 workgroups instead of 256 and two workgroups share a CU: one's barrier / prologue / epilogue is
//     covered by the other's MFMAs;
//   * a problem is a SUM of up to three products accumulated in the same MFMA accumulators
//     (dldY' = .5 (dldY YZ - (Z dldZ) Z - ZY dldY) is one K = 3d product, not three GEMMs with read-modify-write
//     epilogues): 38 -> 9 GEMM launches in the backward, 12 -> 9 in the forward;
//   * 128 x 128 (or 128 x 64) output tile, 4 waves of 64 x 64 (64 x 32): 32 (21) FLOP per staged byte, 64 (32) MFMAs per
//     wave per 32-deep chunk between barriers, <= 256 VGPRs so that two workgroups fit a CU;
//   * the elementwise glue of the reference (3I - ., .5 ., * sqrt(tr), +-) is the epilogue
//         C  = alpha * s_b * acc + diag * I + e1 * E1 + e2 * E2          C2 = alpha2 * acc + diag2 * I   (optional)
//     written once; nothing is read-modify-written.
// Numerics: v_mfma_f32_32x32x2_f32 = exact fp32 fma chain; within an 8-wide k-step lanes 0-31 own k = 8s + t and lanes
// 32-63 own k = 8s + 4 + t (the same fixed permutation as hk_bgemm.h), terms are accumulated in order: deterministic.
// The sign of a term is applied to its A operand while staging (exact).
#pragma once
#include "hk_common.h"

namespace hk {

struct NsTerm {            // sign * A[b] * B[b], row-major d x d, batch strides in elements
    const float* A;
    const float* B;
    long long sa, sb;
    float sign;
    int pad_;
};

struct NsProb {
    NsTerm t[3];
    int nt;
    float alpha, diag;
    const float* bscale;   // nullable: per-sample scale s_b on alpha (and on e1 when e1_scaled)
    float* C;
    long long sc;
    const float* E1;       // nullable
    long long se1;
    float e1;
    int e1_scaled;
    const float* E2;       // nullable
    long long se2;
    float e2;
    float* C2;             // nullable second result of the same accumulator
    long long sc2;
    float alpha2, diag2;
};

struct NsGroup {
    NsProb p[4];
    int np;
};

// TN = columns of the workgroup tile (128 or 64; rows are always 128).  EDGE = true: any d / alignment (guarded scalar
// loads and stores); false: d % 128 == 0, 16-byte aligned operands.
template <int TN, bool EDGE>
__global__ __launch_bounds__(256, 2) void nsmm_kernel(const NsGroup g, int d, int nb, int tilesM, int tilesN) {
    constexpr int TM = 128, BK = 32;
    constexpr int NJ = TN / 64;                 // 32-column MFMA tiles per wave (wave tile = 64 x TN/2)
    constexpr int PA = BK + 4;                  // A chunk [128][32] k-contiguous: pitch 36 (pitch/4 odd: ds_read_b128 conflict-free)
    constexpr int PB = TN + 4;                  // B chunk [32][TN] n-contiguous
    constexpr int SA = TM * PA, SB = BK * PB;
    constexpr int NLB = BK * TN / 4 / 256;      // float4 of B per thread per chunk (4 or 2)
    constexpr int B4 = TN / 4;                  // float4 per staged B row
    __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];

    const int tiles = tilesM * tilesN;
    int b, t_;
    if (!xcd_map(blockIdx.x, nb, g.np * tiles, b, t_)) return;
    const int pi = t_ / tiles, tile = t_ % tiles;
    const NsProb& P = g.p[pi];
    const int m0 = (tile / tilesN) * TM, n0 = (tile % tilesN) * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ar = tid >> 3, ac = 4 * (tid & 7);             // A staging: row ar + 32 u, k offset ac
    const int br = tid / B4, bc = 4 * (tid % B4);             // B staging: k row br + BRS u, column bc
    constexpr int BRS = 256 / B4;
    const int nkt = (d + BK - 1) / BK;                        // chunks per term
    const int nk = P.nt * nkt;

    // running per-thread operand pointers of the term being staged (advanced by one chunk per gload)
    const float *pa = nullptr, *pb = nullptr;
    float sg_term = 1.f, sg_regs = 1.f;                       // sign of the term being staged / of the chunk in ra
    int k0 = 0;                                               // k offset of the next chunk to stage (EDGE guards)
    auto set_term = [&](int ti) {
        const NsTerm& T = ti == 0 ? P.t[0] : (ti == 1 ? P.t[1] : P.t[2]);   // (a dynamic index would spill the table to scratch)
        pa = T.A + (long long)b * T.sa + (long long)(m0 + ar) * d + ac;
        pb = T.B + (long long)b * T.sb + (long long)br * d + n0 + bc;
        sg_term = T.sign;
        k0 = 0;
    };
    // staging registers as named scalars (an indexed array here is not promoted to registers by the compiler once the
    // loads and the LDS stores sit in different conditional blocks: it ends up in scratch)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto lda = [&](int u) -> float4 {
        const float* q = pa + (long long)(32 * u) * d;
        if (!EDGE) return *reinterpret_cast<const float4*>(q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int row = m0 + ar + 32 * u, kk = k0 + ac;
        if (row < d) {
            if (kk < d) v.x = q[0];
            if (kk + 1 < d) v.y = q[1];
            if (kk + 2 < d) v.z = q[2];
            if (kk + 3 < d) v.w = q[3];
        }
        return v;
    };
    auto ldb = [&](int u) -> float4 {
        const float* q = pb + (long long)(BRS * u) * d;
        if (!EDGE) return *reinterpret_cast<const float4*>(q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int kr = k0 + br + BRS * u, col = n0 + bc;
        if (kr < d) {
            if (col < d) v.x = q[0];
            if (col + 1 < d) v.y = q[1];
            if (col + 2 < d) v.z = q[2];
            if (col + 3 < d) v.w = q[3];
        }
        return v;
    };
#define HK_NS_GLOAD()                                  \
    do {                                               \
        ra0 = lda(0); ra1 = lda(1); ra2 = lda(2); ra3 = lda(3); \
        rb0 = ldb(0); rb1 = ldb(1);                    \
        if (NLB == 4) { rb2 = ldb(2); rb3 = ldb(3); }  \
        pa += BK;                                      \
        pb += (long long)BK * d;                       \
        k0 += BK;                                      \
        sg_regs = sg_term;                             \
    } while (0)
    // registers -> LDS stage `buf`; the term's sign is applied here (exact), i.e. AFTER the chunk's MFMAs were issued:
    // touching the loaded values any earlier would park the wave on the global loads at the top of the chunk
    auto sta = [&](float* As, int u, float4 v) {
        v.x *= sg_regs; v.y *= sg_regs; v.z *= sg_regs; v.w *= sg_regs;
        *reinterpret_cast<float4*>(&As[(ar + 32 * u) * PA + ac]) = v;
    };
#define HK_NS_SSTORE(buf)                                                                  \
    do {                                                                                   \
        float* As_ = lds + (buf) * (SA + SB);                                              \
        float* Bs_ = As_ + SA;                                                             \
        sta(As_, 0, ra0); sta(As_, 1, ra1); sta(As_, 2, ra2); sta(As_, 3, ra3);            \
        *reinterpret_cast<float4*>(&Bs_[(br + BRS * 0) * PB + bc]) = rb0;                  \
        *reinterpret_cast<float4*>(&Bs_[(br + BRS * 1) * PB + bc]) = rb1;                  \
        if (NLB == 4) {                                                                    \
            *reinterpret_cast<float4*>(&Bs_[(br + BRS * 2) * PB + bc]) = rb2;              \
            *reinterpret_cast<float4*>(&Bs_[(br + BRS * 3) * PB + bc]) = rb3;              \
        }                                                                                  \
    } while (0)

    int ti = 0, kc = 1;                                       // term / chunks of it already requested
    set_term(0);
    HK_NS_GLOAD();
    HK_NS_SSTORE(0);
    __syncthreads();

    for (int c = 0; c < nk; ++c) {
        const int cur = c & 1;
        const float* As = lds + cur * (SA + SB) + (wm * 64 + l31) * PA + 4 * lh;
        const float* Bs = lds + cur * (SA + SB) + SA + (4 * lh) * PB + wn * (TN / 2) + l31;
        float a[2][4], bb[NJ][4];
#define HK_NS_FRAG(s_, A_, B_)                                                                        \
        do {                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
                const float4 v_ = *reinterpret_cast<const float4*>(&As[i * 32 * PA + 8 * (s_)]);          \
                A_[i][0] = v_.x; A_[i][1] = v_.y; A_[i][2] = v_.z; A_[i][3] = v_.w;                       \
            }                                                                                             \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) B_[j][t] = Bs[(8 * (s_) + t) * PB + 32 * j]; \
        } while (0)
        // order inside a chunk (fenced with sched_barrier so that it survives the scheduler): fragments of step 0, then
        // the global loads of the NEXT chunk (their address arithmetic runs in the shadow of the LDS latency), then per
        // 8-deep step: fragments of step s + 1, 16 (8) MFMAs of step s.  The loaded chunk goes to the other LDS stage
        // before the LAST step's MFMAs, which cover the store; one barrier per chunk.
        HK_NS_FRAG(0, a, bb);
        const bool more = c + 1 < nk;
        if (more) {
            if (kc == nkt) { set_term(++ti); kc = 0; }
            HK_NS_GLOAD();
            ++kc;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float an[2][4], bn[NJ][4];
            if (s + 1 < BK / 8) {
                HK_NS_FRAG(s + 1, an, bn);
            } else if (more) {
                HK_NS_SSTORE(cur ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], bb[j][t], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < BK / 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i][t] = an[i][t];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bb[j][t] = bn[j][t];
                }
            }
        }
#undef HK_NS_FRAG
        __syncthreads();
    }
#undef HK_NS_GLOAD
#undef HK_NS_SSTORE

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const float sb_ = P.bscale ? P.bscale[b] : 1.0f;
    const float al = P.alpha * sb_;
    const float e1 = P.e1_scaled ? P.e1 * sb_ : P.e1;
    float* Cb = P.C + (long long)b * P.sc;
    const float* E1b = P.E1 ? P.E1 + (long long)b * P.se1 : nullptr;
    const float* E2b = P.E2 ? P.E2 + (long long)b * P.se2 : nullptr;
    float* C2b = P.C2 ? P.C2 + (long long)b * P.sc2 : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn * (TN / 2) + j * 32 + l31;
            const int rbase = m0 + wm * 64 + i * 32 + 4 * lh;
            float x1[16], x2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const long long o = (long long)row * d + col;
                const bool ok = !EDGE || (row < d && col < d);
                x1[r] = (E1b && ok) ? E1b[o] : 0.f;
                x2[r] = (E2b && ok) ? E2b[o] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const long long o = (long long)row * d + col;
                if (EDGE && !(row < d && col < d)) continue;
                const float v = acc[i][j][r];
                float out = al * v;
                if (row == col) out += P.diag;
                if (E1b) out += e1 * x1[r];
                if (E2b) out += P.e2 * x2[r];
                Cb[o] = out;
                if (C2b) C2b[o] = P.alpha2 * v + (row == col ? P.diag2 : 0.f);
            }
        }
}

// host side ------------------------------------------------------------------------------------------------------
static inline NsTerm ns_term(const float* A, long long sa, const float* B, long long sb, float sign = 1.0f) {
    NsTerm t;
    t.A = A; t.B = B; t.sa = sa; t.sb = sb; t.sign = sign; t.pad_ = 0;
    return t;
}

static inline NsProb ns_prob(float* C, long long sc, float alpha, float diag, const float* bscale = nullptr) {
    NsProb p;
    for (int i = 0; i < 3; ++i) p.t[i] = ns_term(nullptr, 0, nullptr, 0);
    p.nt = 0;
    p.alpha = alpha; p.diag = diag; p.bscale = bscale;
    p.C = C; p.sc = sc;
    p.E1 = nullptr; p.se1 = 0; p.e1 = 0.f; p.e1_scaled = 0;
    p.E2 = nullptr; p.se2 = 0; p.e2 = 0.f;
    p.C2 = nullptr; p.sc2 = 0; p.alpha2 = 0.f; p.diag2 = 0.f;
    return p;
}
static inline NsProb& operator+=(NsProb& p, const NsTerm& t) {
    p.t[p.nt++] = t;
    return p;
}

static inline bool ns_prob_aligned(const NsProb& p) {
    bool ok = aligned16(p.C) && p.sc % 4 == 0;
    for (int i = 0; i < p.nt; ++i)
        ok = ok && aligned16(p.t[i].A) && aligned16(p.t[i].B) && p.t[i].sa % 4 == 0 && p.t[i].sb % 4 == 0;
    return ok;
}

// tn: 0 = choose (128-wide tiles when that still gives two workgroups per CU, else 64-wide), 64 / 128 = forced
static inline int nsmm_launch(const NsGroup& g, int d, int nb, hipStream_t st, int tn = 0) {
    if (g.np < 1 || g.np > 4 || d <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    bool fast = d % 128 == 0;
    for (int i = 0; i < g.np; ++i) {
        if (g.p[i].nt < 1 || g.p[i].nt > 3 || !g.p[i].C) return HK_ERR_BAD_ARG;
        fast = fast && ns_prob_aligned(g.p[i]);
    }
    const int tm = (d + 127) / 128;
    if (tn == 0) tn = tuning().ns_tn;
    if (tn != 64 && tn != 128) tn = ((long long)g.np * tm * tm * nb >= 512) ? 128 : 64;
    const int tnn = (d + tn - 1) / tn;
    const dim3 grid(xcd_grid(nb, g.np * tm * tnn));
    if (tn == 128) {
        if (fast) hipLaunchKernelGGL((nsmm_kernel<128, false>), grid, dim3(256), 0, st, g, d, nb, tm, tnn);
        else hipLaunchKernelGGL((nsmm_kernel<128, true>), grid, dim3(256), 0, st, g, d, nb, tm, tnn);
    } else {
        if (fast) hipLaunchKernelGGL((nsmm_kernel<64, false>), grid, dim3(256), 0, st, g, d, nb, tm, tnn);
        else hipLaunchKernelGGL((nsmm_kernel<64, true>), grid, dim3(256), 0, st, g, d, nb, tm, tnn);
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
