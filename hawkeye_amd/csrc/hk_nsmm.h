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
    int bscale_fn;         // s_b = bscale[b] (0), sqrt(bscale[b]) (1: MPNCOV.py:161,181), 1 / bscale[b] (2: A = a / trace)
    float* norm_out;       // FIRST launches only: trace(a[b]) is written here
    // LAST launches only (the backward's final product, MPNCOV.py:194-197): next to C = D the workgroup adds up its
    // tile's share of  sum(g o out)  and  sum(D^T o a)  -> rpart[b][tile][0 / 1]
    const float* rg;
    const float* rout;
    const float* ra;
    float* rpart;
    float* tv;             // nullable: the result's upper triangle, row-major packed [b][d (d + 1) / 2] (Triuvec,
                           // MPNCOV.py:205-230), written next to C by the chain's last forward product
};

struct NsGroup {
    NsProb p[4];
    int np;
};

// TN = columns of the workgroup tile (128 or 64; rows are always 128).  EDGE = true: any d / alignment (guarded scalar
// loads and stores); false: d % 128 == 0, 16-byte aligned operands.
// SYM = true (needs EDGE = false, no E1 / E2 / C2): every result of the launch is a SYMMETRIC matrix (all forward products
// of the chain are: the iterates are polynomials in the symmetric input) - only the workgroup tiles that touch the
// 128 x 128 blocks on or above the diagonal are launched (3 of 4 at d = 256) and a tile right of its diagonal block
// also writes its transpose (through the LDS the main loop has finished with: 16-byte stores along the rows).
// FIRST = true: the chain's first launch, straight from the un-normalised input a (term 0 = a a, E1 = a):
//     tr = trace(a[b])  (every workgroup recomputes it: d diagonal elements, fixed-order block sum; written to norm_out)
//     C  = Y_0 = A (3I - A) / 2 = (1.5 / tr) a - (0.5 / tr^2) a a          C2 = Z_0 = (3I - A) / 2 = 1.5 I - (0.5 / tr) a
// (MPNCOV.py:144-154) - the separate pass that normalised a and formed Z_0 (50 MB through HBM and a fork of the helper
// queue behind it: 12.6 + 6.7 us of a 240 us chain, profiles/r3_ns_launch_timeline.csv) is gone.
// LAST = true: the backward's final product; its epilogue also reduces the two trace terms of MPNCOV.py:175,197 over the
// tile (fixed order: per thread in register order, then the block tree) - the separate reduction pass over D, g, out
// and a (16.5 us, profiles/r3_ns_launch_timeline.csv) is gone - and, on the aligned path, writes the tile TRANSPOSED
// and scaled: C = D^T / trace, which is the gradient up to its diagonal term (MPNCOV.py:195-201; the transposing pass
// of ns_bwd_final_kernel shrinks to a diagonal update).
template <int TN>
struct NsTileCfg {
    static constexpr int TM = 128, BK = 32;
    static constexpr int PA = BK + 4;           // A chunk [128][32] k-contiguous: pitch 36 (pitch/4 odd: ds_read_b128 conflict-free)
    static constexpr int PB = TN + 4;           // B chunk [32][TN] n-contiguous
    static constexpr int SA = TM * PA, SB = BK * PB;
    static constexpr int LDS_FLOATS = 2 * (SA + SB);
    static constexpr int RT = TM / TN;          // column tiles per 128-column block
};

// origin of workgroup tile `tile` of a problem: all tilesM x tilesN tiles row-major, or (SYM) only the tiles on or right
// of the diagonal 128 x 128 blocks - block row I owns the column tiles RT * I .. tilesN - 1
template <int TN, bool SYM>
__device__ __forceinline__ void ns_tile_origin(int tile, int tilesN, int& m0, int& n0) {
    constexpr int RT = NsTileCfg<TN>::RT;
    if (SYM) {
        int I = 0, rem = tile;
        while (rem >= tilesN - RT * I) { rem -= tilesN - RT * I; ++I; }
        m0 = I * 128;
        n0 = (RT * I + rem) * TN;
    } else {
        m0 = (tile / tilesN) * 128;
        n0 = (tile % tilesN) * TN;
    }
}
template <int TN, bool SYM>
__host__ __device__ __forceinline__ int ns_tile_count(int tilesM, int tilesN) {
    return SYM ? tilesM * tilesN - NsTileCfg<TN>::RT * (tilesM * (tilesM - 1) / 2) : tilesM * tilesN;
}

// One workgroup tile (rows m0.., columns n0..) of problem P for sample b; `lds` = NsTileCfg<TN>::LDS_FLOATS floats.
template <int TN, bool EDGE, bool SYM, bool FIRST, bool LAST>
__device__ __forceinline__ void nsmm_tile(const NsProb& P, int d, int b, int tile, int tiles, int m0, int n0, float* lds) {
    static_assert(!(SYM && EDGE), "the symmetric schedule is for d % 128 == 0");
    static_assert(!(LAST && (SYM || FIRST)), "the final product of the backward is a general matrix");
    constexpr int TM = 128, BK = 32;
    constexpr int NJ = TN / 64;                 // 32-column MFMA tiles per wave (wave tile = 64 x TN/2)
    constexpr int PA = NsTileCfg<TN>::PA, PB = NsTileCfg<TN>::PB;
    constexpr int SA = NsTileCfg<TN>::SA, SB = NsTileCfg<TN>::SB;
    constexpr int NLB = BK * TN / 4 / 256;      // float4 of B per thread per chunk (4 or 2)
    constexpr int B4 = TN / 4;                  // float4 per staged B row
    const bool mirror = SYM && n0 >= m0 + TM;   // (uniform) the tile lies right of its diagonal block
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
    // chunks per term, rounded up to an even count (the loop below handles chunks in pairs; an extra chunk exists only
    // for d % 64 in 1..32 on the guarded path, where it stages zeros)
    const int nkt = (((d + BK - 1) / BK) + 1) & ~1;
    const int nk = P.nt * nkt;

    // running per-thread operand pointers of the term being staged (advanced by one chunk per gload)
    const float *pa = nullptr, *pb = nullptr;
    float sg_term = 1.f;                                      // sign of the term being staged
    int k0 = 0;                                               // k offset of the next chunk to stage (EDGE guards)
    auto set_term = [&](int ti) {
        NsTerm T;                                             // by value (a dynamic index would spill the table to scratch,
        if (ti == 0) T = P.t[0];                              //  a reference into a register-resident P likewise)
        else if (ti == 1) T = P.t[1];
        else T = P.t[2];
        pa = T.A + (long long)b * T.sa + (long long)(m0 + ar) * d + ac;
        pb = T.B + (long long)b * T.sb + (long long)br * d + n0 + bc;
        sg_term = T.sign;
        k0 = 0;
    };
    // Staging registers: TWO sets of named vectors (an indexed array here is not promoted to registers by the compiler
    // once the loads and the LDS stores sit in different conditional blocks: it ends up in scratch).  Chunk c + 2 is
    // requested while chunk c is computed, so a load has two chunk-times (2 x 2048-4096 matrix-pipe cycles) to arrive -
    // and the prologue has chunks 0 AND 1 in flight at once: one memory latency per launch instead of two.
    struct Regs {
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        float sg;                                             // sign of the term the chunk belongs to
    };
    Regs R0, R1;
    R0.a0 = R0.a1 = R0.a2 = R0.a3 = R0.b0 = R0.b1 = R0.b2 = R0.b3 = make_float4(0.f, 0.f, 0.f, 0.f);
    R0.sg = 1.f;
    R1 = R0;
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
    int ti = 0, kc = 0;                                       // term being staged / chunks of it already requested
    set_term(0);
    // request the next chunk (in chunk order) into register set R
#define HK_NS_GLOAD(R)                                                     \
    do {                                                                   \
        if (kc == nkt) { set_term(++ti); kc = 0; }                         \
        R.a0 = lda(0); R.a1 = lda(1); R.a2 = lda(2); R.a3 = lda(3);        \
        R.b0 = ldb(0); R.b1 = ldb(1);                                      \
        if (NLB == 4) { R.b2 = ldb(2); R.b3 = ldb(3); }                    \
        pa += BK;                                                          \
        pb += (long long)BK * d;                                           \
        k0 += BK;                                                          \
        R.sg = sg_term;                                                    \
        ++kc;                                                              \
    } while (0)
    // registers -> LDS stage `buf`; the term's sign is applied here (exact), i.e. AFTER the chunk's MFMAs were issued:
    // touching the loaded values any earlier would park the wave on the global loads at the top of the chunk
    auto sta = [&](float* As, int u, float4 v, float sg) {
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
        *reinterpret_cast<float4*>(&As[(ar + 32 * u) * PA + ac]) = v;
    };
#define HK_NS_SSTORE(R, buf)                                                               \
    do {                                                                                   \
        float* As_ = lds + (buf) * (SA + SB);                                              \
        float* Bs_ = As_ + SA;                                                             \
        sta(As_, 0, R.a0, R.sg); sta(As_, 1, R.a1, R.sg); sta(As_, 2, R.a2, R.sg); sta(As_, 3, R.a3, R.sg); \
        *reinterpret_cast<float4*>(&Bs_[(br + BRS * 0) * PB + bc]) = R.b0;                 \
        *reinterpret_cast<float4*>(&Bs_[(br + BRS * 1) * PB + bc]) = R.b1;                 \
        if (NLB == 4) {                                                                    \
            *reinterpret_cast<float4*>(&Bs_[(br + BRS * 2) * PB + bc]) = R.b2;             \
            *reinterpret_cast<float4*>(&Bs_[(br + BRS * 3) * PB + bc]) = R.b3;             \
        }                                                                                  \
    } while (0)
#define HK_NS_FRAG(s_, A_, B_)                                                                        \
        do {                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
                const float4 v_ = *reinterpret_cast<const float4*>(&As[i * 32 * PA + 8 * (s_)]);          \
                A_[i][0] = v_.x; A_[i][1] = v_.y; A_[i][2] = v_.z; A_[i][3] = v_.w;                       \
            }                                                                                             \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) B_[j][t] = Bs[(8 * (s_) + t) * PB + 32 * j]; \
        } while (0)
    // One chunk.  Order (fenced with sched_barrier so that it survives the scheduler): fragments of step 0, then the
    // global loads of chunk c + 2 into RLOAD (its previous content, chunk c, is in LDS), then per 8-deep step:
    // fragments of step s + 1, 16 (8) MFMAs of step s.  Chunk c + 1 (in RSTORE, requested a chunk ago) goes to the
    // other LDS stage before the LAST step's MFMAs, which cover the store; one barrier per chunk.
    // LOAD_ / STORE_ are compile-time: a branch around the global loads would make the compiler's s_waitcnt vmcnt
    // accounting conservative at the join (it then waits for the NEWEST loads before every LDS store - the two-deep
    // prefetch is gone and the wait lands at the end of the chunk).
#define HK_NS_CHUNK(c_, RLOAD, RSTORE, LOAD_, STORE_)                                                             \
    do {                                                                                                          \
        const int cur = (c_) & 1;                                                                                 \
        const float* As = lds + cur * (SA + SB) + (wm * 64 + l31) * PA + 4 * lh;                                  \
        const float* Bs = lds + cur * (SA + SB) + SA + (4 * lh) * PB + wn * (TN / 2) + l31;                       \
        float a[2][4], bb[NJ][4];                                                                                 \
        HK_NS_FRAG(0, a, bb);                                                                                     \
        if (LOAD_) HK_NS_GLOAD(RLOAD);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        _Pragma("unroll") for (int s = 0; s < BK / 8; ++s) {                                                      \
            float an[2][4], bn[NJ][4];                                                                            \
            if (s + 1 < BK / 8) {                                                                                 \
                HK_NS_FRAG(s + 1, an, bn);                                                                        \
            } else if (STORE_) {                                                                                  \
                HK_NS_SSTORE(RSTORE, cur ^ 1);                                                                    \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j)       /* operands swapped: see the epilogue */ \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[j][t], a[i][t], acc[i][j], 0, 0, 0);  \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (s + 1 < BK / 8) {                                                                                 \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) a[i][t] = an[i][t];                             \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j) bb[j][t] = bn[j][t];                           \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        __syncthreads();                                                                                          \
    } while (0)

    HK_NS_GLOAD(R0);                                          // chunk 0
    HK_NS_GLOAD(R1);                                          // chunk 1 (nk >= 2): both in flight before any wait
    float tr_inv = 1.f;
    if (FIRST) {                                              // behind the operand requests: its latency is theirs
        __shared__ float red[4];
        const float* ab = P.t[0].A + (long long)b * P.t[0].sa;
        float sd = 0.f;
        for (int i = tid; i < d; i += 256) sd += ab[(long long)i * d + i];
        const float na = block_sum<4>(sd, red);               // the summation order of ns_scale_kernel: the same bits
        if (tile == 0 && tid == 0 && P.norm_out) P.norm_out[b] = na;
        tr_inv = 1.0f / na;
    }
    HK_NS_SSTORE(R0, 0);
    __syncthreads();
    int c = 0;
    for (; c + 2 < nk; c += 2) {                              // steady state
        HK_NS_CHUNK(c, R0, R1, true, true);                   // even chunk: R0 <- chunk c + 2, R1 (chunk c + 1) -> LDS
        HK_NS_CHUNK(c + 1, R1, R0, true, true);               // odd chunk : R1 <- chunk c + 3, R0 (chunk c + 2) -> LDS
    }
    HK_NS_CHUNK(c, R0, R1, false, true);                      // last pair: nothing left to request
    HK_NS_CHUNK(c + 1, R1, R0, false, false);
#undef HK_NS_CHUNK
#undef HK_NS_FRAG
#undef HK_NS_GLOAD
#undef HK_NS_SSTORE

    // epilogue.  The MFMAs above were issued with the operands SWAPPED (B fragment first), so an accumulator holds the
    // transposed tile: lane & 31 = row m of C, (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = column n.  The four registers
    // of a group are four consecutive columns of one row: one 16-byte load / store per group instead of four 4-byte
    // ones (16 store instructions per wave for a 64 x 64 sub-tile instead of 64 - the launch's store burst is
    // issue-bound: every workgroup of a launch reaches its epilogue at the same time).
    // Explicit fmaf everywhere: both tile widths must round identically (a contraction left to the compiler differs
    // between instantiations).
    // every field of the problem descriptor is read ONCE into a scalar here (P lives in the kernarg segment behind a
    // dynamic index: each mention is a scalar load, and a mention inside a per-element condition becomes a branch)
    float sb_ = 1.0f;
    if (P.bscale) {
        const float v_ = P.bscale[b];
        sb_ = P.bscale_fn == 1 ? sqrtf(v_) : (P.bscale_fn == 2 ? 1.0f / v_ : v_);
    }
    float rs0 = 0.f, rs1 = 0.f;                               // LAST: this thread's share of the two sums
    const float* rgb = LAST ? P.rg + (long long)b * d * d : nullptr;
    const float* rob = LAST ? P.rout + (long long)b * d * d : nullptr;
    const float* rab = LAST ? P.ra + (long long)b * d * d : nullptr;
    const float al = FIRST ? -0.5f * tr_inv * tr_inv : P.alpha * sb_, diag = FIRST ? 0.f : P.diag;
    const float alpha2 = FIRST ? -0.5f * tr_inv : P.alpha2, diag2 = FIRST ? 1.5f : P.diag2;
    float* Cb = P.C + (long long)b * P.sc;
    const float* E1b = P.E1 ? P.E1 + (long long)b * P.se1 : nullptr;
    const float* E2b = P.E2 ? P.E2 + (long long)b * P.se2 : nullptr;
    float* C2b = P.C2 ? P.C2 + (long long)b * P.sc2 : nullptr;
    const bool has1 = E1b != nullptr, has2 = E2b != nullptr, has_c2 = C2b != nullptr;
    float* tvb = P.tv ? P.tv + (long long)b * ((long long)d * (d + 1) / 2) : nullptr;        // (uniform)
    const float e1 = FIRST ? 1.5f * tr_inv : (has1 ? (P.e1_scaled ? P.e1 * sb_ : P.e1) : 0.f);   // (a missing term: 0 * 0 added, exact)
    const float e2 = has2 ? P.e2 : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = m0 + wm * 64 + i * 32 + l31;
            const int cbase = n0 + wn * (TN / 2) + j * 32 + 4 * lh;
            const long long orow = (long long)row * d;
            float4 x1[4], x2[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) x1[gq] = x2[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has1) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = cbase + 8 * gq;
                    if (!EDGE) {
                        x1[gq] = *reinterpret_cast<const float4*>(E1b + orow + col);
                    } else if (row < d) {
                        float* u1 = reinterpret_cast<float*>(&x1[gq]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (col + t < d) u1[t] = E1b[orow + col + t];
                    }
                }
            }
            if (has2) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = cbase + 8 * gq;
                    if (!EDGE) {
                        x2[gq] = *reinterpret_cast<const float4*>(E2b + orow + col);
                    } else if (row < d) {
                        float* u2 = reinterpret_cast<float*>(&x2[gq]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (col + t < d) u2[t] = E2b[orow + col + t];
                    }
                }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int col = cbase + 8 * gq;
                float o1[4], o2[4];
                const float xs1[4] = {x1[gq].x, x1[gq].y, x1[gq].z, x1[gq].w};
                const float xs2[4] = {x2[gq].x, x2[gq].y, x2[gq].z, x2[gq].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float v = acc[i][j][4 * gq + t];
                    const bool dg = row == col + t;
                    o1[t] = fmaf(e2, xs2[t], fmaf(e1, xs1[t], fmaf(al, v, dg ? diag : 0.f)));
                    o2[t] = fmaf(alpha2, FIRST ? xs1[t] : v, dg ? diag2 : 0.f);
                }
                if (tvb && row < d) {           // Triuvec: element (row, c >= row) -> row d - row (row - 1) / 2 + c - row
                    float* tr = tvb + (long long)row * d - (long long)row * (row - 1) / 2 - row;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (col + t >= row && col + t < d) tr[col + t] = o1[t];
                }
                if (LAST) {                     // sum(g o out) and sum(D^T o a) = sum_ij D_ij a_ji: a read transposed -
                                                // lanes run along `row`, so a[(col + t) d + row] is a coalesced line
                    if (!EDGE) {
                        const float4 gv = *reinterpret_cast<const float4*>(rgb + orow + col);
                        const float4 ov = *reinterpret_cast<const float4*>(rob + orow + col);
                        rs0 = fmaf(gv.x, ov.x, rs0); rs0 = fmaf(gv.y, ov.y, rs0);
                        rs0 = fmaf(gv.z, ov.z, rs0); rs0 = fmaf(gv.w, ov.w, rs0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) rs1 = fmaf(o1[t], rab[(long long)(col + t) * d + row], rs1);
                    } else if (row < d) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (col + t < d) {
                                rs0 = fmaf(rgb[orow + col + t], rob[orow + col + t], rs0);
                                rs1 = fmaf(o1[t], rab[(long long)(col + t) * d + row], rs1);
                            }
                    }
                }
                if (SYM && mirror) {            // this wave's image [column][row], pitch 68: lanes along the rows
                    float* Tw = lds + wave * ((TN / 2) * 68);
#pragma unroll
                    for (int t = 0; t < 4; ++t) Tw[(j * 32 + 4 * lh + 8 * gq + t) * 68 + i * 32 + l31] = o1[t];
                }
                if (!EDGE && LAST) {            // transposed result through the wave's image (as the mirror below)
                    float* Tw = lds + wave * ((TN / 2) * 68);
#pragma unroll
                    for (int t = 0; t < 4; ++t) Tw[(j * 32 + 4 * lh + 8 * gq + t) * 68 + i * 32 + l31] = o1[t] * sb_;
                } else if (!EDGE) {
                    *reinterpret_cast<float4*>(Cb + orow + col) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                    if (has_c2) *reinterpret_cast<float4*>(C2b + orow + col) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                } else if (row < d) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (col + t < d) {
                            Cb[orow + col + t] = o1[t];
                            if (has_c2) C2b[orow + col + t] = o2[t];
                        }
                }
            }
        }
    if (LAST) {
        __shared__ float red[4];
        rs0 = block_sum<4>(rs0, red);
        rs1 = block_sum<4>(rs1, red);
        if (tid == 0) {
            float* pp = P.rpart + ((long long)b * tiles + tile) * 2;
            pp[0] = rs0;
            pp[1] = rs1;
        }
    }
    // C[n][m] = image[n][m]: the wave reads its image back four columns (= rows of the transposed tile) at a time
    static_assert(4 * (TN / 2) * 68 <= 2 * (SA + SB), "the four wave images fit the LDS of the main loop");
    float* Tw = lds + wave * ((TN / 2) * 68);
    const long long moff = (long long)(n0 + wn * (TN / 2)) * d + m0 + wm * 64;
    auto flush = [&](float* dst) {
        const int q = lane & 15, c4 = lane >> 4;
        HK_WAVE_SYNC();
#pragma unroll
        for (int c0 = 0; c0 < TN / 2; c0 += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&Tw[(c0 + c4) * 68 + 4 * q]);
            *reinterpret_cast<float4*>(dst + moff + (long long)(c0 + c4) * d + 4 * q) = v;
        }
        HK_WAVE_SYNC();
    };
    if (LAST && !EDGE) flush(Cb);
    if (SYM && mirror) {
        flush(Cb);
        if (FIRST) {                                        // Z_0's mirror: its tile again, from a (L2-hot), through the image
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int row = m0 + wm * 64 + i * 32 + l31;
                    const int cbase = n0 + wn * (TN / 2) + j * 32 + 4 * lh;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 xv = *reinterpret_cast<const float4*>(E1b + (long long)row * d + cbase + 8 * gq);
                        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t)       // (right of the diagonal block: no diagonal element here)
                            Tw[(j * 32 + 4 * lh + 8 * gq + t) * 68 + i * 32 + l31] = fmaf(alpha2, xs[t], 0.f);
                    }
                }
            flush(C2b);
        }
    }
}

template <int TN, bool EDGE, bool SYM = false, bool FIRST = false, bool LAST = false>
__global__ __launch_bounds__(256, 2) void nsmm_kernel(const NsGroup g, int d, int nb, int b0, int tilesM, int tilesN) {
    __shared__ __attribute__((aligned(16))) float lds[NsTileCfg<TN>::LDS_FLOATS];
    const int tiles = ns_tile_count<TN, SYM>(tilesM, tilesN);
    int b, t_;
    if (!xcd_map(blockIdx.x, nb, g.np * tiles, b, t_)) return;
    b += b0;                                    // this launch covers samples b0 .. b0 + nb - 1
    const int pi = t_ / tiles, tile = t_ % tiles;
    int m0, n0;
    ns_tile_origin<TN, SYM>(tile, tilesN, m0, n0);
    nsmm_tile<TN, EDGE, SYM, FIRST, LAST>(g.p[pi], d, b, tile, tiles, m0, n0, lds);
}

// host side ------------------------------------------------------------------------------------------------------
__host__ __device__ static inline NsTerm ns_term(const float* A, long long sa, const float* B, long long sb, float sign = 1.0f) {
    NsTerm t;
    t.A = A; t.B = B; t.sa = sa; t.sb = sb; t.sign = sign; t.pad_ = 0;
    return t;
}

__host__ __device__ static inline NsProb ns_prob(float* C, long long sc, float alpha, float diag, const float* bscale = nullptr) {
    NsProb p;
    for (int i = 0; i < 3; ++i) p.t[i] = ns_term(nullptr, 0, nullptr, 0);
    p.nt = 0;
    p.alpha = alpha; p.diag = diag; p.bscale = bscale;
    p.C = C; p.sc = sc;
    p.E1 = nullptr; p.se1 = 0; p.e1 = 0.f; p.e1_scaled = 0;
    p.E2 = nullptr; p.se2 = 0; p.e2 = 0.f;
    p.C2 = nullptr; p.sc2 = 0; p.alpha2 = 0.f; p.diag2 = 0.f;
    p.bscale_fn = 0; p.norm_out = nullptr;
    p.rg = p.rout = p.ra = nullptr; p.rpart = nullptr; p.tv = nullptr;
    return p;
}
static inline NsProb& operator+=(NsProb& p, const NsTerm& t) {
    p.t[p.nt++] = t;
    return p;
}

static inline bool ns_prob_aligned(const NsProb& p) {
    bool ok = aligned16(p.C) && p.sc % 4 == 0 && (!p.C2 || (aligned16(p.C2) && p.sc2 % 4 == 0)) &&
              (!p.E1 || (aligned16(p.E1) && p.se1 % 4 == 0)) && (!p.E2 || (aligned16(p.E2) && p.se2 % 4 == 0));
    for (int i = 0; i < p.nt; ++i)
        ok = ok && aligned16(p.t[i].A) && aligned16(p.t[i].B) && p.t[i].sa % 4 == 0 && p.t[i].sb % 4 == 0;
    return ok;
}

// tn: 0 = choose (128-wide tiles when that still gives two workgroups per CU, else 64-wide), 64 / 128 = forced
// sym: every result is a symmetric matrix (see the kernel); taken when the fast path applies and no problem has a second
// result or epilogue operands, otherwise the launch computes all tiles as usual
// first: the chain's first launch (FIRST instantiation: one problem with term 0 = (a, a), E1 = a, C = Y_0, C2 = Z_0)
// last: the backward's final product (LAST instantiation, 64-wide tiles: rpart holds ceil(d / 128) * ceil(d / 64) pairs
// per sample, whatever the batch size)
static inline int nsmm_last_tiles(int d) { return ((d + 127) / 128) * ((d + 63) / 64); }
static inline bool ns_prob_aligned(const NsProb& p);
// the aligned form of the LAST launch writes C = D^T / trace (else C = D): the caller picks C accordingly
static inline bool nsmm_last_transposes(const NsProb& p, int d) {
    return d % 128 == 0 && ns_prob_aligned(p) && aligned16(p.rg) && aligned16(p.rout);
}
static inline int nsmm_launch(const NsGroup& g, int d, int nb, hipStream_t st, int tn = 0, int b0 = 0, bool sym = false,
                              bool first = false, bool last = false) {
    if (g.np < 1 || g.np > 4 || d <= 0 || nb <= 0) return HK_ERR_BAD_ARG;
    bool fast = d % 128 == 0;
    for (int i = 0; i < g.np; ++i) {
        if (g.p[i].nt < 1 || g.p[i].nt > 3 || !g.p[i].C) return HK_ERR_BAD_ARG;
        fast = fast && ns_prob_aligned(g.p[i]);
    }
    const int tm = (d + 127) / 128;
    if (last) {
        if (g.np != 1 || !g.p[0].rg || !g.p[0].rout || !g.p[0].ra || !g.p[0].rpart) return HK_ERR_BAD_ARG;
        fast = nsmm_last_transposes(g.p[0], d);
        const int tnn64 = (d + 63) / 64;
        const dim3 gl(xcd_grid(nb, tm * tnn64));
        if (fast) hipLaunchKernelGGL((nsmm_kernel<64, false, false, false, true>), gl, dim3(256), 0, st, g, d, nb, b0, tm, tnn64);
        else hipLaunchKernelGGL((nsmm_kernel<64, true, false, false, true>), gl, dim3(256), 0, st, g, d, nb, b0, tm, tnn64);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    if (tn == 0) tn = tuning().ns_tn;
    // 64-wide tiles unless that would put more than 8 workgroups on every CU: measured at B = 64, d = 256 (ns_bench):
    // forced 64 -> 291 / 750 us (fwd / bwd), forced 128 -> 297 / 770, mixed (128 for the multi-problem launches) 291 / 768
    if (tn != 64 && tn != 128) tn = ((long long)g.np * tm * tm * 2 * nb > 2048) ? 128 : 64;
    const int tnn = (d + tn - 1) / tn;
    if (first && (g.np != 1 || g.p[0].nt != 1 || !g.p[0].E1 || !g.p[0].C2)) return HK_ERR_BAD_ARG;
    for (int i = 0; i < g.np; ++i) sym = sym && (first || (!g.p[i].C2 && !g.p[i].E1)) && !g.p[i].E2;
    if (sym && fast) {
        const int tiles = tm * tnn - (128 / tn) * (tm * (tm - 1) / 2);
        const dim3 gs(xcd_grid(nb, g.np * tiles));
        if (first) {
            if (tn == 128) hipLaunchKernelGGL((nsmm_kernel<128, false, true, true>), gs, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
            else hipLaunchKernelGGL((nsmm_kernel<64, false, true, true>), gs, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        } else {
            if (tn == 128) hipLaunchKernelGGL((nsmm_kernel<128, false, true>), gs, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
            else hipLaunchKernelGGL((nsmm_kernel<64, false, true>), gs, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        }
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    const dim3 grid(xcd_grid(nb, g.np * tm * tnn));
    if (first) {
        if (tn == 128) {
            if (fast) hipLaunchKernelGGL((nsmm_kernel<128, false, false, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
            else hipLaunchKernelGGL((nsmm_kernel<128, true, false, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        } else {
            if (fast) hipLaunchKernelGGL((nsmm_kernel<64, false, false, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
            else hipLaunchKernelGGL((nsmm_kernel<64, true, false, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        }
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    if (tn == 128) {
        if (fast) hipLaunchKernelGGL((nsmm_kernel<128, false>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        else hipLaunchKernelGGL((nsmm_kernel<128, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
    } else {
        if (fast) hipLaunchKernelGGL((nsmm_kernel<64, false>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
        else hipLaunchKernelGGL((nsmm_kernel<64, true>), grid, dim3(256), 0, st, g, d, nb, b0, tm, tnn);
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}


}  // namespace hk
