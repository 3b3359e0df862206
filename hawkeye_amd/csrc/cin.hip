// CIN channel interaction: the Gram primitive at C = 2048, HW = 49 followed by a row softmax and a
// second product, plus the contrastive variant that mixes the interaction matrices of two batch halves.
// replaces ChannelInteractionModule.forward, model/methods/CIN.py:24-60 (the bmm / softmax / abs / bmm parts; the 3x3
// convolution, the residual and the 1-output fc stay on PyTorch-ROCm).
//
//   SCI:  W = softmax_rows(-X X^T / HW) ; Y = W X                          hk_cin_sci_fwd / hk_cin_sci_bwd
//   CCI:  Wc[b] = | W[b] - w_b W[partner(b)] | ; Yc = Wc X                 hk_cin_cci_fwd / hk_cin_cci_bwd
//         partner(b) = (b + B/2) mod B  (CIN.py:45-51: the two batch halves are contrast pairs)
// For C % 64 == 0 and 7x7 / 8x8 / 6x6 maps (every CIN backbone) the row-owned work is three two-pass / one-pass kernels
// on the same tile scheme (64 rows of one sample per workgroup, four waves = row half x column half, K = HW Gram tiles
// recomputed on the matrix pipe instead of stored): cin_sci_flash_kernel (forward), cin_sci_bwd_flash_kernel (dW, softmax
// backward and dG X), cin_cci_dw_flash_kernel (the gradient reaching W from the contrastive branch); the C x C by C x HW
// products that remain (W^T dY, dG^T X, |W - w W'| X and its transpose) stream their big operand once through the generic
// tile with branch-free loaders (hk_bgemm.h: DEEP, LdPlainV / LdPlainN; LdAbsDiffV forms |W - w W'| on the way to LDS -
// it is never stored).  14x14 / 12x12 / 10x10 maps (a 448^2 input) have their own forward - the scores materialised by the Gram
// panel kernel of bcnn_fast.hip, row statistics, and cin_ax_kernel, which applies the softmax on the way into the second
// product - and the same pipeline runs the backward's W^T dY and (dG + dG^T) X (see cin_ax_kernel).  Other shapes take the chains on the generic tile (bounds-checked loaders, row softmax and its
// backward as one workgroup per row, fixed reduction order); knob bcnn_generic = 1 forces them (A/B, tests).
#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// bcnn_fast.hip: g = alpha X X^T on the Gram panel kernel (every tile of the symmetric matrix computed once, both halves written)
int gram_fast_scaled(const float* x, float alpha, float* g, int B, int C, int HW, hipStream_t st);

// in place: row <- softmax(row), n columns
__global__ __launch_bounds__(256) void cin_softmax_rows_kernel(float* __restrict__ w, int n) {
    __shared__ float red[4];
    float* p = w + (long long)blockIdx.x * n;
    float m = -3.402823466e38f;
    for (int c = threadIdx.x; c < n; c += 256) m = fmaxf(m, p[c]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) {
        const float e = expf(p[c] - m);
        p[c] = e;
        s += e;
    }
    s = block_sum<4>(s, red);
    for (int c = threadIdx.x; c < n; c += 256) p[c] = p[c] / s;
}

// in place on dW: dG = -dS = -W (.) (dW - <dW, W>_row)        (S = -G, softmax backward)
__global__ __launch_bounds__(256) void cin_softmax_bwd_rows_kernel(const float* __restrict__ w, float* __restrict__ dw,
                                                                  int n) {
    __shared__ float red[4];
    const float* p = w + (long long)blockIdx.x * n;
    float* q = dw + (long long)blockIdx.x * n;
    float t = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) t += p[c] * q[c];
    t = block_sum<4>(t, red);
    for (int c = threadIdx.x; c < n; c += 256) q[c] = -p[c] * (q[c] - t);
}

// |W[b] - w_b W[partner]| formed on the fly; (r, c) index the C x C matrix in memory order
struct LdAbsDiff {
    const float* p;      // W [B][C][C]
    const float* wt;     // w [B]
    int C, B;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < C) {
            const int pb = (b + B / 2) % B;
            const float wb = wt[b];
            const float* a = p + ((long long)b * C + r) * C;
            const float* o = p + ((long long)pb * C + r) * C;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (c + t < C) v[t] = fabsf(a[c + t] - wb * o[c + t]);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// The same operand for C % 64 == 0 and a 16-byte aligned W: two unconditional 16-byte requests (hk_bgemm.h, LdPlainV), the
// arithmetic applied when the tile goes to LDS.
struct LdAbsDiffV {
    struct Raw { float4 a, o; };
    const float* p;
    const float* wt;
    int C, B;
    float wb;
    __device__ __forceinline__ void begin(int b, int, int) { wb = wt[b]; }
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ Raw ldraw(int b, int r, int c) const {
        const int pb = (b + B / 2) % B;
        Raw x;
        x.a = *reinterpret_cast<const float4*>(p + ((long long)b * C + r) * C + c);
        x.o = *reinterpret_cast<const float4*>(p + ((long long)pb * C + r) * C + c);
        return x;
    }
    __device__ __forceinline__ float4 cook(const Raw& x) const {
        return make_float4(fabsf(x.a.x - wb * x.o.x), fabsf(x.a.y - wb * x.o.y), fabsf(x.a.z - wb * x.o.z),
                           fabsf(x.a.w - wb * x.o.w));
    }
};

// C x C by C x HW products on tiles that need no bounds branch (every CIN backbone: C = 2048)
static inline bool cin_inside(int C, const void* w) {     // (knob bcnn_generic = 1 keeps the loaders with bounds checks: A/B)
    return C % 64 == 0 && aligned16(w) && tuning().bcnn_generic != 1;
}
static inline LdPlainV cin_mat(const float* w, int C) {
    LdPlainV l;
    l.p = w; l.bs = (long long)C * C; l.ld = C;
    return l;
}
static inline LdPlainN cin_cols(const float* x, int C, int HW) {          // a C x HW map as the K x N operand
    LdPlainN l;
    l.p = x; l.bs = (long long)C * HW; l.ld = HW; l.C = HW;
    return l;
}
static inline LdPlainC cin_map(const float* x, int C, int HW) {
    LdPlainC l;
    l.p = x; l.bs = (long long)C * HW; l.ld = HW; l.C = HW;
    return l;
}

// dW[b] (+)= sign(D_b) (.) dWc[b] - w_pb sign(D_pb) (.) dWc[pb],  D_b = W[b] - w_b W[pb]   (sign(0) = 0: torch's abs)
// dwpart[b][blk] = - sum over this block's elements of sign(D_b) dWc[b] W[pb]
__global__ __launch_bounds__(256) void cin_cci_dw_kernel(const float* __restrict__ W, const float* __restrict__ wt,
                                                        const float* __restrict__ dWc, float* __restrict__ dW,
                                                        float* __restrict__ dwpart, int C, int B, int nblk) {
    __shared__ float red[4];
    const int b = blockIdx.y, pb = (b + B / 2) % B;
    const long long n = (long long)C * C;
    const float wb = wt[b], wp = wt[pb];
    const float* Wb = W + b * n;
    const float* Wp = W + pb * n;
    const float* Gb = dWc + b * n;
    const float* Gp = dWc + pb * n;
    float acc = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)nblk * 256) {
        const float a = Wb[e], o = Wp[e];
        const float db = a - wb * o, dp = o - wp * a;
        const float sb = (db > 0.f) ? 1.f : ((db < 0.f) ? -1.f : 0.f);
        const float sp = (dp > 0.f) ? 1.f : ((dp < 0.f) ? -1.f : 0.f);
        dW[b * n + e] = sb * Gb[e] - wp * sp * Gp[e];
        acc += sb * Gb[e] * o;
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) dwpart[(long long)b * nblk + blockIdx.x] = -acc;
}

__global__ __launch_bounds__(64) void cin_cci_dw_reduce_kernel(const float* __restrict__ dwpart, float* __restrict__ dwt,
                                                              int nblk) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nblk; ++i) s += dwpart[(long long)b * nblk + i];
        dwt[b] = s;
    }
}

constexpr int CIN_DW_BLOCKS = 64;

// A block of 64 rows x HW floats (contiguous in memory) on its way into LDS in two halves: blk_request issues all of a
// thread's 16-byte loads at once (index-clamped: never behind a branch), blk_commit writes them to the stage - placed
// BEHIND the column block's MFMAs, so that the loads have the whole tile to arrive.  (Rounds 3-4 staged these blocks with
// `for (f = tid; f < BLK / 4; f += 256) dst[f] = src[f]` at the TOP of the column block, which the compiler kept as a
// loop of load - s_waitcnt vmcnt(0) - ds_write: three to four memory round trips in a row before the first MFMA of every
// block, draining the W / E prefetch with them.)
template <int BLK>
struct BlkRegs {
    static constexpr int N = (BLK / 4 + 255) / 256;
    f32x4 v[N];
};
template <int BLK>
__device__ __forceinline__ void blk_request(const float* src, BlkRegs<BLK>& r, int tid) {
#pragma unroll
    for (int u = 0; u < BlkRegs<BLK>::N; ++u) {
        const int f = tid + 256 * u;
        r.v[u] = reinterpret_cast<const f32x4*>(src)[f < BLK / 4 ? f : BLK / 4 - 1];
    }
    __builtin_amdgcn_sched_barrier(0);                   // (left to itself the scheduler sinks the loads to just above their use)
}
template <int BLK>
__device__ __forceinline__ void blk_commit(float* dst, const BlkRegs<BLK>& r, int tid) {
#pragma unroll
    for (int u = 0; u < BlkRegs<BLK>::N; ++u) {
        // (threads past the end write the block's last piece once more - the piece blk_request gave them: the same bytes to the
        //  same place.  A store under `if (f < BLK / 4)` invites the compiler to sink the load into that branch - load, wait,
        //  store, one memory round trip behind the block's MFMAs, with every other wave waiting at the barrier)
        const int f = tid + 256 * u;
        reinterpret_cast<f32x4*>(dst)[f < BLK / 4 ? f : BLK / 4 - 1] = r.v[u];
    }
}
constexpr int CIN_SETS = 2;            // chunks of a streamed C x C operand requested ahead per workgroup (hk_bgemm.h, DEEP; 4 measured no faster: 174 / 590 / 1022 us against 168 / 580 / 1003 - the products are then paced by the matrix pipe, 64-column tiles for 49 columns)

// ---------------------------------------------------------------------------------------------------------------------
// SCI forward in ONE kernel (round 3).  The chain above - Gram -> row softmax -> W X on the generic tile - writes S, reads
// and rewrites it as W and reads W again: 1.3 GB for a result of 335 MB (B = 20, C = 2048), 706 us against 610 us for
// rocBLAS bmm + softmax + bmm.  Here a workgroup owns 64 rows i of one sample and walks the 64-row column blocks j TWICE:
//   pass 1  S_ij = -(x_i . x_j) / HW on the matrix pipe (K = HW = 49: the Gram is cheap to recompute), running row max
//           m_i and row sum l_i = sum_j exp(S_ij - m_i) (rescaled when the max moves) - nothing is written;
//   pass 2  S again, P = exp(S - m_i) / l_i - the FINAL softmax - is written to W once (16-byte stores) and, still in the
//           accumulator registers, is the A operand of the second product Y_i += P X_j.
// The MFMAs of S are issued with the operands swapped (x_j first), so an accumulator holds the transposed tile: lane = row
// i, registers = 16 columns j - row max / sum are per-lane loops plus one exchange between the lane halves, the stores
// are four consecutive columns per register group, and a register is directly the A fragment (i = lane, k = j) of the
// second product: lanes 0-31 hold columns 8 g + t, lanes 32-63 columns 8 g + 4 + t - a pair (j, j + 4) per MFMA step.
// W is written once and never read; X (401 KB per sample) streams from L2.  LDS: x_i block + two x_j blocks, rows as
// they lie in memory (pitch HW: odd for 7 x 7 maps, conflict-free 4-byte fragment reads).
// exp = v_exp_f32 on (S - m) log2 e (arguments <= 0): ~1e-6 relative, inside the 1e-4 budget.
// FOUR waves per workgroup: wave = (row half, column half of every block).  B x C / 64 = 640 workgroups on 256 CUs are
// 2.5 per CU whatever the workgroup size; as two waves of 32 rows x 64 columns a CU with three of them had two waves on
// two of its SIMDs (makespan 2 wave-times for 1.25 of work), as four waves of 32 x 32 it has three half-size waves on
// every SIMD (1.5): 415 -> 297 us.  The two column halves of a row meet twice through LDS: (m, l) after pass 1 (combined
// in a fixed order) and the partial Y tiles at the end.
template <int HW>
__global__ __launch_bounds__(256, 3) void cin_sci_flash_kernel(const float* __restrict__ x, float* __restrict__ w,
                                                               float* __restrict__ y, int C, int B) {
    constexpr int RB = 64;                               // rows per workgroup and per column block
    constexpr int BLK = RB * HW;                         // floats of one 64-row block of X (contiguous in memory)
    constexpr int KS = (HW + 1) / 2;                     // MFMA k-steps of the Gram (two k per step)
    constexpr int NT2 = (HW + 31) / 32;                  // 32-column tiles of Y
    static_assert(BLK % 4 == 0 && NT2 <= 2, "64 x HW block as float4; maps up to 8 x 8");
    static_assert(2 * NT2 * 16 * 64 <= 3 * BLK, "the partial Y tiles of two waves fit the block stages");
    __shared__ __attribute__((aligned(16))) float lds[3 * BLK + 32];
    __shared__ float comb[2][2][32][2];                  // [column half][row half][row]: (m, l) of pass 1
    float* sI = lds;
    float* sJ = lds + BLK;                               // two stages

    const int nrb = C / RB;
    int b, I;
    if (!xcd_map(blockIdx.x, B, nrb, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rw = wave & 1, cw = wave >> 1;             // this wave: rows 32 rw .. + 31, columns 32 cw .. + 31 of every block
    const int l31 = lane & 31, lh = lane >> 5;
    const float* xb = x + (long long)b * C * HW;
    const float inv_hw = 1.0f / (float)HW;
    constexpr float LOG2E = 1.4426950408889634f;

    BlkRegs<BLK> nx;                                     // the next column block of X on its way in
    auto load_blk = [&](int blk, float* dst) {           // 64 rows of X, at once (prologue)
        BlkRegs<BLK> r;
        blk_request<BLK>(xb + (long long)blk * BLK, r, tid);
        blk_commit<BLK>(dst, r, tid);
    };
    // block J + 1 (the last step: block J again, into the idle stage - requests are never behind a branch)
    auto request_next = [&](int J) { blk_request<BLK>(xb + (long long)(J + 1 < nrb ? J + 1 : J) * BLK, nx, tid); };
    auto commit_next = [&](int J) { blk_commit<BLK>(sJ + ((J + 1) & 1) * BLK, nx, tid); };
    if (tid < 32) lds[3 * BLK + tid] = 0.f;              // (read past the last block: the last k-step of an odd HW, columns >= HW of Y's last tile)
    load_blk(I, sI);
    load_blk(0, sJ);
    __syncthreads();

    // S tile of this wave's 32 rows x 32 columns of column block `cur` (one 32x32 accumulator, transposed: lane = row)
    const float* ai = sI + (rw * 32 + l31) * HW + lh;                         // x_i[k = 2 s + lh]
    auto gram = [&](const float* sj, f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bj = sj + (32 * cw + l31) * HW + lh;                    // x_j[j = 32 cw + l31][k = 2 s + lh]
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bool tail = (HW & 1) && s_ == KS - 1;                      // k = HW - 1 alone: the upper lane half adds 0
            const float av = (tail && lh) ? 0.f : ai[2 * s_];
            const float bv = (tail && lh) ? 0.f : bj[2 * s_];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
        }
    };

    // ---- pass 1: row max and row sum over this wave's columns
    float m = -3.402823466e38f, l = 0.f;
    for (int J = 0; J < nrb; ++J) {
        const float* sj = sJ + (J & 1) * BLK;
        request_next(J);
        f32x16 acc;
        gram(sj, acc);
        float tm = m;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = -(acc[r] * inv_hw); tm = fmaxf(tm, acc[r]); }
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                              // both halves of a row agree on the running max
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f((acc[r] - tm) * LOG2E);
        l = l * __builtin_amdgcn_exp2f((m - tm) * LOG2E) + ps;
        m = tm;
        commit_next(J);                                                      // (the other stage: free since the last barrier)
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);                                              // the two lane halves hold disjoint columns
    if (lh == 0) { comb[cw][rw][l31][0] = m; comb[cw][rw][l31][1] = l; }
    load_blk(0, sJ);
    __syncthreads();
    {   // the two column halves of a row, in a fixed order
        const float m0 = comb[0][rw][l31][0], l0 = comb[0][rw][l31][1], m1 = comb[1][rw][l31][0], l1 = comb[1][rw][l31][1];
        m = fmaxf(m0, m1);
        l = l0 * __builtin_amdgcn_exp2f((m0 - m) * LOG2E) + l1 * __builtin_amdgcn_exp2f((m1 - m) * LOG2E);
    }
    const float rl = 1.0f / l;

    // ---- pass 2: W = exp(S - m) / l, written once; Y += W X_j (this wave: its 32 columns of every block)
    f32x16 yacc[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[n][r] = 0.f;
    float* wrow = w + ((long long)b * C + I * RB + rw * 32 + l31) * C + 32 * cw;   // this lane's row of W, this wave's columns
    for (int J = 0; J < nrb; ++J) {
        const float* sj = sJ + (J & 1) * BLK;
        request_next(J);
        f32x16 acc;
        gram(sj, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_exp2f((-(acc[r] * inv_hw) - m) * LOG2E) * rl;
#pragma unroll
        for (int g = 0; g < 4; ++g)                                          // columns 32 cw + 8 g + 4 lh .. + 3
            *reinterpret_cast<f32x4*>(wrow + J * RB + 8 * g + 4 * lh) =
                (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        // Y_i += P X_j: A = P registers (i = lane & 31, k = column), B = x_j[column][n = lane & 31 (+ 32)]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * cw + (r & 3) + 8 * (r >> 2) + 4 * lh;         // this lane half's column of register r
#pragma unroll
            for (int n = 0; n < NT2; ++n) {
                const int col = 32 * n + l31;
                const float bv = sj[j * HW + col];       // (columns >= HW: whatever follows in LDS - unconditional reads keep the
                                                         //  loop one basic block; those columns of the tile are never stored)
                yacc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], bv, yacc[n], 0, 0, 0);
            }
        }
        commit_next(J);
        __syncthreads();
    }
    // the partial Y of the upper column half goes through LDS (the stages are free: everybody passed the last barrier)
    float* ybuf = lds + rw * (NT2 * 16 * 64);
    if (cw == 1) {
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ybuf[(n * 16 + r) * 64 + lane] = yacc[n][r];
    }
    __syncthreads();
    if (cw == 1) return;
    // yacc[n]: standard layout - lane & 31 = column n, registers = rows (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32
    float* yb = y + ((long long)b * C + I * RB + rw * 32) * HW;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int col = 32 * n + l31;
        if (col < HW) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                yb[((r & 3) + 8 * (r >> 2) + 4 * lh) * HW + col] = yacc[n][r] + ybuf[(n * 16 + r) * 64 + lane];
        }
    }
}

// SCI backward, the row-owned part in ONE kernel (same structure as cin_sci_flash_kernel; rows i of one sample per
// workgroup, four waves = row half x column half):
//   dW_ij = dY_i . X_j (+ E_ij, the gradient that reaches W from the contrastive branch)   recomputed per tile, never stored
//   pass 1   t_i = sum_j W_ij dW_ij
//   pass 2   dG_ij = -W_ij (dW_ij - t_i)  written once (over E), and dx_i += (dG X)_i / HW from the accumulator registers
// instead of dW = dY X^T (335 MB written, E read), the row softmax backward (670 MB read, 335 MB written) and the product
// dG X (335 MB read): the chain had two 335 MB results, each at the ~1.5 TB/s such a write gets here.  W and E are read
// twice (16-byte requests issued ahead of the tile's MFMAs).  dx must hold W^T dY already; dG^T X is added afterwards by
// the streamed product.
template <int HW, bool EXTRA>
__global__ __launch_bounds__(256, EXTRA ? 2 : 3) void cin_sci_bwd_flash_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ dy, float* dg,
                                                                   float* __restrict__ dx, int C, int B) {
    constexpr int RB = 64;
    constexpr int BLK = RB * HW;
    constexpr int KS = (HW + 1) / 2;
    constexpr int NT2 = (HW + 31) / 32;
    static_assert(BLK % 4 == 0 && NT2 <= 2, "64 x HW block as float4; maps up to 8 x 8");
    static_assert(2 * NT2 * 16 * 64 <= 3 * BLK, "the partial dx tiles of two waves fit the block stages");
    __shared__ __attribute__((aligned(16))) float lds[3 * BLK + 32];
    __shared__ float comb[2][2][32];
    float* sI = lds;
    float* sJ = lds + BLK;

    const int nrb = C / RB;
    int b, I;
    if (!xcd_map(blockIdx.x, B, nrb, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rw = wave & 1, cw = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* xb = x + (long long)b * C * HW;
    const float inv_hw = 1.0f / (float)HW;
    BlkRegs<BLK> nx;                                     // the next column block of X on its way in
    auto load_blk = [&](const float* src_, float* dst) {  // a block at once (prologue, between the passes)
        BlkRegs<BLK> r;
        blk_request<BLK>(src_, r, tid);
        blk_commit<BLK>(dst, r, tid);
    };
    if (tid < 32) lds[3 * BLK + tid] = 0.f;
    load_blk(dy + ((long long)b * C + I * RB) * HW, sI);
    load_blk(xb, sJ);
    __syncthreads();

    const float* ai = sI + (rw * 32 + l31) * HW + lh;                         // dY_i[k = 2 s + lh]
    auto gram = [&](const float* sj, f32x16& acc) {                           // dW tile, transposed: lane = row i, registers = columns
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bj = sj + (32 * cw + l31) * HW + lh;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bool tail = (HW & 1) && s_ == KS - 1;
            const float av = (tail && lh) ? 0.f : ai[2 * s_];
            const float bv = (tail && lh) ? 0.f : bj[2 * s_];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
        }
    };
    const long long rowoff = ((long long)b * C + I * RB + rw * 32 + l31) * C + 32 * cw + 4 * lh;
    const float* wrow = w + rowoff;                                           // this lane's row, this wave's columns
    float* grow = dg + rowoff;
    auto ldrow = [&](const float* base, int J, f32x4 (&v)[4]) {               // columns 8 g + 4 lh .. + 3 of block J
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(base + J * RB + 8 * g);
    };

    // W / E of block J + 1 are requested while block J is computed (two register sets, the loop walks block pairs; the
    // request of the X block comes first - its wait would otherwise wait for these too - and is never behind a branch)
    auto next_x = [&](int J) {
        const int Jn = J + 1 < nrb ? J + 1 : J;                              // (last block: a redundant copy into the idle stage)
        blk_request<BLK>(xb + (long long)Jn * BLK, nx, tid);
    };
    auto commit_x = [&](int J) { blk_commit<BLK>(sJ + ((J + 1) & 1) * BLK, nx, tid); };   // behind the block's MFMAs
    auto next_we = [&](int J, f32x4 (&wn)[4], f32x4 (&en)[4]) {
        const int Jn = J + 1 < nrb ? J + 1 : J;
        ldrow(wrow, Jn, wn);
        if (EXTRA) ldrow(grow, Jn, en);
    };

    // ---- pass 1: t_i = sum_j W_ij dW_ij
    float t = 0.f;
    f32x4 wa[4], ea[4], wb[4], eb[4];
    auto pass1 = [&](int J, f32x4 (&wv)[4], f32x4 (&ev)[4], f32x4 (&wn)[4], f32x4 (&en)[4]) {
        const float* sj = sJ + (J & 1) * BLK;
        next_x(J);
        next_we(J, wn, en);
        f32x16 acc;
        gram(sj, acc);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) t += wv[g][k] * (EXTRA ? acc[4 * g + k] + ev[g][k] : acc[4 * g + k]);
        commit_x(J);
        __syncthreads();
    };
    ldrow(wrow, 0, wa);
    if (EXTRA) ldrow(grow, 0, ea);
    {
        int J = 0;
        for (; J + 1 < nrb; J += 2) { pass1(J, wa, ea, wb, eb); pass1(J + 1, wb, eb, wa, ea); }
        if (J < nrb) pass1(J, wa, ea, wb, eb);
    }
    t += __shfl_xor(t, 32, 64);
    if (lh == 0) comb[cw][rw][l31] = t;
    load_blk(xb, sJ);
    __syncthreads();
    t = comb[0][rw][l31] + comb[1][rw][l31];

    // ---- pass 2: dG written once, dx_i += dG X / HW
    f32x16 yacc[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[n][r] = 0.f;
    auto pass2 = [&](int J, f32x4 (&wv)[4], f32x4 (&ev)[4], f32x4 (&wn)[4], f32x4 (&en)[4]) {
        const float* sj = sJ + (J & 1) * BLK;
        next_x(J);
        next_we(J, wn, en);                                                  // (E of block J + 1: other columns than the dG stored below)
        f32x16 acc;
        gram(sj, acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = EXTRA ? acc[4 * g + k] + ev[g][k] : acc[4 * g + k];
                acc[4 * g + k] = -wv[g][k] * (d - t);
            }
            *reinterpret_cast<f32x4*>(grow + J * RB + 8 * g) = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * cw + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int n = 0; n < NT2; ++n) {
                const int col = 32 * n + l31;
                const float bv = sj[j * HW + col];       // (columns >= HW: whatever follows in LDS - unconditional reads keep the
                                                         //  loop one basic block; those columns of the tile are never stored)
                yacc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], bv, yacc[n], 0, 0, 0);
            }
        }
        commit_x(J);
        __syncthreads();
    };
    ldrow(wrow, 0, wa);
    if (EXTRA) ldrow(grow, 0, ea);
    {
        int J = 0;
        for (; J + 1 < nrb; J += 2) { pass2(J, wa, ea, wb, eb); pass2(J + 1, wb, eb, wa, ea); }
        if (J < nrb) pass2(J, wa, ea, wb, eb);
    }
    float* ybuf = lds + rw * (NT2 * 16 * 64);
    if (cw == 1) {
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ybuf[(n * 16 + r) * 64 + lane] = yacc[n][r];
    }
    __syncthreads();
    if (cw == 1) return;
    float* xo = dx + ((long long)b * C + I * RB + rw * 32) * HW;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int col = 32 * n + l31;
        if (col < HW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* q = xo + ((r & 3) + 8 * (r >> 2) + 4 * lh) * HW + col;
                *q += (yacc[n][r] + ybuf[(n * 16 + r) * 64 + lane]) * inv_hw;
            }
        }
    }
}

static int cin_sci_bwd_flash(const float* x, const float* w, const float* dy, float* dg, int has_extra, float* dx, int B,
                             int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(w) || !aligned16(dy) || !aligned16(dg)) return HK_ERR_UNSUPPORTED;
    const dim3 grid(xcd_grid(B, C / 64));
#define HK_CIN_BF(HW_)                                                                                                 \
    do {                                                                                                                \
        if (has_extra) hipLaunchKernelGGL((cin_sci_bwd_flash_kernel<HW_, true>), grid, dim3(256), 0, st, x, w, dy, dg, dx, C, B);  \
        else hipLaunchKernelGGL((cin_sci_bwd_flash_kernel<HW_, false>), grid, dim3(256), 0, st, x, w, dy, dg, dx, C, B);           \
    } while (0)
    switch (HW) {
        case 49: HK_CIN_BF(49); break;
        case 64: HK_CIN_BF(64); break;
        case 36: HK_CIN_BF(36); break;
        default: return HK_ERR_UNSUPPORTED;
    }
#undef HK_CIN_BF
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// CCI backward, the gradient that reaches W: dW[b] = sign(D_b) (.) dWc[b] - w_pb sign(D_pb) (.) dWc[pb] with dWc = dY X^T.
// The chain wrote dWc (335 MB at the ~1.5 TB/s such a result gets) and read it back in cin_cci_dw_kernel; here the two
// tiles dWc[b]_ij = dY[b]_i . X[b]_j and dWc[pb]_ij are recomputed on the matrix pipe (25 MFMAs each, K = HW) in the
// transposed accumulator layout of cin_sci_flash_kernel, combined with the W[b] / W[pb] tiles (16-byte requests issued
// ahead of the block's barriers) and written once.  dwpart[b][I] = - sum over the workgroup's rows of sign(D_b) dWc[b] W[pb].
template <int HW>
__global__ __launch_bounds__(256, 2) void cin_cci_dw_flash_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                  const float* __restrict__ wt, const float* __restrict__ dy,
                                                                  float* __restrict__ dW, float* __restrict__ dwpart, int C,
                                                                  int B) {
    constexpr int RB = 64;
    constexpr int BLK = RB * HW;
    constexpr int KS = (HW + 1) / 2;
    static_assert(BLK % 4 == 0, "64 x HW block as float4");
    // dY[b]_I, dY[pb]_I, then the pair X[b]_J, X[pb]_J - in TWO stages where six blocks leave room for two workgroups per CU
    // (7 x 7 and 6 x 6 maps: 75 / 55 KB): the next pair is requested before this one's tiles and written to the other stage
    // behind them, one barrier per column block; 8 x 8 maps keep one stage (request, two barriers around the copy)
    constexpr bool DB = 6 * BLK * 4 <= 80 * 1024;
    constexpr int NXS = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float lds[(2 + 2 * NXS) * BLK + 8];
    __shared__ float red[4];

    const int nrb = C / RB;
    int b, I;
    if (!xcd_map(blockIdx.x, B, nrb, b, I)) return;
    const int pb = (b + B / 2) % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rw = wave & 1, cw = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const float wb = wt[b], wp = wt[pb];
    const long long sx = (long long)C * HW;
    BlkRegs<BLK> nxb, nxp;                               // the next pair of X blocks on its way in
    auto load_blk = [&](const float* src_, float* dst) {  // a block at once (prologue)
        BlkRegs<BLK> r;
        blk_request<BLK>(src_, r, tid);
        blk_commit<BLK>(dst, r, tid);
    };
    auto request_x = [&](int J) {                        // (clamped: requests are never behind a branch)
        const int Jc = J < nrb ? J : nrb - 1;
        blk_request<BLK>(x + b * sx + (long long)Jc * BLK, nxb, tid);
        blk_request<BLK>(x + pb * sx + (long long)Jc * BLK, nxp, tid);
    };
    auto commit_x = [&](int stage) {
        blk_commit<BLK>(lds + (2 + 2 * stage) * BLK, nxb, tid);
        blk_commit<BLK>(lds + (3 + 2 * stage) * BLK, nxp, tid);
    };
    if (tid < 8) lds[(2 + 2 * NXS) * BLK + tid] = 0.f;   // (the last k-step of an odd HW reads one float past a block)
    load_blk(dy + b * sx + (long long)I * BLK, lds);
    load_blk(dy + pb * sx + (long long)I * BLK, lds + BLK);
    if (DB) {
        request_x(0);
        commit_x(0);
        __syncthreads();
    }

    auto gram = [&](const float* si, const float* sj, f32x16& acc) {          // transposed tile: lane = row i, registers = columns
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ai = si + (rw * 32 + l31) * HW + lh;
        const float* bj = sj + (32 * cw + l31) * HW + lh;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bool tail = (HW & 1) && s_ == KS - 1;
            const float av = (tail && lh) ? 0.f : ai[2 * s_];
            const float bv = (tail && lh) ? 0.f : bj[2 * s_];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
        }
    };
    const long long roff = ((long long)(I * RB + rw * 32 + l31)) * C + 32 * cw + 4 * lh;
    const float* wrow = W + (long long)b * C * C + roff;
    const float* orow = W + (long long)pb * C * C + roff;
    float* drow = dW + (long long)b * C * C + roff;
    float part = 0.f;
    for (int J = 0; J < nrb; ++J) {
        f32x4 wv[4], ov[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            wv[g] = *reinterpret_cast<const f32x4*>(wrow + J * RB + 8 * g);
            ov[g] = *reinterpret_cast<const f32x4*>(orow + J * RB + 8 * g);
        }
        const int stg = DB ? (J & 1) : 0;
        if (DB) {
            request_x(J + 1);
        } else {
            request_x(J);
            __syncthreads();                                                 // the previous block's tiles are done with
            commit_x(0);
            __syncthreads();
        }
        f32x16 ab, ap;
        gram(lds, lds + (2 + 2 * stg) * BLK, ab);
        gram(lds + BLK, lds + (3 + 2 * stg) * BLK, ap);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = wv[g][k], o = ov[g][k];
                const float db = a - wb * o, dp = o - wp * a;
                const float sb = (db > 0.f) ? 1.f : ((db < 0.f) ? -1.f : 0.f);
                const float sp = (dp > 0.f) ? 1.f : ((dp < 0.f) ? -1.f : 0.f);
                const float gb = sb * ab[4 * g + k];
                v[k] = gb - wp * sp * ap[4 * g + k];
                part += gb * o;
            }
            *reinterpret_cast<f32x4*>(drow + J * RB + 8 * g) = v;
        }
        if (DB) {
            commit_x((J + 1) & 1);                                           // (the other stage: free since the last barrier)
            __syncthreads();
        }
    }
    part = block_sum<4>(part, red);
    if (tid == 0) dwpart[(long long)b * nrb + I] = -part;
}

static int cin_cci_dw_flash(const float* x, const float* w, const float* wt, const float* dy, float* dw, float* dwpart, int B,
                            int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || C / 64 > CIN_DW_BLOCKS || !aligned16(x) || !aligned16(w) || !aligned16(dy) || !aligned16(dw) ||
        tuning().bcnn_generic == 1)
        return HK_ERR_UNSUPPORTED;
    const dim3 grid(xcd_grid(B, C / 64));
    switch (HW) {
        case 49: hipLaunchKernelGGL((cin_cci_dw_flash_kernel<49>), grid, dim3(256), 0, st, x, w, wt, dy, dw, dwpart, C, B); break;
        case 64: hipLaunchKernelGGL((cin_cci_dw_flash_kernel<64>), grid, dim3(256), 0, st, x, w, wt, dy, dw, dwpart, C, B); break;
        case 36: hipLaunchKernelGGL((cin_cci_dw_flash_kernel<36>), grid, dim3(256), 0, st, x, w, wt, dy, dw, dwpart, C, B); break;
        default: return HK_ERR_UNSUPPORTED;
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// HK_ERR_UNSUPPORTED when the shape is not one the kernel covers (the caller takes the three-kernel chain)
static int cin_sci_flash(const float* x, float* w, float* y, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(w)) return HK_ERR_UNSUPPORTED;
    const dim3 grid(xcd_grid(B, C / 64));
    switch (HW) {
        case 49: hipLaunchKernelGGL((cin_sci_flash_kernel<49>), grid, dim3(256), 0, st, x, w, y, C, B); break;
        case 64: hipLaunchKernelGGL((cin_sci_flash_kernel<64>), grid, dim3(256), 0, st, x, w, y, C, B); break;
        case 36: hipLaunchKernelGGL((cin_sci_flash_kernel<36>), grid, dim3(256), 0, st, x, w, y, C, B); break;
        default: return HK_ERR_UNSUPPORTED;
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// SCI forward for 14 x 14 / 12 x 12 / 10 x 10 maps (a 448^2 input: HW = 196).  With K = HW = 196 the Gram is no longer cheap
// to recompute (two passes of the one-kernel form above would issue three 2 C^2 HW products where the chain issues two),
// so S = -X X^T / HW is MATERIALISED - by the Gram panel kernel of bcnn_fast.hip, which computes every 64 x 64 tile of the
// symmetric matrix once and writes it and its mirror image - and then
//   cin_row_stats_kernel    one wave per row: m_i = max_j S_ij, l_i = sum_j exp(S_ij - m_i), the row held in registers between
//                           the two sweeps, fixed order; parked in the first two floats of the row's own Y storage;
//   cin_ax_kernel<HW, 0>    a workgroup owns 32 rows of one sample and walks the 64-row blocks X_j (LDS-DMA, three stages, each
//                           wave staging the quarter it reads: no workgroup barrier inside the loop);
//                           wave q takes columns 16 q .. + 15 of every block: it loads its 32 x 16 piece of S in the A-operand
//                           layout of the 32x32x2 MFMA (lane = row, 16-byte loads), turns it into P = exp(S - m) / l in
//                           registers, writes P over S - W is written ONCE and never read back - and issues Y += P X_j from
//                           the same registers; the four partial Y tiles meet in LDS at the end (fixed order).
// Against the chain on the generic tile (S written, read and rewritten by the row softmax, read again by the second product:
// 1.3 GB) this moves 1.0 GB and runs both products on kernels built for their shapes.  32-row workgroups: B C / 32 = 1280 at
// the plugin's batch of 20 - five full rounds of the 256 CUs (150 KB of LDS: one workgroup per CU, ONE wave per SIMD: every
// latency is covered by the pipeline below, not by other waves); 64-row ones would be 2.5 rounds.
// Measured at B = 20, C = 2048, 14 x 14 (rocprofv3, MI355X): Gram 196 us + statistics 81 us + this kernel 384 us; the chain on
// the generic tile 588 + 225 + 549 us; rocBLAS bmm + softmax + bmm 940 us end to end (tools/cin_rows.py).  What each step of
// the way to 384 us was worth: operands of MFMA group r + 1 requested before group r issues - nothing on its own (535 us);
// eighths of the sample-major list per XCD instead of whole samples 535 -> 453; three stages and the pieces of S several blocks
// ahead in untracked registers 453 -> 460 (no gain: the wait was never latency); the step's 17 vector-memory instructions
// dealt over the MFMA groups instead of issued in one burst 460 -> 407; the last four columns of Y on the vector ALU 407 ->
// 384.  Without S and W traffic the same loop takes 330 us (= 0.62 of the matrix pipe's nominal peak, where the other fp32
// MFMA kernels of this library sit); streaming hints on the loads of S / stores of W changed nothing (382 vs 385 us).
// NV > 0: C = 256 NV - the row stays in registers between the two sweeps (one read of S); NV = 0: any C % 4 == 0, the second
// sweep re-reads the row (from L2)
template <int NV>
__global__ __launch_bounds__(256) void cin_row_stats_kernel(const float* __restrict__ s, float* __restrict__ y, int C, int HW,
                                                            long long rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr float LOG2E = 1.4426950408889634f;
    const f32x4* p = reinterpret_cast<const f32x4*>(s + row * C);
    float m = -3.402823466e38f, l = 0.f;
    auto vmax = [](float a, const f32x4& v) { return fmaxf(fmaxf(a, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); };
    auto vsum = [&](const f32x4& v) {
        return (__builtin_amdgcn_exp2f((v[0] - m) * LOG2E) + __builtin_amdgcn_exp2f((v[1] - m) * LOG2E)) +
               (__builtin_amdgcn_exp2f((v[2] - m) * LOG2E) + __builtin_amdgcn_exp2f((v[3] - m) * LOG2E));
    };
    if constexpr (NV > 0) {
        f32x4 v[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u] = p[lane + 64 * u];
#pragma unroll
        for (int u = 0; u < NV; ++u) m = vmax(m, v[u]);
        m = wave_max(m);
#pragma unroll
        for (int u = 0; u < NV; ++u) l += vsum(v[u]);
    } else {
        for (int f = lane; f < C / 4; f += 64) m = vmax(m, p[f]);
        m = wave_max(m);
        for (int f = lane; f < C / 4; f += 64) l += vsum(p[f]);
    }
    l = wave_sum(l);
    if (lane == 0) { y[row * HW] = m; y[row * HW + 1] = l; }
}

// MODE 0: the forward above (w: S in, W out; y: statistics in, Y out).
// The same pipeline serves the two products of the backward whose big operand is a C x C matrix read ONCE (hk_cin_sci_bwd at
// these map sizes; x = the C x HW operand that is staged, w = the matrix, read only):
// MODE 1: y  = scale * w^T x          (W^T dY: the piece of w^T is a COLUMN piece of w - eight 4-byte loads per lane and block,
//                                      32 lanes on 128 consecutive bytes of one row of w)
// MODE 2: y += scale * (w + w^T) x    ((dG + dG^T) X / HW in ONE pass: row piece + column piece added in registers - one
//                                      product on the matrix pipe where the chain on the generic tile ran two)
template <int HW, int MODE>
__global__ __launch_bounds__(256, 1) void cin_ax_kernel(const float* __restrict__ x, float* __restrict__ w, float* __restrict__ y,
                                                        int C, int B, float scale) {
    constexpr int RB = 32;                               // rows per workgroup
    constexpr int CB = 64;                               // rows of X per column block
    constexpr int BLK = CB * HW;                         // floats of one block of X (contiguous in memory)
    // A wave reads only ITS quarter of a staged block (rows 16 q .. 16 q + 15: the k of its 16 columns of w), so it also
    // requests exactly those rows - 16 HW floats = NP16 pieces of 1 KB and, for 14 x 14 and 10 x 10 maps, one of 256 bytes -
    // and the four waves never meet inside the loop: a counted wait ends a step, not a workgroup barrier.
    constexpr int QF = 16 * HW;                          // floats of a wave's quarter of a block
    constexpr int NP16 = QF / 256;                       // its 1 KB pieces
    constexpr int NP4 = (QF % 256) / 64;                 // ... and 256-byte pieces (4 bytes per lane)
    constexpr int NPW = NP16 + NP4;                      // requests per wave and block
    static_assert(QF % 64 == 0, "a quarter block is a whole number of 256-byte pieces");
    constexpr int PPG = (NPW + 6) / 7;                   // ... issued per MFMA group (seven of the eight groups of a step)
    constexpr bool REMV = HW % 32 == 4;                  // 14 x 14 and 10 x 10 maps: the last FOUR columns of Y on the vector ALU (an
                                                         // eighth 32-column tile of MFMAs for 4 of 196 columns is 12.5 % of the matrix work)
    constexpr int NT2 = REMV ? HW / 32 : (HW + 31) / 32; // 32-column tiles of Y
    constexpr bool ROWP = MODE != 1, COLP = MODE != 0;   // which pieces of w a step loads: 16-byte row pieces, 4-byte column pieces
    constexpr int NL = (ROWP ? 2 : 0) + (COLP ? 8 : 0);  // loads per step and lane
    static_assert(BLK % 256 == 0, "a block of X is a whole number of 1 KB pieces");
    static_assert(3 * (NT2 * 16 * 64 + 256) <= 3 * BLK, "the partial Y tiles of three waves fit the stages");
    __shared__ __attribute__((aligned(16))) float lds[3 * BLK + 32];

    const int nrb = C / RB, ncb = C / CB;
    // workgroup -> (sample, row block): XCD k (blockIdx % 8) takes the k-th EIGHTH of the sample-major list - neighbours in time
    // walk the same X from the same L2 - rather than whole samples (xcd_map): 20 samples on 8 XCDs are 3 on four of them and
    // 2 on the others, six rounds of the 32 CUs where the work is five
    const int per = (B * nrb + NXCD - 1) / NXCD;
    const int lin = (int)(blockIdx.x % NXCD) * per + (int)(blockIdx.x / NXCD);
    if (lin >= B * nrb) return;
    const int b = lin / nrb, I = lin % nrb;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* xb = x + (long long)b * C * HW;
    constexpr float LOG2E = 1.4426950408889634f;

    // request u of this wave's quarter of a block (src / dst: the block in memory / its stage)
    auto dma_piece = [&](const float* src, float* dst, int u) {
        if (u < NP16) glds16(src + QF * q + 256 * u + 4 * lane, dst + QF * q + 256 * u);
        else glds4(src + QF * q + 256 * NP16 + 64 * (u - NP16) + lane, dst + QF * q + 256 * NP16 + 64 * (u - NP16));
    };
    auto dma_blk = [&](int blk, float* dst) {
        const float* src = xb + (long long)blk * BLK;
#pragma unroll
        for (int u = 0; u < NPW; ++u) dma_piece(src, dst, u);
    };
    const long long row = (long long)b * C + I * RB + l31;
    float* wrow = w + row * C + 16 * q + 4 * lh;         // this lane's row of S / W: columns 16 q + 8 g + 4 lh .. + 3 of a block
    // column piece: rows 64 blk + 16 q + 4 lh + (r & 3) + 8 (r >> 2) of w, column 32 I + l31 (= element [l31][k] of the transpose)
    const float* wcol = w + ((long long)b * C + 16 * q + 4 * lh) * C + I * RB + l31;
    float m = 0.f, rl = 1.f;
    if (MODE == 0) { m = y[row * HW]; rl = 1.0f / y[row * HW + 1]; }   // (this row's Y is written at the very end)
    auto load_s = [&](int blk, f32x4 (&d)[2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) HK_LOAD16_ASYNC(d[g], wrow + (long long)blk * CB + 8 * g);
    };
    auto load_t1 = [&](int blk, int r, float& d) { HK_LOAD4_ASYNC(d, wcol + ((long long)blk * CB + (r & 3) + 8 * (r >> 2)) * C); };
    // element r of the A operand from the pieces that arrived: P = exp(S - m) / l, or w^T, or w + w^T
    auto aval = [&](const f32x4 (&sv)[2], const float (&tv)[8], int r) {
        if (MODE == 0) return __builtin_amdgcn_exp2f((sv[r >> 2][r & 3] - m) * LOG2E) * rl;
        if (MODE == 1) return tv[r];
        return sv[r >> 2][r & 3] + tv[r];
    };

    // Pipeline.  In step J the MFMAs consume P(J) - registers - and X_J - stage J % 3 - while block J + 2 of X is on its way
    // into the third stage and the A operand of step J + 1 is formed from the piece of w that has arrived, one element per
    // MFMA group.  The pieces travel in registers the compiler keeps no books on (HK_LOAD16_ASYNC), so the rule that makes
    // it correct is structural: the loop body is TWO steps of straight-line code (C % 128 == 0: an even number of blocks);
    // the pieces for the next body's two steps (cur -> nxt) are requested in the first step, the counted wait at the end of
    // the SECOND step (everything but that step's NPW LDS-DMA requests has landed) covers them, and only behind it are
    // they copied nxt -> cur.  No register in flight is ever read, moved or renamed, and none is in flight when the loop is
    // left.  (Learnt on the GPU, invisible in the emulator: a set carried across the loop edge while in flight gets copied
    // by the compiler at the edge - stale values, 7e-3 errors; a conditional step makes the compiler rename registers at
    // the join behind it, and behind the loop it moves the accumulators into whatever registers it considers free.
    // tests/test_isa_static.py walks the ISA for any instruction that touches a register between its request and the
    // wait that covers it.)
    if (tid < 32) lds[3 * BLK + tid] = 0.f;              // (read by the last tile's columns >= HW of a block's last row)
    f32x4 s0[2], scur[2][2], snxt[2][2];
    float t0[8], tcur[2][8], tnxt[2][8];
    auto load_set = [&](int blk, f32x4 (&sd)[2], float (&td)[8]) {
        const int bc_ = blk < ncb ? blk : ncb - 1;       // (past the end: the last block again - loaded, never used)
        if (ROWP) load_s(bc_, sd);
        else sd[0] = sd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (COLP) load_t1(bc_, r, td[r]);
            else td[r] = 0.f;
        }
    };
    load_set(0, s0, t0);
    load_set(1, scur[0], tcur[0]);
    load_set(2, scur[1], tcur[1]);
    dma_blk(0, lds);
    dma_blk(1, lds + BLK);
    f32x16 yacc[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[n][r] = 0.f;
    float pr[8], pn[8];
    float yrem[4] = {0.f, 0.f, 0.f, 0.f};                // REMV: row l31, columns 32 NT2 .. + 3, this lane half's k
    __builtin_amdgcn_s_waitcnt(HK_VMCNT_IMM(0));         // (the loads of the pieces are not the compiler's to wait for)
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) pr[r] = aval(s0, t0, r);

    int cur = 0;                                         // J % 3: the stage of X_J
    auto step = [&](auto k_tag, int J) __attribute__((always_inline)) {
        constexpr int K = decltype(k_tag)::value;        // step of the body: the piece of block J + 1 is set K of `cur`
        f32x4 (&sa)[2] = scur[K];
        float (&ta)[8] = tcur[K];
        const float* bj = lds + cur * BLK + (16 * q + 4 * lh) * HW + l31;    // register r: row (r & 3) + 8 (r >> 2) of these
        const int J2 = J + 2 < ncb ? J + 2 : ncb - 1;    // (past the end: the last block again, into a stage nobody reads)
        // Y_i += P X_j: A = P registers (i = lane & 31, k = lane half), B = x_j[column][n = lane & 31 (+ 32 n)].  The operands
        // of step r + 1 are requested before the MFMAs of step r issue (one wave per SIMD: nobody else covers an LDS round
        // trip); columns >= HW of the last tile read on into the next row (values of X, the next stage or the pad behind the
        // stages) - unconditional reads keep the loop one basic block - and those columns of Y are never stored.
        float bc[NT2], bn[NT2];
        f32x4 xc = {0.f, 0.f, 0.f, 0.f}, xn = {0.f, 0.f, 0.f, 0.f};   // REMV: X[k][32 NT2 .. + 3], one address per lane half (broadcast)
        const float* bj4 = lds + cur * BLK + (16 * q + 4 * lh) * HW + 32 * NT2;
#pragma unroll
        for (int n = 0; n < NT2; ++n) bc[n] = bj[32 * n];
        if (REMV) xc = *reinterpret_cast<const f32x4*>(bj4);
        __builtin_amdgcn_sched_barrier(0);
        // The step's vector-memory instructions - W(J) out first (MODE 0), then NPW pieces of X_{J+2} and, in the body's first
        // step, the pieces of w for the next body - are dealt over the eight MFMA groups: issued in one burst at the top of
        // the step they fill the CU's address queue and the wave sits in front of it instead of issuing MFMAs (measured:
        // 367 us with the burst and neither S nor W, 273 at the matrix pipe's pace).
        const float* xsrc = xb + (long long)J2 * BLK;
        float* xdst = lds + (cur == 0 ? 2 : cur - 1) * BLK;          // stage (J + 2) % 3
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r + 1 < 8) {
#pragma unroll
                for (int n = 0; n < NT2; ++n) bn[n] = bj[(((r + 1) & 3) + 8 * ((r + 1) >> 2)) * HW + 32 * n];
                if (REMV) xn = *reinterpret_cast<const f32x4*>(bj4 + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * HW);
            }
            pn[r] = aval(sa, ta, r);
            if (MODE == 0 && r == 0) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    *reinterpret_cast<f32x4*>(wrow + (long long)J * CB + 8 * g) = (f32x4){pr[4 * g], pr[4 * g + 1], pr[4 * g + 2], pr[4 * g + 3]};
                __builtin_amdgcn_sched_barrier(0);
            }
            if (r < 7) {
#pragma unroll
                for (int u = r * PPG; u < (r + 1) * PPG && u < NPW; ++u) dma_piece(xsrc, xdst, u);
            }
            if (K == 0) {                                // the next body's pieces: blocks J + 3, J + 4
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int blk = J + 3 + u < ncb ? J + 3 + u : ncb - 1;
                    if (COLP) load_t1(blk, r, tnxt[u][r]);
                    if (ROWP && r == 2 * u + 1) load_s(blk, snxt[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);           // the requests stay here, ahead of this step's MFMAs
#pragma unroll
            for (int n = 0; n < NT2; ++n) {
                yacc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], bc[n], yacc[n], 0, 0, 0);
                if (REMV && n == 0) {                                          // (in the shadow of the MFMA just issued)
#pragma unroll
                    for (int c = 0; c < 4; ++c) HK_FMAC_PINNED(yrem[c], pr[r], xc[c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT2; ++n) bc[n] = bn[n];
            xc = xn;
        }
        // (no workgroup barrier: the stage a wave refills next is the part of it that only this wave has read)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(HK_VMCNT_IMM(NPW + (K == 0 ? 2 * NL : 0)));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) pr[r] = pn[r];
        cur = cur == 2 ? 0 : cur + 1;
    };
    for (int J = 0; J < ncb; J += 2) {
        step(std::integral_constant<int, 0>{}, J);
        step(std::integral_constant<int, 1>{}, J + 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            scur[u][0] = snxt[u][0]; scur[u][1] = snxt[u][1];
#pragma unroll
            for (int r = 0; r < 8; ++r) tcur[u][r] = tnxt[u][r];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    HK_VM_BARRIER(0);                                    // the stages are free - nothing is on its way into them, or into a register
    __builtin_amdgcn_sched_barrier(0);
    // the partial Y of waves 1 .. 3 goes through LDS
    constexpr int YB = NT2 * 16 * 64 + 256;              // floats per wave: the tiles, then the four remainder columns of its 32 rows
    if (REMV) {                                          // the two lane halves hold disjoint k: lower + upper, the same in both
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float o = __shfl_xor(yrem[c], 32, 64);
            yrem[c] = lh ? o + yrem[c] : yrem[c] + o;
        }
    }
    if (q > 0) {
        float* ybuf = lds + (q - 1) * YB;
        if (REMV && lh == 0) *reinterpret_cast<f32x4*>(ybuf + NT2 * 16 * 64 + 4 * l31) = (f32x4){yrem[0], yrem[1], yrem[2], yrem[3]};
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) ybuf[(n * 16 + r) * 64 + lane] = yacc[n][r];
    }
    __syncthreads();
    if (q > 0) return;
    // yacc[n]: standard layout - lane & 31 = column of tile n, registers = rows (r & 3) + 8 (r >> 2) + 4 lh of the 32
    float* yb = y + ((long long)b * C + I * RB) * HW;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int col = 32 * n + l31;
        if (col < HW) {
            float* dst = yb + 4 * lh * HW + col;           // register r: row (r & 3) + 8 (r >> 2) of these
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = MODE == 2 ? dst[((r & 3) + 8 * (r >> 2)) * HW] : 0.f;   // (all 16 in flight at once)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (n * 16 + r) * 64 + lane;
                const float t = ((yacc[n][r] + lds[o]) + lds[YB + o]) + lds[2 * YB + o];
                dst[((r & 3) + 8 * (r >> 2)) * HW] = MODE == 0 ? t : MODE == 1 ? scale * t : fmaf(scale, t, old[r]);
            }
        }
    }
    if (REMV && lh == 0) {
        const float* rb = lds + NT2 * 16 * 64 + 4 * l31;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ((yrem[c] + rb[c]) + rb[YB + c]) + rb[2 * YB + c];
        f32x4* dst = reinterpret_cast<f32x4*>(yb + l31 * HW + 32 * NT2);
        if (MODE == 1) v = v * scale;
        if (MODE == 2) {
            const f32x4 old = *dst;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaf(scale, v[c], old[c]);
        }
        *dst = v;
    }
}

// (C % 128: the kernel's loop body is two 64-row blocks)
static bool cin_ax_covers(int C, int HW) { return C % 128 == 0 && (HW == 196 || HW == 144 || HW == 100); }

template <int MODE>
static int cin_ax_launch(const float* x, float* w, float* y, float scale, int B, int C, int HW, hipStream_t st) {
    const dim3 grid(NXCD * ((B * (C / 32) + NXCD - 1) / NXCD));
    switch (HW) {
        case 196: hipLaunchKernelGGL((cin_ax_kernel<196, MODE>), grid, dim3(256), 0, st, x, w, y, C, B, scale); break;
        case 144: hipLaunchKernelGGL((cin_ax_kernel<144, MODE>), grid, dim3(256), 0, st, x, w, y, C, B, scale); break;
        case 100: hipLaunchKernelGGL((cin_ax_kernel<100, MODE>), grid, dim3(256), 0, st, x, w, y, C, B, scale); break;
        default: return HK_ERR_UNSUPPORTED;
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// HK_ERR_UNSUPPORTED when the shape is not one the three kernels cover
static int cin_sci_stored(const float* x, float* w, float* y, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(w) || !aligned16(y)) return HK_ERR_UNSUPPORTED;
    if (!cin_ax_covers(C, HW)) return HK_ERR_UNSUPPORTED;
    const int rc = gram_fast_scaled(x, -1.0f / (float)HW, w, B, C, HW, st);                     // S = -X X^T / HW   :31-32
    if (rc != HK_OK) return rc;
    const long long rows = (long long)B * C;
    const dim3 sgrid((unsigned)((rows + 3) / 4));
    switch (C) {                                                                                // row max / sum of exp
        case 2048: hipLaunchKernelGGL(cin_row_stats_kernel<8>, sgrid, dim3(256), 0, st, (const float*)w, y, C, HW, rows); break;
        case 1024: hipLaunchKernelGGL(cin_row_stats_kernel<4>, sgrid, dim3(256), 0, st, (const float*)w, y, C, HW, rows); break;
        case 512: hipLaunchKernelGGL(cin_row_stats_kernel<2>, sgrid, dim3(256), 0, st, (const float*)w, y, C, HW, rows); break;
        default: hipLaunchKernelGGL(cin_row_stats_kernel<0>, sgrid, dim3(256), 0, st, (const float*)w, y, C, HW, rows); break;
    }
    HK_LAUNCH_CHECK();
    return cin_ax_launch<0>(x, w, y, 1.f, B, C, HW, st);
}

}  // namespace hk

using namespace hk;

#define HK_TRY(x)                       \
    do {                                \
        int rc__ = (x);                 \
        if (rc__ != HK_OK) return rc__; \
    } while (0)

extern "C" int hk_cin_sci_fwd(const float* x, float* w, float* y, int B, int C, int HW, hk_stream_t stream) {
    if (!x || !w || !y || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (tuning().bcnn_generic != 1) {                  // one kernel where the shape allows (C % 64 == 0, 7x7 / 8x8 / 6x6 maps)
        const int rc = cin_sci_flash(x, w, y, B, C, HW, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
        const int rc2 = cin_sci_stored(x, w, y, B, C, HW, st);        // 14x14 / 12x12 / 10x10 maps: Gram panel kernel, row statistics, softmax . X
        if (rc2 != HK_ERR_UNSUPPORTED) return rc2;
    }
    const LdPlain lx = make_plain(x, (long long)C * HW, HW, C, HW);
    HK_TRY((bgemm_launch<true, true>(lx, lx, make_affine(w, (long long)C * C, C, -1.0f / (float)HW, nullptr, 0.f, 0.f), C, C,
                                     HW, B, st)));                                              // -X X^T / HW   :31-32
    hipLaunchKernelGGL(cin_softmax_rows_kernel, dim3((unsigned)B * C), dim3(256), 0, st, w, C);
    HK_LAUNCH_CHECK();
    const LdPlain lw = make_plain(w, (long long)C * C, C, C, C);
    return bgemm_launch<true, false, CIN_SETS>(lw, lx, make_affine(y, (long long)C * HW, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B,
                                     st);                                                        // Y = W X       :34
}

// dx = W^T dY + (dG + dG^T) X / HW with dG = softmax-backward of (dY X^T + dw_extra).  `dwbuf` [B,C,C] is scratch; on
// entry it holds dw_extra (the gradient that reaches W from the CCI branch) when has_extra != 0.
extern "C" int hk_cin_sci_bwd(const float* x, const float* w, const float* dy, float* dwbuf, int has_extra, float* dx,
                              int B, int C, int HW, hk_stream_t stream) {
    if (!x || !w || !dy || !dwbuf || !dx || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long sx = (long long)C * HW, sw = (long long)C * C;
    const LdPlain lx = make_plain(x, sx, HW, C, HW);
    const LdPlain ldy = make_plain(dy, sx, HW, C, HW);
    const LdPlain lw = make_plain(w, sw, C, C, C);
    if (cin_inside(C, w) && (HW == 49 || HW == 64 || HW == 36) && aligned16(dwbuf) && aligned16(x) && aligned16(dy)) {
        // W^T dY, then the row-owned part in one kernel (dW never stored, dG written once, dG X from registers), then dG^T X
        const LdPlainN cx = cin_cols(x, C, HW), cdy = cin_cols(dy, C, HW);
        HK_TRY((bgemm_launch<false, false, CIN_SETS>(cin_mat(w, C), cdy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));
        HK_TRY(cin_sci_bwd_flash(x, w, dy, dwbuf, has_extra, dx, B, C, HW, st));
        return bgemm_launch<false, false, CIN_SETS>(cin_mat(dwbuf, C), cx, make_affine(dx, sx, HW, 1.0f / (float)HW, nullptr, 1.f, 0.f),
                                                    C, HW, C, B, st);
    }
    // dW = dY X^T (+ extra): 335 MB of result for 49-deep products - the tile leaves as 16-byte stores where it can
    const EpAffine epw = make_affine(dwbuf, sw, C, 1.f, nullptr, has_extra ? 1.f : 0.f, 0.f);
    if (cin_inside(C, dwbuf)) HK_TRY((bgemm_launch<true, true, 0, true>(cin_map(dy, C, HW), cin_map(x, C, HW), epw, C, C, HW, B, st)));
    else if (C % 4 == 0 && aligned16(dwbuf)) HK_TRY((bgemm_launch<true, true, 0, true>(ldy, lx, epw, C, C, HW, B, st)));
    else HK_TRY((bgemm_launch<true, true>(ldy, lx, epw, C, C, HW, B, st)));
    hipLaunchKernelGGL(cin_softmax_bwd_rows_kernel, dim3((unsigned)B * C), dim3(256), 0, st, w, dwbuf, C);
    HK_LAUNCH_CHECK();
    if (tuning().bcnn_generic != 1 && cin_ax_covers(C, HW) && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(w) &&
        aligned16(dwbuf)) {
        // 14x14 / 12x12 / 10x10 maps: each C x C matrix streamed once through the forward's pipeline (cin_ax_kernel) -
        // dx = W^T dY, then dx += (dG + dG^T) X / HW as ONE product
        HK_TRY((cin_ax_launch<1>(dy, const_cast<float*>(w), dx, 1.f, B, C, HW, st)));
        return cin_ax_launch<2>(x, dwbuf, dx, 1.0f / (float)HW, B, C, HW, st);
    }
    if (cin_inside(C, w) && aligned16(dwbuf)) {
        // (dG + dG^T) X as two products over dG - the transposed half of LdSym is a 4-byte gather with an 8 KB stride
        const LdPlainN cx = cin_cols(x, C, HW), cdy = cin_cols(dy, C, HW);
        HK_TRY((bgemm_launch<false, false, CIN_SETS>(cin_mat(w, C), cdy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));
        const EpAffine acc = make_affine(dx, sx, HW, 1.0f / (float)HW, nullptr, 1.f, 0.f);
        HK_TRY((bgemm_launch<true, false, CIN_SETS>(cin_mat(dwbuf, C), cx, acc, C, HW, C, B, st)));
        return bgemm_launch<false, false, CIN_SETS>(cin_mat(dwbuf, C), cx, acc, C, HW, C, B, st);
    }
    HK_TRY((bgemm_launch<false, false, CIN_SETS>(lw, ldy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));  // W^T dY
    LdSym ls;
    ls.p = dwbuf; ls.bs = sw; ls.d = C;
    return bgemm_launch<true, false, CIN_SETS>(ls, lx, make_affine(dx, sx, HW, 1.0f / (float)HW, nullptr, 1.f, 0.f), C, HW, C, B, st);
}

extern "C" int hk_cin_cci_fwd(const float* x, const float* w, const float* wt, float* y, int B, int C, int HW,
                              hk_stream_t stream) {
    if (!x || !w || !wt || !y || B <= 0 || (B & 1) || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (cin_inside(C, w)) {
        LdAbsDiffV lv;
        lv.p = w; lv.wt = wt; lv.C = C; lv.B = B; lv.wb = 0.f;
        return bgemm_launch<true, false, CIN_SETS>(lv, cin_cols(x, C, HW), make_affine(y, (long long)C * HW, HW, 1.f, nullptr, 0.f, 0.f),
                                               C, HW, C, B, (hipStream_t)stream);
    }
    LdAbsDiff la;
    la.p = w; la.wt = wt; la.C = C; la.B = B;
    const LdPlain lx = make_plain(x, (long long)C * HW, HW, C, HW);
    return bgemm_launch<true, false, CIN_SETS>(la, lx, make_affine(y, (long long)C * HW, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B,
                                     (hipStream_t)stream);                                       // :52-54
}

extern "C" size_t hk_cin_cci_ws_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    return ((size_t)B * C * C + (size_t)B * CIN_DW_BLOCKS) * sizeof(float) + 256;
}

// dy [B,C,HW] -> dx [B,C,HW] (through Wc X), dw [B,C,C] (gradient reaching W), dwt [B] (gradient of the weights w_b)
extern "C" int hk_cin_cci_bwd(const float* x, const float* w, const float* wt, const float* dy, float* dx, float* dw,
                              float* dwt, int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !w || !wt || !dy || !dx || !dw || !dwt || B <= 0 || (B & 1) || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_cin_cci_ws_bytes(B, C)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long sx = (long long)C * HW, sw = (long long)C * C;
    float* dwc = (float*)ws;                       // dL/dWc = dY X^T
    float* dwpart = dwc + (size_t)B * sw;
    const LdPlain lx = make_plain(x, sx, HW, C, HW);
    const LdPlain ldy = make_plain(dy, sx, HW, C, HW);
    if (cin_inside(C, w)) {                        // dWc recomputed inside the kernel that consumes it, never stored
        LdAbsDiffV lv;
        lv.p = w; lv.wt = wt; lv.C = C; lv.B = B; lv.wb = 0.f;
        const int rcf = cin_cci_dw_flash(x, w, wt, dy, dw, dwpart, B, C, HW, st);
        if (rcf == HK_OK) {
            HK_TRY((bgemm_launch<false, false, CIN_SETS>(lv, cin_cols(dy, C, HW), make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C,
                                                         B, st)));                                 // Wc^T dY
            hipLaunchKernelGGL(cin_cci_dw_reduce_kernel, dim3(B), dim3(64), 0, st, (const float*)dwpart, dwt, C / 64);
            HK_LAUNCH_CHECK();
            return HK_OK;
        }
        if (rcf != HK_ERR_UNSUPPORTED) return rcf;
    }
    const EpAffine epw = make_affine(dwc, sw, C, 1.f, nullptr, 0.f, 0.f);
    if (cin_inside(C, dwc)) HK_TRY((bgemm_launch<true, true, 0, true>(cin_map(dy, C, HW), cin_map(x, C, HW), epw, C, C, HW, B, st)));
    else if (C % 4 == 0 && aligned16(dwc)) HK_TRY((bgemm_launch<true, true, 0, true>(ldy, lx, epw, C, C, HW, B, st)));
    else HK_TRY((bgemm_launch<true, true>(ldy, lx, epw, C, C, HW, B, st)));
    LdAbsDiff la;
    la.p = w; la.wt = wt; la.C = C; la.B = B;
    if (cin_inside(C, w)) {
        LdAbsDiffV lv;
        lv.p = w; lv.wt = wt; lv.C = C; lv.B = B; lv.wb = 0.f;
        HK_TRY((bgemm_launch<false, false, CIN_SETS>(lv, cin_cols(dy, C, HW), make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));
    } else {
        HK_TRY((bgemm_launch<false, false, CIN_SETS>(la, ldy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));  // Wc^T dY
    }
    hipLaunchKernelGGL(cin_cci_dw_kernel, dim3(CIN_DW_BLOCKS, B), dim3(256), 0, st, w, wt, (const float*)dwc, dw, dwpart, C, B,
                       CIN_DW_BLOCKS);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(cin_cci_dw_reduce_kernel, dim3(B), dim3(64), 0, st, (const float*)dwpart, dwt, CIN_DW_BLOCKS);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
