// Compact bilinear pooling (replaces model/methods/CBCNN.py:96-135).
//
// The reference computes, per spatial position p, two count sketches of x_p
// (dense [512,6000] matmuls), multiplies their FFTs, inverse-FFTs, then sums over
// positions.  Circular convolution of two count sketches is the count sketch of
// the outer product with hash (h1[i]+h2[j]) mod D and sign s1[i] s2[j], and the
// sum over positions commutes with it, so
//     c[b,k] = sum_{(i,j): (h1[i]+h2[j]) mod D = k} s1[i] s2[j] (X X^T)[b,i,j]
// exactly (2.8e-7 norm-wise vs the FFT route).  MI355X
// design: one fp32-MFMA Gram (the BCNN primitive) + a deterministic CSR
// gather-reduce per bin + one normalisation kernel; no FFT, no [B*HW, 6000]
// intermediates (the reference moves ~1 GB per step through them).
//
// Backward: dc from the signed-sqrt / l2 chain, then
//     dX = (dG + dG^T) X,  dG_ij = s1_i s2_j dc[(h1_i + h2_j) mod D]
// with dG + dG^T generated on the fly by the GEMM's A-operand loader.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include <mutex>
#include <unordered_map>
#include "hk_bgemm.h"
#include "hk_cbp_fused.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

int gram_fast_raw(const float* x, float* mu, float alpha, float* g, int B, int C, int HW, hipStream_t st);
int cbp_fast_bwd(const float* x, const int* h1, const int* h2, const float* s1, const float* s2, const float* dc, int D,
                 float* dx, int B, int C, int HW, hipStream_t st);
int cbp_fast_bwd_fused(const float* x, const int* h1, const int* h2, const float* s1, const float* s2, const float* y,
                       const float* dy, const float* c_raw, const float* inv_norm, int D, float* dx, int B, int C, int HW,
                       hipStream_t st);
static inline bool force_generic() { return tuning().bcnn_generic == 1; }   // A/B lever (hk_tuning_set)

struct CbpPlan {  // device-side view of the plan blob
    const int* h1;
    const int* h2;
    const float* s1;
    const float* s2;
    const int* off;      // [D+1]
    const unsigned* ent; // [C*C]  bit31 = negative sign, low bits = i*C + j
    // inverse of h2 over its NON-EMPTY bins (row-sketch kernel): slot t holds bin nzb[t] and the channels
    // nzj[nzo[t] .. nzo[t+1]) hashed there (bit31 = negative s2)
    const int* nzb;      // [C]
    const int* nzo;      // [C+1]
    const unsigned* nzj; // [C]
    int nzn;             // number of non-empty h2 bins (<= C)
    int emax;            // largest number of channels sharing one h2 bin
    const unsigned* lists;   // fused forward (hk_cbp_fused.h): the (C / 64)^2 sorted tile lists; nullptr when not built
};

__host__ __device__ inline size_t cbp_align(size_t x) { return (x + 15) & ~(size_t)15; }
// the blob carries the tile lists of the fused forward for every shape that kernel covers (whether THESE hashes allow
// it is decided at build time and recorded in the word BEHIND the lists and in the host-side directory below:
// hk_cbp_plan_build)
static inline bool cbp_has_lists(int C, int D) { return C % 64 == 0 && C <= 1024 && D + 1 <= CBF_DMAX; }

// Whether the tile lists of a plan blob were built (they are not when some bin holds more entries of one 64x64 tile than
// a lane's steps - tiny D): decided on the host by hk_cbp_plan_build, needed on the host by hk_cbp_fwd, and the blob is
// device memory.  A directory device pointer -> flag, written by plan_build, erased by hk_cbp_plan_destroy and read by fwd
// under a mutex; a blob that is not in it (copied by the caller) takes the unfused path.  An address is only trusted
// between its build and its destroy: a caller that frees a plan's memory calls hk_cbp_plan_destroy first, so a different
// blob later placed at the same address cannot inherit a stale "fused ok".
static std::mutex g_plan_mu;
static std::unordered_map<const void*, int> g_plan_fused;
static inline void plan_note(const void* plan, int fused_ok) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_fused[plan] = fused_ok;
}
static inline int plan_fused_ok(const void* plan) {          // 0 no lists, 1 lists, 2 lists that allow paired steps
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const auto it = g_plan_fused.find(plan);
    return it != g_plan_fused.end() ? it->second : 0;
}

static inline CbpPlan cbp_view(const void* plan, int C, int D) {
    const char* p = (const char*)plan + 16;
    CbpPlan v;
    v.h1 = (const int*)p;            p += cbp_align((size_t)C * 4);
    v.h2 = (const int*)p;            p += cbp_align((size_t)C * 4);
    v.s1 = (const float*)p;          p += cbp_align((size_t)C * 4);
    v.s2 = (const float*)p;          p += cbp_align((size_t)C * 4);
    v.off = (const int*)p;           p += cbp_align((size_t)(D + 1) * 4);
    v.ent = (const unsigned*)p;      p += cbp_align((size_t)C * C * 4);
    v.nzb = (const int*)p;           p += cbp_align((size_t)C * 4);
    v.nzo = (const int*)p;           p += cbp_align((size_t)(C + 1) * 4);
    v.nzj = (const unsigned*)p;          p += cbp_align((size_t)C * 4);
    v.lists = cbp_has_lists(C, D) ? (const unsigned*)p : nullptr;
    v.nzn = ((const int*)plan)[2];
    v.emax = ((const int*)plan)[3];
    return v;
}

// c_raw[b,k] = sum over the bin's (i,j) list of +-G[b,i,j]; one wave per bin, lanes stride over its ~44 entries,
// butterfly reduction (fixed order).  The gathers touch one 64-B sector per 4-B entry, so this stage is bound by
// L2 sector traffic (~130 us at B=64); a lane-per-bin walk of a transposed (ELL) table - coalesced entry reads,
// 8 gathers in flight per lane - was measured SLOWER (190 us): same sector traffic, fewer waves.  Kept as the
// fallback for C > 512 / D > 8192; the default is cbp_rowsketch_kernel below.
__global__ __launch_bounds__(256) void cbp_bin_kernel(const float* __restrict__ G, const int* __restrict__ off,
                                                      const unsigned* __restrict__ ent, float* __restrict__ c_raw,
                                                      int CC, int D) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= D) return;
    const int lane = threadIdx.x & 63;
    const float* g = G + (long long)b * CC;
    const int lo = off[k], hi = off[k + 1];
    float s = 0.f;
    for (int e = lo + lane; e < hi; e += 64) {
        const unsigned u = ent[e];
        const float v = g[u & 0x7fffffffu];
        s += (u >> 31) ? -v : v;
    }
    s = wave_sum(s);
    if (lane == 0) c_raw[(long long)b * D + k] = s;
}

// Row-sketch binning (default for C <= 512, D <= 8192).  For one row i of G, its contribution to the output is a
// circular shift of the count sketch of that row:  c[k] += s1_i * r_i[(k - h1_i) mod D],  r_i[m] = sum_{j: h2_j = m} s2_j G_ij.
// A workgroup owns 64 rows of one sample's G: r_i (D floats, <= 512 non-zeros) lives in LDS, each thread owns the
// non-empty h2 bins t, t+256 (their G entries are fetched one row ahead, so the gathers overlap the accumulate
// phase) and D/256 output bins in registers.  Reads of G are confined to one 2 KB row at a time (vs one 64-B sector
// per 4-B entry anywhere in a 1 MB matrix for the CSR gather above), every bin is summed in a fixed order, and the
// 64-row partials are added in a fixed order by cbp_partsum_kernel: deterministic, no atomics.

constexpr size_t CBP_LOC_LDS_MAX = 144 * 1024;   // dynamic LDS of cbp_loc_bwd_kernel (dc [D] + columns + hashes): of the CU's 160 KB
constexpr int CBP_EMAX = 4;      // channels per non-empty h2 bin held in registers (plan build checks the hashes)
constexpr int CBP_RB = 4;        // rows of G staged per LDS block

// Measured history of this kernel (C=512, D=6000, per launch, independent of B up to 2 workgroups / CU):
//   v1 entries gathered from global memory one row ahead                       ~190 us (a memory latency per row)
//   v2 G streamed through LDS in row blocks, coalesced 16-B loads              112 us (rocprofv3): the per-bin `if`s
//      made every LDS read its own basic block, so ~32 LDS latencies per row were exposed
//   v3 (this) branch-free: padded entries carry sign 0, invalid bins write to a dump slot, and the sketch is kept
//      REPLICATED (r[i + D] = r[i]) so the circular shift by h1[i] is a plain base + 256 q with immediate offsets
template <int NBT, int NQ8>
__global__ __launch_bounds__(256) void cbp_rowsketch_kernel(const float* __restrict__ G, CbpPlan pl,
                                                            float* __restrict__ part, int C, int D, int nchunk) {
    constexpr int NQ = 8 * NQ8;                        // output bins per thread
    HK_DYN_LDS16(smem);
    const int RS = D + 256 * NQ;                       // replicated sketch length; r[RS] is the dump slot
    float* r = smem;
    const int poff = ((RS + 1 + 3) / 4) * 4;
    int* sh1 = reinterpret_cast<int*>(smem + poff);    // [64]  h1 of the chunk's rows
    float* ss1 = smem + poff + 64;                     // [64]  s1 of the chunk's rows
    float* gb = smem + poff + 128;                     // [2][CBP_RB * C] staged rows of G
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int i0 = ch * 64, i1 = (i0 + 64 < C) ? i0 + 64 : C;
    const int nrows = i1 - i0, nblk = (nrows + CBP_RB - 1) / CBP_RB;
    const int blk4 = CBP_RB * C / 4;                   // float4 per full block (<= 512 for C <= 512)
    const float* gbase = G + ((long long)b * C + i0) * C;

    for (int k = tid; k <= RS; k += 256) r[k] = 0.f;
    if (tid < 64 && tid < nrows) {
        sh1[tid] = pl.h1[i0 + tid];
        ss1[tid] = pl.s1[i0 + tid];
    }
    int jx[NBT][CBP_EMAX], w[NBT][3];
    float sg[NBT][CBP_EMAX];
#pragma unroll
    for (int u = 0; u < NBT; ++u) {
        const int t = tid + 256 * u;
        const bool ok = t < pl.nzn;
        const int mb = ok ? pl.nzb[t] : -1;
#pragma unroll
        for (int cpy = 0; cpy < 3; ++cpy) {            // D >= RS / 3 (dispatch): at most 3 replicas of a bin
            const int idx = mb + cpy * D;
            w[u][cpy] = (mb >= 0 && idx < RS) ? idx : RS;
        }
        const int lo = ok ? pl.nzo[t] : 0, hi = ok ? pl.nzo[t + 1] : 0;
#pragma unroll
        for (int e = 0; e < CBP_EMAX; ++e) {
            const bool valid = lo + e < hi;
            const unsigned v = valid ? pl.nzj[lo + e] : 0u;
            jx[u][e] = (int)(v & 0x7fffffffu);
            sg[u][e] = valid ? ((v >> 31) ? -1.f : 1.f) : 0.f;
        }
    }
    float c[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) c[q] = 0.f;

    f32x4 st[2];                                       // this thread's share of one staged block
#define HK_BLK_LOAD(blk_)                                                                            \
    do {                                                                                             \
        const int left_ = nrows - (blk_) * CBP_RB;                                                   \
        const int lim4_ = ((left_ < CBP_RB ? left_ : CBP_RB) * C) / 4;                               \
        const f32x4* src_ = reinterpret_cast<const f32x4*>(gbase + (long long)(blk_) * CBP_RB * C);  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                              \
            const int f_ = tid + 256 * u;                                                            \
            st[u] = src_[f_ < lim4_ ? f_ : 0];                                                       \
        }                                                                                            \
    } while (0)
#define HK_BLK_STORE(buf_)                                                                           \
    do {                                                                                             \
        f32x4* dst_ = reinterpret_cast<f32x4*>(gb + (buf_) * CBP_RB * C);                            \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                              \
            const int f_ = tid + 256 * u;                                                            \
            if (f_ < blk4) dst_[f_] = st[u];                                                         \
        }                                                                                            \
    } while (0)

    HK_BLK_LOAD(0);
    HK_BLK_STORE(0);
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        if (blk + 1 < nblk) HK_BLK_LOAD(blk + 1);            // in flight while this block is consumed
        const float* gcur = gb + cur * CBP_RB * C;
        const int left = nrows - blk * CBP_RB;
        const int rmax = left < CBP_RB ? left : CBP_RB;
        for (int rr = 0; rr < rmax; ++rr) {
            const float* grow = gcur + rr * C;
            float x[NBT][CBP_EMAX];
#pragma unroll
            for (int u = 0; u < NBT; ++u)
#pragma unroll
                for (int e = 0; e < CBP_EMAX; ++e) x[u][e] = grow[jx[u][e]];
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                float sacc = 0.f;                            // signed sum in channel order (padding adds 0 * x)
#pragma unroll
                for (int e = 0; e < CBP_EMAX; ++e) sacc += sg[u][e] * x[u][e];
                r[w[u][0]] = sacc;                           // non-empty bins are overwritten every row, empty stay 0
                r[w[u][1]] = sacc;
                r[w[u][2]] = sacc;
            }
            HK_LDS_BARRIER();
            const int li = blk * CBP_RB + rr;
            const float s1i = ss1[li];
            int base = tid - sh1[li];
            if (base < 0) base += D;
            const float* rb = r + base;                      // c[k] += s1 r[(k - h1) mod D], k = tid + 256 q
#pragma unroll
            for (int q = 0; q < NQ; ++q) c[q] += s1i * rb[256 * q];
            HK_LDS_BARRIER();
        }
        if (blk + 1 < nblk) {
            HK_BLK_STORE(cur ^ 1);
            HK_LDS_BARRIER();
        }
    }
#undef HK_BLK_LOAD
#undef HK_BLK_STORE
    float* pp = part + ((long long)b * nchunk + ch) * D;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int k = tid + 256 * q;
        if (k < D) pp[k] = c[q];
    }
}

// Row-scatter variant (HK_CBP_CSR=2; written after round 1's GPU budget was spent, emulation-validated, not yet timed).
// The row-sketch kernel above is bound by LDS bandwidth, not latency: every row it re-reads the whole replicated sketch
// (24 KB per row and workgroup, 91 % of it zeros - a row has <= C = 512 non-empty bins out of D = 6000), 1.25 GB of
// LDS traffic per launch at B = 64.  Here the partial result c[D] of the chunk lives in LDS instead and each row ADDS
// its <= 512 non-zero sketch entries into it at (m + h1_i) mod D: within one row the targets are distinct (distinct m),
// so plain read-modify-writes are race-free, and one LDS barrier per row orders the rows - the same per-bin summation
// order (rows ascending) and the same `c += s1 * r` expression as the row-sketch kernel, hence bit-identical partials,
// with ~8x less LDS traffic, one barrier per row instead of two and 41 KB of LDS (3 workgroups per CU).

// rows of G per staged block of the row-scatter kernel, and float4 per thread of one block (C <= 512).  Eight rows: the
// block's global loads are requested one block ahead, and with four-row blocks (~1 us of work) the ~2 us of HBM latency
// was exposed at every one of the 16 blocks of a chunk - that, not the ordered steps, was most of the kernel's 33 us.
constexpr int CBP_SRB = 8;
constexpr int CBP_SLD = CBP_SRB * 512 / 4 / 256;

template <int NBT>
__global__ __launch_bounds__(256) void cbp_rowscatter_kernel(const float* __restrict__ G, CbpPlan pl,
                                                             float* __restrict__ part, int C, int D, int nchunk) {
    HK_DYN_LDS16(smem);
    float* c = smem;                                   // [D] bins of this chunk; c[D] is the dump slot
    const int poff = ((D + 1 + 3) / 4) * 4;
    int* sh1 = reinterpret_cast<int*>(smem + poff);    // [64]  h1 of the chunk's rows
    float* ss1 = smem + poff + 64;                     // [64]  s1 of the chunk's rows
    float* gb = smem + poff + 128;                     // [2][CBP_SRB * C] staged rows of G
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int i0 = ch * 64, i1 = (i0 + 64 < C) ? i0 + 64 : C;
    const int nrows = i1 - i0, nblk = (nrows + CBP_SRB - 1) / CBP_SRB;
    const int blk4 = CBP_SRB * C / 4;
    const float* gbase = G + ((long long)b * C + i0) * C;

    for (int k = tid; k <= D; k += 256) c[k] = 0.f;
    if (tid < 64 && tid < nrows) {
        sh1[tid] = pl.h1[i0 + tid];
        ss1[tid] = pl.s1[i0 + tid];
    }
    int jx[NBT][CBP_EMAX], mb[NBT];
    float sg[NBT][CBP_EMAX];
    bool one = true, one_hi = true;                        // every bin of this lane / of its units u >= 1 has one channel
#pragma unroll
    for (int u = 0; u < NBT; ++u) {
        const int t = tid + 256 * u;
        const bool ok = t < pl.nzn;
        mb[u] = ok ? pl.nzb[t] : -1;
        const int lo = ok ? pl.nzo[t] : 0, hi = ok ? pl.nzo[t + 1] : 0;
#pragma unroll
        for (int e = 0; e < CBP_EMAX; ++e) {
            const bool valid = lo + e < hi;
            const unsigned v = valid ? pl.nzj[lo + e] : 0u;
            jx[u][e] = (int)(v & 0x7fffffffu);
            sg[u][e] = valid ? ((v >> 31) ? -1.f : 1.f) : 0.f;
        }
        one = one && (hi - lo <= 1);
        if (u > 0) one_hi = one_hi && (hi - lo <= 1);
    }
    // no lane of this WAVE owns a bin with more than one channel (true for all but the first wave: the plan orders the
    // bins by channel count): the sketches then take one gather per bin instead of the padded CBP_EMAX.  The kernel
    // issues instructions 35 % of its wave cycles (profiles/r2_pool_kernels_pmc.csv) - it is bound by their number.
    const bool single = __builtin_amdgcn_readfirstlane((int)wave_max(one ? 0.f : 1.f)) == 0;
    const bool single_hi = __builtin_amdgcn_readfirstlane((int)wave_max(one_hi ? 0.f : 1.f)) == 0;   // (the first wave)

    f32x4 st[CBP_SLD];                                  // this thread's share of one staged block
#define HK_BLK_LOAD(blk_)                                                                            \
    do {                                                                                             \
        const int left_ = nrows - (blk_) * CBP_SRB;                                                   \
        const int lim4_ = ((left_ < CBP_SRB ? left_ : CBP_SRB) * C) / 4;                               \
        const f32x4* src_ = reinterpret_cast<const f32x4*>(gbase + (long long)(blk_) * CBP_SRB * C);  \
        _Pragma("unroll") for (int u = 0; u < CBP_SLD; ++u) {                                              \
            const int f_ = tid + 256 * u;                                                            \
            st[u] = src_[f_ < lim4_ ? f_ : 0];                                                       \
        }                                                                                            \
    } while (0)
#define HK_BLK_STORE(buf_)                                                                           \
    do {                                                                                             \
        f32x4* dst_ = reinterpret_cast<f32x4*>(gb + (buf_) * CBP_SRB * C);                            \
        _Pragma("unroll") for (int u = 0; u < CBP_SLD; ++u) {                                              \
            const int f_ = tid + 256 * u;                                                            \
            if (f_ < blk4) dst_[f_] = st[u];                                                         \
        }                                                                                            \
    } while (0)

    HK_BLK_LOAD(0);
    HK_BLK_STORE(0);
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1;
        if (blk + 1 < nblk) HK_BLK_LOAD(blk + 1);
        const float* gcur = gb + cur * CBP_SRB * C;
        const int left = nrows - blk * CBP_SRB;
        const int rmax = left < CBP_SRB ? left : CBP_SRB;
        // Everything of the block that does not depend on the bins first - the rows' hashes and signs, the row sketches
        // (4 x NBT x 4 gathers of staged G values) and the target bins - all LDS reads in flight together.  What is left
        // in the ordered part is, per row, ONE dependent LDS round trip (read the bin, add, write) and the barrier; a
        // row at a time it was three (hash -> bin address -> bin value) behind the gathers (tools/cbp_lab.py: ~1000
        // ticks per row).  Same operations in the same order per bin: bit-identical partials.
        float add[CBP_SRB][NBT];
        int idx[CBP_SRB][NBT];
        // (the instances of the sketch loops differ in the compile-time gather counts only - of the first bin of a
        //  lane and of its other bins; one wave-uniform branch)
#define HK_SC_SKETCH(EN, EH)                                                                                   \
        _Pragma("unroll") for (int rr = 0; rr < CBP_SRB; ++rr) {                                               \
            const int r_ = rr < rmax ? rr : 0;                                                                 \
            const float* grow = gcur + r_ * C;                                                                 \
            const int li = blk * CBP_SRB + r_;                                                                 \
            const int h1i = sh1[li];                                                                           \
            const float s1i = ss1[li];                                                                         \
            _Pragma("unroll") for (int u = 0; u < NBT; ++u) {                                                  \
                float sacc = 0.f;                            /* signed sum in channel order (padding adds 0 * x) */ \
                _Pragma("unroll") for (int e = 0; e < (u == 0 ? (EN) : (EH)); ++e) sacc += sg[u][e] * grow[jx[u][e]]; \
                int ix = mb[u] + h1i;                        /* bin (h1_i + h2_j) mod D of this row's entry */ \
                if (ix >= D) ix -= D;                                                                          \
                idx[rr][u] = mb[u] >= 0 ? ix : D;                                                              \
                add[rr][u] = s1i * sacc;                                                                       \
            }                                                                                                  \
        }
        if (single) { HK_SC_SKETCH(1, 1) } else if (single_hi) { HK_SC_SKETCH(CBP_EMAX, 1) } else { HK_SC_SKETCH(CBP_EMAX, CBP_EMAX) }
#undef HK_SC_SKETCH
#pragma unroll
        for (int rr = 0; rr < CBP_SRB; ++rr) {
            if (rr < rmax) {                                 // uniform
#pragma unroll
                for (int u = 0; u < NBT; ++u) c[idx[rr][u]] += add[rr][u];
                HK_LDS_BARRIER();                            // the next row may hit the same bins from other lanes
                
            }
        }
        if (blk + 1 < nblk) {
            HK_BLK_STORE(cur ^ 1);
            HK_LDS_BARRIER();
        }
    }
#undef HK_BLK_LOAD
#undef HK_BLK_STORE
    __syncthreads();
    float* pp = part + ((long long)b * nchunk + ch) * D;
    for (int k = tid; k < D; k += 256) pp[k] = c[k];
}

static inline size_t rowscatter_lds(int C, int D) {
    return ((size_t)((D + 1 + 3) / 4) * 4 + 128 + 2 * CBP_SRB * (size_t)C) * sizeof(float);
}

template <int NBT>
static int rowscatter_launch(const float* G, const CbpPlan& pl, float* part, int B, int C, int D, int nchunk,
                             hipStream_t st) {
    const size_t lds = rowscatter_lds(C, D);
    if (lds > 150 * 1024) return HK_ERR_UNSUPPORTED;
    HK_ALLOW_BIG_LDS(&cbp_rowscatter_kernel<NBT>, lds);
    hipLaunchKernelGGL((cbp_rowscatter_kernel<NBT>), dim3(nchunk, B), dim3(256), lds, st, G, pl, part, C, D, nchunk);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

template <int NBT, int NQ8>
static int rowsketch_launch(const float* G, const CbpPlan& pl, float* part, int B, int C, int D, int nchunk,
                            hipStream_t st) {
    const int RS = D + 256 * 8 * NQ8;
    const size_t lds = ((size_t)((RS + 1 + 3) / 4) * 4 + 128 + 2 * CBP_RB * (size_t)C) * sizeof(float);
    HK_ALLOW_BIG_LDS((&cbp_rowsketch_kernel<NBT, NQ8>), lds);
    hipLaunchKernelGGL((cbp_rowsketch_kernel<NBT, NQ8>), dim3(nchunk, B), dim3(256), lds, st, G, pl, part, C, D, nchunk);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// The finishing stage runs one thread per bin on a (ceil(D / 256), B) grid in two launches; round 1's single
// one-workgroup-per-sample norm kernel (two dependent passes of 24 elements per thread on 64 of the 256 CUs) took 12 us
// for 1.5 MB.
// (1) c_raw[b][k] = sum over the 64-row chunks of part[b][chunk][k], in chunk order, and the workgroup's share of
//     |u|^2 = sum_k (|c_k| + 1e-10 where c_k != 0) to ssq[b][blockIdx.x].  part may be c_raw itself (nchunk = 1: the
//     CSR gather wrote c_raw directly) - no __restrict__ on the two.
__global__ __launch_bounds__(256) void cbp_partsum_kernel(const float* part, float* c_raw, float* __restrict__ ssq,
                                                         int D, int nchunk) {
    __shared__ float red[4];
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    float s = 0.f;
    if (k < D) {
        const float* pp = part + (long long)b * nchunk * D + k;
#pragma unroll 8
        for (int q = 0; q < nchunk; ++q) s += pp[(long long)q * D];
        c_raw[(long long)b * D + k] = s;
    }
    // u^2 = |c| + 1e-10 where c != 0; sign(0) = 0 makes u = 0 exactly there
    const float ss = block_sum<4>(s != 0.f ? fabsf(s) + 1e-10f : 0.f, red);
    if (threadIdx.x == 0) ssq[(long long)b * gridDim.x + blockIdx.x] = ss;
}

// u = sign(c) sqrt(|c| + 1e-10) ; y = u / max(|u|_2, 1e-12)      (CBCNN.py:132-133)
// part != nullptr: c_raw[b,k] = sum of the nchunk row-chunk partials (fixed order) is formed here first
// (2) every thread adds the sample's gridDim.x partial sums in the same order (so all agree on n) and writes its bin
__global__ __launch_bounds__(256) void cbp_norm_kernel(const float* __restrict__ c_raw, const float* __restrict__ ssq,
                                                       float* __restrict__ y, float* __restrict__ inv_norm, int D) {
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    float ss = 0.f;
    for (int q = 0; q < (int)gridDim.x; ++q) ss += ssq[(long long)b * gridDim.x + q];
    const float n = fmaxf(sqrtf(ss), 1e-12f);
    if (k < D) {
        const float v = c_raw[(long long)b * D + k];
        const float sg = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
        y[(long long)b * D + k] = sg * sqrtf(fabsf(v) + 1e-10f) / n;
    }
    if (k == 0) inv_norm[b] = 1.0f / n;
}

// One-launch finishing stage (round 3; the two launches above cost ~5 us each at B = 64 for 6 MB of data - launch
// latency).  grid (CBP_FQ, B), 1024 threads: every workgroup of a sample adds the sample's `npart` partial bin vectors
// for ALL D bins (fixed order) - it needs them for the norm - keeps the bins of its own quarter, reduces
// |u|^2 = sum_k (|c_k| + 1e-10 where c_k != 0) in a fixed order (the same value in all CBP_FQ workgroups of the sample),
// and writes c_raw / y for its quarter.  The redundant reads are L2 hits (B x npart x 24 KB in total).
constexpr int CBP_FQ = 4;
constexpr int CBP_FT = 1024;
constexpr int CBP_FK = 8;                             // bins per thread: D <= CBP_FK * CBP_FT
__global__ __launch_bounds__(CBP_FT) void cbp_finish_kernel(const float* __restrict__ part, float* __restrict__ c_raw,
                                                           float* __restrict__ y, float* __restrict__ inv_norm, int D,
                                                           int npart) {
    __shared__ float red[CBP_FT / 64];
    const int b = blockIdx.y, qd = blockIdx.x, tid = threadIdx.x;
    const int per = (D + CBP_FQ - 1) / CBP_FQ, k0 = qd * per, k1 = (k0 + per < D) ? k0 + per : D;
    const float* pb = part + (long long)b * npart * D;
    // bins tid, tid + 1024, ..: CBP_FK independent loads per partial vector in flight (a loop over the bins with the loop
    // over the partials inside is one dependent chain of L2 latencies: 15 us for 6 MB)
    float s[CBP_FK];
#pragma unroll
    for (int j = 0; j < CBP_FK; ++j) s[j] = 0.f;
    for (int q0 = 0; q0 < npart; q0 += 8) {           // eight partial vectors x CBP_FK bins = 64 loads in flight per thread
        float v[8][CBP_FK];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u < npart ? q0 + u : npart - 1;
            const float* pq = pb + (long long)q * D;
#pragma unroll
            for (int j = 0; j < CBP_FK; ++j) {
                const int k = tid + CBP_FT * j;
                v[u][j] = (k < D && q0 + u < npart) ? pq[k] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < CBP_FK; ++j) s[j] += v[u][j];          // per bin: partials added in order q = 0, 1, ..
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CBP_FK; ++j) ss += s[j] != 0.f ? fabsf(s[j]) + 1e-10f : 0.f;   // u^2 = |c| + 1e-10 where c != 0
    const float tot = block_sum<CBP_FT / 64>(ss, red);
    const float n = fmaxf(sqrtf(tot), 1e-12f);
#pragma unroll
    for (int j = 0; j < CBP_FK; ++j) {
        const int k = tid + CBP_FT * j;
        if (k >= k0 && k < k1) {
            const float v = s[j];
            const float sg = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);             // sign(0) = 0: u = 0 exactly there
            c_raw[(long long)b * D + k] = v;
            y[(long long)b * D + k] = sg * sqrtf(fabsf(v) + 1e-10f) / n;
        }
    }
    if (qd == 0 && tid == 0) inv_norm[b] = 1.0f / n;
}

// backward, one launch: every workgroup of a sample recomputes t = <y, dy> over all D bins (fixed order), then
// dc = ((dy - y t) / n) / (2 sqrt(|c| + 1e-10)) for its quarter
__global__ __launch_bounds__(CBP_FT) void cbp_dc1_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                        const float* __restrict__ c_raw, const float* __restrict__ inv_norm,
                                                        float* __restrict__ dc, int D) {
    __shared__ float red[CBP_FT / 64];
    const int b = blockIdx.y, qd = blockIdx.x, tid = threadIdx.x;
    const int per = (D + CBP_FQ - 1) / CBP_FQ, k0 = qd * per, k1 = (k0 + per < D) ? k0 + per : D;
    const long long o = (long long)b * D;
    float tt = 0.f;
#pragma unroll
    for (int j = 0; j < CBP_FK; ++j) {                 // independent loads, fixed order of the adds
        const int k = tid + CBP_FT * j;
        tt += k < D ? y[o + k] * dy[o + k] : 0.f;
    }
    const float t = block_sum<CBP_FT / 64>(tt, red);
    const float in = inv_norm[b];
    for (int k = k0 + tid; k < k1; k += CBP_FT) {
        const float c = c_raw[o + k];
        const float du = (dy[o + k] - y[o + k] * t) * in;
        dc[o + k] = (c != 0.f) ? du / (2.0f * sqrtf(fabsf(c) + 1e-10f)) : 0.f;       // (c == 0: see cbp_dc_kernel)
    }
}

// backward of the finishing stage, same two-launch shape:  tp[b][blockIdx.x] = the workgroup's share of <y, dy>
__global__ __launch_bounds__(256) void cbp_dot_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ tp, int D) {
    __shared__ float red[4];
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const long long o = (long long)b * D;
    const float t = block_sum<4>(k < D ? y[o + k] * dy[o + k] : 0.f, red);
    if (threadIdx.x == 0) tp[(long long)b * gridDim.x + blockIdx.x] = t;
}

// dc = ((dy - y <y,dy>) / n) / (2 sqrt(|c| + 1e-10))
__global__ __launch_bounds__(256) void cbp_dc_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                     const float* __restrict__ c_raw, const float* __restrict__ inv_norm,
                                                     const float* __restrict__ tp, float* __restrict__ dc, int D) {
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const long long o = (long long)b * D;
    float t = 0.f;
    for (int q = 0; q < (int)gridDim.x; ++q) t += tp[(long long)b * gridDim.x + q];
    const float in = inv_norm[b];
    if (k < D) {
        const float c = c_raw[o + k];
        const float du = (dy[o + k] - y[o + k] * t) * in;
        // c == 0 exactly (a bin whose Gram entries are all exactly 0): torch's autograd of sign(c)*sqrt(|c|+1e-10)
        // gives 0 there (sign' = 0, abs'(0) = 0).  The reference's FFT route turns such a bin into round-off noise
        // and differentiates THAT (slope 1/(2 sqrt(|noise|)): 5e2 in fp32, 5e4 in fp64) - not reproducible by any
        // deterministic algorithm; see DESIGN.md "CBP zero bins".
        dc[o + k] = (c != 0.f) ? du / (2.0f * sqrtf(fabsf(c) + 1e-10f)) : 0.f;
    }
}

// A-operand loader of the backward GEMM: (dG + dG^T)[i][k]
struct LdCbpDG {
    CbpPlan pl;
    const float* dc;
    int C, D;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < C) {
            const float* d = dc + (long long)b * D;
            const int h1r = pl.h1[r], h2r = pl.h2[r];
            const float s1r = pl.s1[r], s2r = pl.s2[r];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = c + t;
                if (k < C) {
                    int ba = h1r + pl.h2[k]; if (ba >= D) ba -= D;
                    int bb = pl.h1[k] + h2r; if (bb >= D) bb -= D;
                    v[t] = s1r * pl.s2[k] * d[ba] + pl.s1[k] * s2r * d[bb];
                }
            }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};


// ------------------------------------------------------------------ the parts of CompactBilinearPooling Hawkeye's own
// CBCNN never takes (CBCNN.py:96-102 two DIFFERENT inputs; :127-130 sum_pool = False) - same count-sketch identity:
//   two inputs:       c[b,k]   = sum_{(i,j) -> k} s1_i s2_j (X1 X2^T)[b,i,j]          a binning of the CROSS Gram
//   per location:     c[b,p,k] = sum_{(i,j) -> k} s1_i s2_j x1[b,i,p] x2[b,j,p]       no sum over the map
// Plain kernels (fixed summation orders, no atomics); the signed square root and F.normalize behind them are left to
// the caller (hawkeye_amd/model/methods/CBCNN.py keeps the reference's own two lines for them).

// dG[b,i,j] = s1_i s2_j dc[b, (h1_i + h2_j) mod D]        dG [B, C1, C2]
__global__ __launch_bounds__(256) void cbp_unbin_kernel(const float* __restrict__ dc, const int* __restrict__ h1,
                                                        const int* __restrict__ h2, const float* __restrict__ s1,
                                                        const float* __restrict__ s2, float* __restrict__ dG, int C1, int C2, int D) {
    const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= C2) return;
    int k = h1[i] + h2[j];
    if (k >= D) k -= D;
    dG[((long long)b * C1 + i) * C2 + j] = s1[i] * s2[j] * dc[(long long)b * D + k];
}

// one workgroup per (location p, sample b): the two channel columns in LDS, thread t owns bins t, t + 256, ..; a bin's
// entries (i * C2 + j) in the plan's (i, j)-ascending order
__global__ __launch_bounds__(256) void cbp_loc_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          const int* __restrict__ off, const unsigned* __restrict__ ent,
                                                          float* __restrict__ c, int C1, int C2, int HW, int D) {
    HK_DYN_LDS(sm);                                    // x1 column [C1] | x2 column [C2]
    const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < C1; i += 256) sm[i] = x1[((long long)b * C1 + i) * HW + p];
    for (int j = tid; j < C2; j += 256) sm[C1 + j] = x2[((long long)b * C2 + j) * HW + p];
    __syncthreads();
    const float rc = 1.0f / (float)C2;
    float* cp = c + ((long long)b * HW + p) * D;
    for (int k = tid; k < D; k += 256) {
        float s = 0.f;
        for (int e = off[k]; e < off[k + 1]; ++e) {
            const unsigned u = ent[e];
            const int idx = (int)(u & 0x7fffffffu);
            const int i = (int)(((float)idx + 0.5f) * rc);          // idx / C2, exact for C1, C2 <= 1024 (entry point checks)
            const float v = sm[i] * sm[C1 + idx - i * C2];
            s += (u >> 31) ? -v : v;
        }
        cp[k] = s;
    }
}

// dx1[b,i,p] = s1_i sum_j s2_j dc[b,p,bin(i,j)] x2[b,j,p] ; dx2[b,j,p] = s2_j sum_i s1_i dc[b,p,bin(i,j)] x1[b,i,p]
// one workgroup per (p, b): dc[b,p,:], the two columns and the hashes in LDS; thread per channel, j (i) ascending
__global__ __launch_bounds__(256) void cbp_loc_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          const float* __restrict__ dc, const int* __restrict__ h1,
                                                          const int* __restrict__ h2, const float* __restrict__ s1,
                                                          const float* __restrict__ s2, float* __restrict__ dx1,
                                                          float* __restrict__ dx2, int C1, int C2, int HW, int D) {
    HK_DYN_LDS(sm);                                    // dc [D] | x1 s1 [C1] | x2 s2 [C2] | h1 [C1] | h2 [C2]
    const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float* sd = sm;
    float* a1 = sm + D;
    float* a2 = a1 + C1;
    int* g1 = reinterpret_cast<int*>(a2 + C2);
    int* g2 = g1 + C1;
    const float* dcp = dc + ((long long)b * HW + p) * D;
    for (int k = tid; k < D; k += 256) sd[k] = dcp[k];
    for (int i = tid; i < C1; i += 256) {
        a1[i] = s1[i] * x1[((long long)b * C1 + i) * HW + p];
        g1[i] = h1[i];
    }
    for (int j = tid; j < C2; j += 256) {
        a2[j] = s2[j] * x2[((long long)b * C2 + j) * HW + p];
        g2[j] = h2[j];
    }
    __syncthreads();
    if (dx1)
        for (int i = tid; i < C1; i += 256) {          // row i of dG against x2
            const int hi = g1[i];
            float acc = 0.f;
            for (int j = 0; j < C2; ++j) {
                int k = hi + g2[j];
                if (k >= D) k -= D;
                acc += sd[k] * a2[j];
            }
            dx1[((long long)b * C1 + i) * HW + p] = s1[i] * acc;
        }
    if (dx2)
        for (int j = tid; j < C2; j += 256) {          // column j of dG against x1
            const int hj = g2[j];
            float acc = 0.f;
            for (int i = 0; i < C1; ++i) {
                int k = g1[i] + hj;
                if (k >= D) k -= D;
                acc += sd[k] * a1[i];
            }
            dx2[((long long)b * C2 + j) * HW + p] = s2[j] * acc;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// input_dim1 != input_dim2 (CompactBilinearPooling(C1, C2, D), CBCNN.py:68-94): the same identity over a C1 x C2 cross
// Gram.  A plan of its own - hashes, signs and the CSR table bin -> (i * C2 + j, sign) - and the plain kernels above; the
// fused one-launch forms (hk_cbp_fwd / bwd) are for the square one-input case Hawkeye's CBCNN builds.
struct CbpRectPlan {
    const int* h1;       // [C1]
    const int* h2;       // [C2]
    const float* s1;
    const float* s2;
    const int* off;      // [D+1]
    const unsigned* ent; // [C1*C2]
};
static inline size_t cbp_rect_bytes(int C1, int C2, int D) {
    return 16 + 2 * cbp_align((size_t)C1 * 4) + 2 * cbp_align((size_t)C2 * 4) + cbp_align((size_t)(D + 1) * 4) +
           cbp_align((size_t)C1 * C2 * 4);
}
static inline CbpRectPlan cbp_rect_view(const void* plan, int C1, int C2, int D) {
    const char* p = (const char*)plan + 16;
    CbpRectPlan v;
    v.h1 = (const int*)p;            p += cbp_align((size_t)C1 * 4);
    v.h2 = (const int*)p;            p += cbp_align((size_t)C2 * 4);
    v.s1 = (const float*)p;          p += cbp_align((size_t)C1 * 4);
    v.s2 = (const float*)p;          p += cbp_align((size_t)C2 * 4);
    v.off = (const int*)p;           p += cbp_align((size_t)(D + 1) * 4);
    v.ent = (const unsigned*)p;
    return v;
}

}  // namespace hk

using namespace hk;

extern "C" int hk_cbp_bin_matrix(const float* G, const void* plan, float* c_raw, int B, int C, int D, hk_stream_t stream) {
    if (!G || !plan || !c_raw || B <= 0 || C <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (B > 65535) return HK_ERR_UNSUPPORTED;
    const CbpPlan pl = cbp_view(plan, C, D);
    hipLaunchKernelGGL(cbp_bin_kernel, dim3((D + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, G, pl.off, pl.ent, c_raw, C * C, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_unbin_matrix(const float* dc, const void* plan, float* dG, int B, int C, int D, hk_stream_t stream) {
    if (!dc || !plan || !dG || B <= 0 || C <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (B > 65535 || C > 65535) return HK_ERR_UNSUPPORTED;
    const CbpPlan pl = cbp_view(plan, C, D);
    hipLaunchKernelGGL(cbp_unbin_kernel, dim3((C + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, dc, pl.h1, pl.h2, pl.s1,
                       pl.s2, dG, C, C, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_loc_fwd(const float* x1, const float* x2, const void* plan, float* c, int B, int C, int HW, int D,
                              hk_stream_t stream) {
    if (!x1 || !x2 || !plan || !c || B <= 0 || C <= 0 || HW <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (C > 1024 || B > 65535) return HK_ERR_UNSUPPORTED;
    const CbpPlan pl = cbp_view(plan, C, D);
    hipLaunchKernelGGL(cbp_loc_fwd_kernel, dim3(HW, B), dim3(256), (size_t)2 * C * sizeof(float), (hipStream_t)stream, x1, x2,
                       pl.off, pl.ent, c, C, C, HW, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_loc_bwd(const float* x1, const float* x2, const float* dc, const void* plan, float* dx1, float* dx2, int B,
                              int C, int HW, int D, hk_stream_t stream) {
    if (!x1 || !x2 || !dc || !plan || (!dx1 && !dx2) || B <= 0 || C <= 0 || HW <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    const size_t lds = ((size_t)D + 4 * (size_t)C) * sizeof(float);
    if (lds > CBP_LOC_LDS_MAX || B > 65535) return HK_ERR_UNSUPPORTED;
    HK_ALLOW_BIG_LDS(cbp_loc_bwd_kernel, lds);          // D = 16000, C = 512 is 72 KB: above the default 64 KB, well inside 160
    const CbpPlan pl = cbp_view(plan, C, D);
    hipLaunchKernelGGL(cbp_loc_bwd_kernel, dim3(HW, B), dim3(256), lds, (hipStream_t)stream, x1, x2, dc, pl.h1, pl.h2, pl.s1, pl.s2,
                       dx1, dx2, C, C, HW, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// ------------------------------------------------------------------ input_dim1 != input_dim2
extern "C" size_t hk_cbp_rect_plan_bytes(int C1, int C2, int D) {
    if (C1 <= 0 || C2 <= 0 || D <= 0) return 0;
    return cbp_rect_bytes(C1, C2, D);
}

extern "C" int hk_cbp_rect_plan_build(const int32_t* h1, const float* s1, int C1, const int32_t* h2, const float* s2, int C2, int D,
                                      void* plan, hk_stream_t stream) {
    if (!h1 || !s1 || !h2 || !s2 || !plan || C1 <= 0 || C2 <= 0 || D <= 0 || (long long)C1 * C2 >= (1ll << 31)) return HK_ERR_BAD_ARG;
    for (int i = 0; i < C1; ++i)
        if (h1[i] < 0 || h1[i] >= D) return HK_ERR_BAD_ARG;                              // CBCNN.py:153
    for (int j = 0; j < C2; ++j)
        if (h2[j] < 0 || h2[j] >= D) return HK_ERR_BAD_ARG;
    std::vector<char> blob(cbp_rect_bytes(C1, C2, D), 0);
    ((int*)blob.data())[0] = C1;
    ((int*)blob.data())[1] = C2;
    ((int*)blob.data())[2] = D;
    char* p = blob.data() + 16;
    memcpy(p, h1, (size_t)C1 * 4); p += cbp_align((size_t)C1 * 4);
    memcpy(p, h2, (size_t)C2 * 4); p += cbp_align((size_t)C2 * 4);
    memcpy(p, s1, (size_t)C1 * 4); p += cbp_align((size_t)C1 * 4);
    memcpy(p, s2, (size_t)C2 * 4); p += cbp_align((size_t)C2 * 4);
    int* off = (int*)p; p += cbp_align((size_t)(D + 1) * 4);
    unsigned* ent = (unsigned*)p;
    std::vector<int> cnt(D, 0);
    for (int i = 0; i < C1; ++i)
        for (int j = 0; j < C2; ++j) cnt[(h1[i] + h2[j]) % D]++;
    off[0] = 0;
    for (int k = 0; k < D; ++k) off[k + 1] = off[k] + cnt[k];
    std::vector<int> cur(off, off + D);
    for (int i = 0; i < C1; ++i)       // (i,j) ascending inside each bin: fixed summation order
        for (int j = 0; j < C2; ++j) {
            const int k = (h1[i] + h2[j]) % D;
            const unsigned neg = (s1[i] * s2[j] < 0.f) ? 0x80000000u : 0u;
            ent[cur[k]++] = neg | (unsigned)(i * C2 + j);
        }
    hipError_t e = hipMemcpyAsync(plan, blob.data(), blob.size(), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);   // one-time setup: the host blob dies at return
    return e == hipSuccess ? HK_OK : (int)e;
}

extern "C" int hk_cbp_rect_bin_matrix(const float* G, const void* plan, float* c_raw, int B, int C1, int C2, int D, hk_stream_t stream) {
    if (!G || !plan || !c_raw || B <= 0 || C1 <= 0 || C2 <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (B > 65535) return HK_ERR_UNSUPPORTED;
    const CbpRectPlan pl = cbp_rect_view(plan, C1, C2, D);
    hipLaunchKernelGGL(cbp_bin_kernel, dim3((D + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, G, pl.off, pl.ent, c_raw, C1 * C2, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_rect_unbin_matrix(const float* dc, const void* plan, float* dG, int B, int C1, int C2, int D, hk_stream_t stream) {
    if (!dc || !plan || !dG || B <= 0 || C1 <= 0 || C2 <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (B > 65535 || C1 > 65535) return HK_ERR_UNSUPPORTED;
    const CbpRectPlan pl = cbp_rect_view(plan, C1, C2, D);
    hipLaunchKernelGGL(cbp_unbin_kernel, dim3((C2 + 255) / 256, C1, B), dim3(256), 0, (hipStream_t)stream, dc, pl.h1, pl.h2, pl.s1,
                       pl.s2, dG, C1, C2, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_rect_loc_fwd(const float* x1, const float* x2, const void* plan, float* c, int B, int C1, int C2, int HW, int D,
                                   hk_stream_t stream) {
    if (!x1 || !x2 || !plan || !c || B <= 0 || C1 <= 0 || C2 <= 0 || HW <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (C1 > 1024 || C2 > 1024 || B > 65535) return HK_ERR_UNSUPPORTED;
    const CbpRectPlan pl = cbp_rect_view(plan, C1, C2, D);
    hipLaunchKernelGGL(cbp_loc_fwd_kernel, dim3(HW, B), dim3(256), (size_t)(C1 + C2) * sizeof(float), (hipStream_t)stream, x1, x2,
                       pl.off, pl.ent, c, C1, C2, HW, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_rect_loc_bwd(const float* x1, const float* x2, const float* dc, const void* plan, float* dx1, float* dx2, int B,
                                   int C1, int C2, int HW, int D, hk_stream_t stream) {
    if (!x1 || !x2 || !dc || !plan || (!dx1 && !dx2) || B <= 0 || C1 <= 0 || C2 <= 0 || HW <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    const size_t lds = ((size_t)D + 2 * ((size_t)C1 + (size_t)C2)) * sizeof(float);
    if (lds > CBP_LOC_LDS_MAX || B > 65535) return HK_ERR_UNSUPPORTED;
    HK_ALLOW_BIG_LDS(cbp_loc_bwd_kernel, lds);
    const CbpRectPlan pl = cbp_rect_view(plan, C1, C2, D);
    hipLaunchKernelGGL(cbp_loc_bwd_kernel, dim3(HW, B), dim3(256), lds, (hipStream_t)stream, x1, x2, dc, pl.h1, pl.h2, pl.s1, pl.s2,
                       dx1, dx2, C1, C2, HW, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

namespace hk {
}  // namespace hk

using namespace hk;

extern "C" size_t hk_cbp_plan_bytes(int C, int D) {
    const size_t nb = (size_t)(C / 64);
    return 16 + 4 * cbp_align((size_t)C * 4) + cbp_align((size_t)(D + 1) * 4) + cbp_align((size_t)C * C * 4) +
           cbp_align((size_t)C * 4) + cbp_align((size_t)(C + 1) * 4) + cbp_align((size_t)C * 4) +
           (cbp_has_lists(C, D) ? nb * nb * CBF_LLEN * sizeof(unsigned) + 16 : 0);
}

extern "C" int hk_cbp_plan_build(const int32_t* h1, const float* s1, const int32_t* h2, const float* s2, int C, int D,
                                 void* plan, hk_stream_t stream) {
    if (!h1 || !s1 || !h2 || !s2 || !plan || C <= 0 || D <= 0 || (long long)C * C >= (1ll << 31)) return HK_ERR_BAD_ARG;
    for (int i = 0; i < C; ++i)
        if (h1[i] < 0 || h1[i] >= D || h2[i] < 0 || h2[i] >= D) return HK_ERR_BAD_ARG;   // CBCNN.py:153
    std::vector<char> blob(hk_cbp_plan_bytes(C, D), 0);
    int fused_ok = 0;
    ((int*)blob.data())[0] = C;
    ((int*)blob.data())[1] = D;
    char* p = blob.data() + 16;
    memcpy(p, h1, (size_t)C * 4); p += cbp_align((size_t)C * 4);
    memcpy(p, h2, (size_t)C * 4); p += cbp_align((size_t)C * 4);
    memcpy(p, s1, (size_t)C * 4); p += cbp_align((size_t)C * 4);
    memcpy(p, s2, (size_t)C * 4); p += cbp_align((size_t)C * 4);
    int* off = (int*)p; p += cbp_align((size_t)(D + 1) * 4);
    unsigned* ent = (unsigned*)p;
    std::vector<int> cnt(D, 0);
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) cnt[(h1[i] + h2[j]) % D]++;
    off[0] = 0;
    for (int k = 0; k < D; ++k) off[k + 1] = off[k] + cnt[k];
    std::vector<int> cur(off, off + D);
    for (int i = 0; i < C; ++i)        // (i,j) ascending inside each bin: fixed summation order
        for (int j = 0; j < C; ++j) {
            const int k = (h1[i] + h2[j]) % D;
            const unsigned neg = (s1[i] * s2[j] < 0.f) ? 0x80000000u : 0u;
            ent[cur[k]++] = neg | (unsigned)(i * C + j);
        }
    // inverse of h2 over its non-empty bins (row-sketch kernel), channels ascending inside a bin
    {
        int* nzb = (int*)((char*)ent + cbp_align((size_t)C * C * 4));
        int* nzo = (int*)((char*)nzb + cbp_align((size_t)C * 4));
        unsigned* nzj = (unsigned*)((char*)nzo + cbp_align((size_t)(C + 1) * 4));
        std::vector<std::vector<int>> inv(D);
        for (int j = 0; j < C; ++j) inv[h2[j]].push_back(j);
        // slots ordered by channel count, descending (ties: bin ascending): the few bins that two or more channels hash
        // to (~22 of ~490 at C = 512, D = 6000) all land in the first wave of the binning kernels; every other wave has
        // exactly one channel per bin and takes the single-gather path of cbp_rowscatter_kernel
        std::vector<int> order;
        for (int mbin = 0; mbin < D; ++mbin)
            if (!inv[mbin].empty()) order.push_back(mbin);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return inv[a].size() > inv[b].size(); });
        int t = 0, e = 0;
        nzo[0] = 0;
        for (int mbin : order) {
            nzb[t] = mbin;
            for (int j : inv[mbin]) nzj[e++] = (unsigned)j | (s2[j] < 0.f ? 0x80000000u : 0u);
            nzo[++t] = e;
        }
        ((int*)blob.data())[2] = t;
        int emax = 0;
        for (int mbin = 0; mbin < D; ++mbin) emax = (int)inv[mbin].size() > emax ? (int)inv[mbin].size() : emax;
        ((int*)blob.data())[3] = emax;
        // tile lists of the fused forward + one word behind them: 1 = these hashes allow it
        if (cbp_has_lists(C, D)) {
            unsigned* lists = (unsigned*)((char*)nzj + cbp_align((size_t)C * 4));
            std::vector<unsigned> L;
            const int ok = cbf_build_lists(h1, s1, h2, s2, C, D, L);
            const size_t nw = (size_t)(C / 64) * (C / 64) * CBF_LLEN;
            if (ok) memcpy(lists, L.data(), nw * sizeof(unsigned));
            lists[nw] = (unsigned)ok;
            fused_ok = ok;
        }
    }
    hipError_t e = hipMemcpyAsync(plan, blob.data(), blob.size(), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);   // one-time setup: the host blob dies at return
    if (e != hipSuccess) return (int)e;
    plan_note(plan, fused_ok);
    return HK_OK;
}

extern "C" int hk_cbp_plan_destroy(const void* plan) {
    if (!plan) return HK_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_fused.erase(plan);
    return HK_OK;
}

extern "C" size_t hk_cbp_ws_bytes(int B, int C, int HW, int D) {
    (void)HW;
    const size_t fin = (size_t)B * ((D + 255) / 256) * sizeof(float);            // per-workgroup sums of the finishing stage
    const size_t g = (size_t)B * C * C * sizeof(float) + (size_t)B * ((C + 63) / 64) * D * sizeof(float) + fin;  // G + row-chunk partials
    const size_t dc = (size_t)B * D * sizeof(float) + fin;
    return (g > dc ? g : dc) + 256;
}

extern "C" int hk_cbp_fwd(const float* x, const void* plan, float* y, float* c_raw, float* inv_norm, int B, int C,
                          int HW, int D, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !plan || !y || !c_raw || !inv_norm || B <= 0 || C <= 0 || HW <= 0 || D <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_cbp_ws_bytes(B, C, HW, D)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const CbpPlan pl = cbp_view(plan, C, D);
    float* G = (float*)ws;
    const int nchunk = (C + 63) / 64;
    float* part = G + (long long)B * C * C;
    // Binning stage, measured at C=512, D=6000 (whole hk_cbp_fwd, HIP events, BENCH_r01): row-scatter 89.0 us @B=64 /
    // 76.4 @B=16, row-sketch 108.3 / 85.7, CSR gather 176.7 / 79.9.  Row-scatter (bins in LDS) is the default wherever
    // its bins fit the LDS; otherwise the row-sketch kernel once B * C/64 workgroups fill the 256 CUs, else the CSR
    // gather.  tuning().cbp_bin forces one (0 row-sketch, 1 CSR gather, 2 row-scatter); all produce identical partials.
    const int bin = tuning().cbp_bin;                        // -1: automatic
    const int nq8 = ((D + 255) / 256 + 7) / 8;               // 8-bin groups per thread
    const bool rowsketch = C <= 512 && C % 4 == 0 && nq8 <= 4 && D >= 1024 * nq8 && pl.emax <= CBP_EMAX &&
                           (bin >= 0 ? bin == 0 : B * nchunk >= 256);
    const bool scatter = (bin < 0 || bin == 2) && C <= 512 && C % 4 == 0 && pl.emax <= CBP_EMAX &&
                         rowscatter_lds(C, D) <= 150 * 1024;
    // Fused Gram + binning (hk_cbp_fused.h): the default wherever the plan carries its tile lists and the shape is one
    // of the panel kernels'; cbp_bin = 3 forces it, 0 / 1 / 2 force the unfused binning kernels.
    int npart = nchunk;                                      // partial bin vectors per sample handed to cbp_partsum_kernel
    bool fused = false;
    const int fok = plan_fused_ok(plan);
    if ((bin < 0 || bin == 3 || bin == 4) && !force_generic() && pl.lists && fok) {          // 4: never pair the steps
        const CbfSchedule sch = cbf_schedule(C / 64, tuning().sched_b > 0 ? tuning().sched_b : B);
        if ((size_t)B * sch.nitems * D * sizeof(float) + (size_t)B * ((D + 255) / 256) * sizeof(float) <= ws_bytes) {
            const int rc = cbf_launch(x, pl.lists, (float*)ws, B, C, HW, D, sch, fok == 2 && bin != 4, st);
            if (rc == HK_OK) { fused = true; npart = sch.nitems; part = (float*)ws; }
            else if (rc != HK_ERR_UNSUPPORTED) return rc;
        }
    }
    // Gram + binning of samples b0 .. b0 + nb - 1 on queue q
    auto gram_and_bin = [&](int b0, int nb, hipStream_t q) -> int {
        const float* xh = x + (long long)b0 * C * HW;
        float* Gh = G + (long long)b0 * C * C;
        float* ph = part + (long long)b0 * nchunk * D;
        int rc = force_generic() ? HK_ERR_UNSUPPORTED : gram_fast_raw(xh, nullptr, 1.0f, Gh, nb, C, HW, q);
        if (rc == HK_ERR_UNSUPPORTED) {                                                            // raw Gram, no 1/HW
            const LdPlain xa = make_plain(xh, (long long)C * HW, HW, C, HW);
            const EpAffine ep = make_affine(Gh, (long long)C * C, C, 1.0f, nullptr, 0.f, 0.f);
            rc = bgemm_launch<true, true>(xa, xa, ep, C, C, HW, nb, q);
        }
        if (rc != HK_OK) return rc;
        if (scatter)
            return C <= 256 ? rowscatter_launch<1>(Gh, pl, ph, nb, C, D, nchunk, q)
                            : rowscatter_launch<2>(Gh, pl, ph, nb, C, D, nchunk, q);
        if (rowsketch) {
            const bool one = C <= 256;
            switch (nq8) {
                case 1: return one ? rowsketch_launch<1, 1>(Gh, pl, ph, nb, C, D, nchunk, q) : rowsketch_launch<2, 1>(Gh, pl, ph, nb, C, D, nchunk, q);
                case 2: return one ? rowsketch_launch<1, 2>(Gh, pl, ph, nb, C, D, nchunk, q) : rowsketch_launch<2, 2>(Gh, pl, ph, nb, C, D, nchunk, q);
                case 3: return one ? rowsketch_launch<1, 3>(Gh, pl, ph, nb, C, D, nchunk, q) : rowsketch_launch<2, 3>(Gh, pl, ph, nb, C, D, nchunk, q);
                case 4: return one ? rowsketch_launch<1, 4>(Gh, pl, ph, nb, C, D, nchunk, q) : rowsketch_launch<2, 4>(Gh, pl, ph, nb, C, D, nchunk, q);
            }
            return HK_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(cbp_bin_kernel, dim3((D + 3) / 4, nb), dim3(256), 0, q, (const float*)Gh, pl.off, pl.ent,
                           c_raw + (long long)b0 * D, C * C, D);
        HK_LAUNCH_CHECK();
        return HK_OK;
    };
    // (Running the two halves of the batch on two HIP queues, so that one half's binning overlaps the other's Gram - what
    //  pays for the Newton-Schulz chain - was measured here and is NOT done: 121.9 us against 81.5 us on one queue at
    //  B = 64; both kernels want most of a CU's LDS and each half-batch Gram fills only half the CUs.)
    if (!fused) {
        const int rc = gram_and_bin(0, B, st);
        if (rc != HK_OK) return rc;
    }
    if ((fused || rowsketch || scatter) && D <= CBP_FK * CBP_FT) {
        hipLaunchKernelGGL(cbp_finish_kernel, dim3(CBP_FQ, B), dim3(CBP_FT), 0, st, (const float*)part, c_raw, y, inv_norm, D,
                           npart);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    float* ssq = part + (long long)B * npart * D;
    const dim3 fgrid((D + 255) / 256, B);
    if (fused || rowsketch || scatter)
        hipLaunchKernelGGL(cbp_partsum_kernel, fgrid, dim3(256), 0, st, (const float*)part, c_raw, ssq, D, npart);
    else
        hipLaunchKernelGGL(cbp_partsum_kernel, fgrid, dim3(256), 0, st, (const float*)c_raw, c_raw, ssq, D, 1);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(cbp_norm_kernel, fgrid, dim3(256), 0, st, (const float*)c_raw, (const float*)ssq, y, inv_norm, D);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_cbp_bwd(const float* x, const void* plan, const float* y, const float* c_raw, const float* inv_norm,
                          const float* dy, float* dx, int B, int C, int HW, int D, void* ws, size_t ws_bytes,
                          hk_stream_t stream) {
    if (!x || !plan || !y || !c_raw || !inv_norm || !dy || !dx || B <= 0 || C <= 0 || HW <= 0 || D <= 0)
        return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_cbp_ws_bytes(B, C, HW, D)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!force_generic()) {          // one launch: dc is formed inside the GEMM kernel (hk_bwd3c.h)
        const CbpPlan pv = cbp_view(plan, C, D);
        const int rc = cbp_fast_bwd_fused(x, pv.h1, pv.h2, pv.s1, pv.s2, y, dy, c_raw, inv_norm, D, dx, B, C, HW, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    float* dc = (float*)ws;
    float* tp = dc + (long long)B * D;
    const dim3 fgrid((D + 255) / 256, B);
    if (D <= CBP_FK * CBP_FT) {
        hipLaunchKernelGGL(cbp_dc1_kernel, dim3(CBP_FQ, B), dim3(CBP_FT), 0, st, y, dy, c_raw, inv_norm, dc, D);
        HK_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(cbp_dot_kernel, fgrid, dim3(256), 0, st, y, dy, tp, D);
        HK_LAUNCH_CHECK();
        hipLaunchKernelGGL(cbp_dc_kernel, fgrid, dim3(256), 0, st, y, dy, c_raw, inv_norm, (const float*)tp, dc, D);
        HK_LAUNCH_CHECK();
    }
    if (!force_generic()) {
        const CbpPlan pv = cbp_view(plan, C, D);
        const int rc = cbp_fast_bwd(x, pv.h1, pv.h2, pv.s1, pv.s2, dc, D, dx, B, C, HW, st);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    LdCbpDG la;
    la.pl = cbp_view(plan, C, D);
    la.dc = dc; la.C = C; la.D = D;
    const LdPlain xb = make_plain(x, (long long)C * HW, HW, C, HW);
    const EpAffine ep = make_affine(dx, (long long)C * HW, HW, 1.0f, nullptr, 0.f, 0.f);
    return bgemm_launch<true, false>(la, xb, ep, C, HW, C, B, st);
}

