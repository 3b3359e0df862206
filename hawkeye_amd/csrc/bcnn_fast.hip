// Specialised BCNN bilinear-pooling kernels for the shapes the configs use
// (C % 64 == 0, HW in {196, 144, 100, 64}: VGG/ResNet 14x14 .. 8x8 maps).  Anything
// else takes the generic hk::bgemm_kernel path in bcnn_pool.hip.
//
// Forward  (bcnn_gram_panel_kernel): G is symmetric, so only the 64x64 tiles (I,J),
//   J >= I, are computed and each is written twice (direct + mirrored).  A workgroup
//   owns a row-block I (or the balanced pair {p, nb-1-p}: nb+1 tiles each, so that
//   B=64 x C=512 is exactly 256 equal workgroups = one per CU), keeps the 64 x HW
//   A-panel resident in LDS (50 KB, full K) and streams the J-panels through two more
//   LDS buffers: the next panel is fetched into registers before the MFMA loop and
//   written after it, one barrier per tile.  X[b] row-panels are contiguous in HBM and
//   HW = 196 = 4 * 49 floats per row gives a conflict-free ds_read_b128 pitch with no
//   padding.  4 waves = 2x2 sub-tiles of 32x32 on v_mfma_f32_32x32x2_f32; the
//   sqrt / normalise / store epilogue of tile t is interleaved, register by register,
//   with the MFMAs of tile t+1 so the matrix pipe does not drain between tiles.
//
// Backward (bcnn_bwd_panel_kernel): dX[I-rows] = sum_K P(I,K) X(K) with
//   P = (dy + dy^T) / (2 n^2 M y) built per 64x64 tile in LDS from coalesced reads of
//   y(I,K), dy(I,K) and dy(K,I) (transposed through LDS).  N = HW = 196 is covered by
//   13 tiles of 16 columns (v_mfma_f32_16x16x4_f32: 94 % useful instead of 77 % with
//   64-wide tiles); each of the 4 waves owns 16 rows x 208 columns (13 accumulators).
//   The next K-block's operands are prefetched into registers during the MFMA loop;
//   68 KB of LDS per workgroup -> 2 workgroups per CU overlap each other's phases.
#include <cstdlib>
#include "hk_common.h"
#include "hk_bwd3.h"
#include "hk_bwd3c.h"
#include "hk_gram_tile.h"

namespace hk {


// ----------------------------------------------------------------------------- forward
// (GramEpi and gram_tile: hk_gram_tile.h)
// Covariance (CENTER): centre the 64 x HW row panel `p` (LDS) in place and write the 64 row means to mu_out.  Sixteen
// lanes per row (a wave takes four rows at a time, four times): lane q of row r owns the 16-byte slots
// ((q - r R4) mod 16) + 16 k of the row, k = 0, 1, .. - ROTATED by the row's own start slot, so that the sixteen lanes a
// ds_read_b128 / the eight lanes a ds_write_b128 serves together always fall on different slots of the 256-byte bank row,
// whatever the row pitch (round 4: four lanes per row in place, 23.6 % of the kernel's LDS cycles were conflict cycles).
// Each lane adds its slots in k order, then a four-step butterfly over the row's sixteen lanes: a fixed order, so the
// means do not depend on which workgroup computes them.  Called between two barriers.
template <int HW>
__device__ __forceinline__ void center_panel(float* p, float* __restrict__ mu_out, int tid) {
    constexpr int R4 = HW / 4;                       // 16-byte slots per row
    constexpr int NK = (R4 + 15) / 16;
    const int wave = tid >> 6, lane = tid & 63, q = lane & 15, rr = lane >> 4;
    f32x4 v[4][NK];
    float s[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = 16 * wave + 4 * pass + rr;
        const int q0 = (q - r * R4) & 15;
        const f32x4* row = reinterpret_cast<const f32x4*>(p + r * HW);
        s[pass] = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int f = q0 + 16 * k;
            v[pass][k] = f < R4 ? row[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s[pass] += (v[pass][k][0] + v[pass][k][1]) + (v[pass][k][2] + v[pass][k][3]);
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = 16 * wave + 4 * pass + rr;
        const int q0 = (q - r * R4) & 15;
        float t = s[pass];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        const float m = t / (float)HW;
        f32x4* row = reinterpret_cast<f32x4*>(p + r * HW);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int f = q0 + 16 * k;
            if (f < R4) row[f] = v[pass][k] - m;
        }
        if (q == 0) mu_out[r] = m;
    }
}

// CENTER (covariance, MPNCOV.py:115-117: cov = X Ibar X^T = (1/M) (X - mu 1^T) X^T): the row panel (the A operand) is
// centred in LDS by the workgroup that owns it - which also writes its 64 channel means to mu for the backward - and
// the column panels stay RAW: sum_k (x_ik - mu_i)(x_jk - mu_j) = sum_k (x_ik - mu_i) x_jk because a centred row sums
// to zero, which is exactly the reference's (X Ibar) X^T.  No separate row-mean kernel, no mean look-ups while
// staging.  (Diagonal tiles read the centred panel on both sides: the same number to rounding.)
// NormSrc (MODE 0, hk_bcnn_pool_fwd): instead of reading inv_norm[b] the workgroup forms it from the 64-channel-group
// partial column sums of bcnn_colsum_partial4_kernel - the arithmetic of bcnn_norm_finalize_kernel (bcnn_pool.hip), bit
// for bit, behind its first panel loads - and the workgroup of row block 0 writes colsum / inv_norm for the backward:
// the finalize launch (4.5 us between two 6 us / 42 us kernels) is gone.
constexpr int SSQ_STRIDE = 64;         // MODE 2: partial sums per image in the caller's buffer (= SS_CHUNKS of bcnn_pool.hip)
struct GramNormSrc {
    const float* part;      // [B][G][HW], nullable: then inv_norm is read - or, `direct`, the sums are formed here from x
    float* colsum;          // [B][HW]
    float* inv_out;         // [B]
    int G;
    int direct;             // 1: no partials from another launch - the workgroup adds up the sample's columns itself
};

// NormSrc.direct (round 5: hk_bcnn_pool_fwd in ONE launch).  Every workgroup of a sample forms the sample's column sums
// itself from x - 4 C HW bytes, read from HBM once per sample (its workgroups run side by side on one XCD: the others
// hit the L2) - with exactly the arithmetic of bcnn_colsum_partial4_kernel + bcnn_norm_finalize_kernel (bcnn_pool.hip):
// per 64-channel group, thread (r, q) adds rows r, r + R, .. of column quad q in order, the R phase sums of a quad
// are added in phase order, the groups in group order - so colsum, 1 / |z| and y are the bits of the two-launch route.
// The loads of NG groups are in flight together (the A panel's loads were issued before them and are consumed first);
// `scr` = the two free panel buffers.  46.0 us against 48.5 in two launches and 40.1 for the Gram kernel alone: what
// it costs is L2 BANDWIDTH - every workgroup reads its whole sample, 103 MB chip-wide - not latency: the same sums
// streamed behind the MFMA steps of the first tile (40 loads per thread in flight, the head of the stream issued in the
// prologue) measured 45.95 us and are not kept.
// the phase sums red [nb][256] (16-byte pieces, in LDS) -> colsum, 1 / |z|: bcnn_colsum_partial4_kernel's phase order, then
// bcnn_norm_finalize_kernel's group order and norm.  Starts with a barrier (the phase sums are published).
template <int HW>
__device__ __forceinline__ float gram_norm_from_phase_sums(const f32x4* red, int C, int nb, float* red4, const GramNormSrc& ns,
                                                           int b, bool writer, int tid) {
    constexpr int Q = HW / 4, R = 256 / Q;
    __syncthreads();
    float ssq = 0.f;
    for (int hw = tid; hw < HW; hw += 256) {
        const int qq = hw >> 2, e = hw & 3;
        float cs = 0.f;
        for (int g = 0; g < nb; ++g) {
            float t = red[g * 256 + qq][e];
            for (int k = 1; k < R; ++k) t += red[g * 256 + qq + k * Q][e];
            cs += t;
        }
        if (writer) ns.colsum[(long long)b * HW + hw] = cs;
        ssq += cs * cs;
    }
    const float tot = block_sum<4>(ssq, red4);
    const float n2 = tot / (float)HW + (float)C * (float)C * 1e-5f;
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    if (writer && tid == 0) ns.inv_out[b] = inv;
    return inv;
}

template <int HW>
__device__ __forceinline__ float gram_direct_norm(const float* __restrict__ xb, int C, int nb, float* scr, float* red4,
                                                  const GramNormSrc& ns, int b, bool writer, int tid) {
    constexpr int Q = HW / 4, R = 256 / Q, NRW = (64 + R - 1) / R, NG = 4;
    static_assert(HW % 4 == 0 && Q <= 64, "column quads");
    const int q = tid % Q, r = tid / Q;
    const bool act = r < R;
    f32x4* red = reinterpret_cast<f32x4*>(scr);                      // [nb][256] phase sums
    for (int g0 = 0; g0 < nb; g0 += NG) {
        f32x4 v[NG][NRW];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = g0 + gi < nb ? g0 + gi : nb - 1;
            const f32x4* xg = reinterpret_cast<const f32x4*>(xb + (long long)g * 64 * HW);
#pragma unroll
            for (int i = 0; i < NRW; ++i) {
                int c = (act ? r : 0) + i * R;
                c = c < 64 ? c : 63;
                v[gi][i] = xg[c * Q + q];
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NRW; ++i)
                if (act && r + i * R < 64) sacc += v[gi][i];
            if (g0 + gi < nb) red[(g0 + gi) * 256 + tid] = sacc;
        }
    }
    return gram_norm_from_phase_sums<HW>(red, C, nb, red4, ns, b, writer, tid);
}

template <int HW, int MODE, bool CENTER>
__global__ __launch_bounds__(256, 1) void bcnn_gram_panel_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ inv_norm,
                                                                 float* __restrict__ y, int C, int nb, int B,
                                                                 int pair_mode, float* __restrict__ mu,
                                                                 float alpha, const GramNormSrc ns) {
    constexpr int PANEL = 64 * HW;
    constexpr int N4 = PANEL / 4;
    constexpr int NST = (N4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[3 * PANEL];
    __shared__ __attribute__((aligned(16))) float epi_stage[4][512];      // 2 KB per wave: the epilogue's turn-table (GramEpi)

    int b, w;
    const int per = pair_mode ? (nb + 1) / 2 : nb;
    if (!xcd_map(blockIdx.x, B, per, b, w)) return;
    const int rb0 = w;
    int rb1 = pair_mode ? nb - 1 - w : -1;
    if (rb1 == rb0) rb1 = -1;
    const int nrb = rb1 >= 0 ? 2 : 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const float* xb = x + (long long)b * C * HW;

    GramEpi<MODE> ep;
    ep.yb = y + (long long)b * C * C;
    ep.C = C;
    ep.inv = MODE == 0 ? ((ns.part || ns.direct) ? 0.f : inv_norm[b]) : alpha;
    float* mub = CENTER ? mu + (long long)b * C : nullptr;
    ep.inv_m = 1.0f / (float)HW;
    ep.l31 = l31;
    ep.lh = lh;
    ep.i0 = ep.j0 = ep.offdiag = 0;
    ep.ss = 0.f;
    ep.stg = epi_stage[wave];
    ep.t0 = ep.t1 = (f32x4){0.f, 0.f, 0.f, 0.f};

    {   // first A panel: all loads in flight at once, then the LDS writes
        const f32x4* src = reinterpret_cast<const f32x4*>(xb + (long long)rb0 * PANEL);
        f32x4 st0[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int f = tid + 256 * u, fc = f < N4 ? f : N4 - 1;
            st0[u] = src[fc];
        }
        if (MODE == 0 && ns.direct == 1) {                // (uniform) one launch: the column sums from x itself
            __shared__ float redd[4];
            f32x4* dst0 = reinterpret_cast<f32x4*>(lds);
            // (the A panel first - its loads are the oldest in flight - so that the tile loop's operand is in place while
            //  the column loads are still arriving)
            const float inv_d = gram_direct_norm<HW>(xb, C, nb, lds + PANEL, redd, ns, b, w == 0, tid);
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int f = tid + 256 * u;
                if (f < N4) dst0[f] = st0[u];
            }
            ep.inv = inv_d;
        } else if (MODE == 0 && ns.part) {                // (uniform) the sample's norm, while the panel loads are in flight
            __shared__ float redn[4];
            const float* pp = ns.part + (long long)b * ns.G * HW;
            float ssq = 0.f;
            for (int hw = tid; hw < HW; hw += 256) {
                float cs = 0.f;
                for (int gq = 0; gq < ns.G; ++gq) cs += pp[(long long)gq * HW + hw];
                if (w == 0) ns.colsum[(long long)b * HW + hw] = cs;
                ssq += cs * cs;
            }
            const float tot = block_sum<4>(ssq, redn);
            const float n2 = tot / (float)HW + (float)C * (float)C * 1e-5f;
            ep.inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
            if (w == 0 && tid == 0) ns.inv_out[b] = ep.inv;
        }
        if (!(MODE == 0 && ns.direct == 1)) {
            f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int f = tid + 256 * u;
                if (f < N4) dst[f] = st0[u];
            }
        }
    }
    __syncthreads();
    if (CENTER) {
        center_panel<HW>(lds, mub + rb0 * 64, tid);
        __syncthreads();
    }

    int a_idx = 0, b_idx = 0;
    // HW = 64 (row pitch 256 B: every row starts in the same LDS bank): this wave's A rows are held in registers for the
    // whole row block (hk_gram_tile.h, 28.6 -> 23.4 us at B = 64, C = 512); at the other sizes it measured level
    // (HW = 100, 144) or 0.6 us slower (196), and they keep reading A from the panel.
    constexpr bool AREG = HW == 64;
    GramAReg<AREG ? HW : 8> areg;
    if (AREG) areg.load(lds + (wm * 32 + l31) * HW + 4 * lh, lh);
    f32x16 prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0.f;

    // Column blocks of a row block.  Paired rows (pair_mode 1: rows w and nb - 1 - w, nb + 1 tiles per workgroup) walk
    // J = I .. nb - 1.  A single row (pair_mode 0) walks CYCLICALLY, J = I, I + 1, .. (mod nb), nb / 2 + 1 columns for
    // the first half of the rows and nb / 2 for the second ((nb + 1) / 2 for odd nb): every unordered pair {I, J} is
    // produced once (tile (I, J) and its mirror are both written) and no workgroup has more than nb / 2 + 1 tiles -
    // the triangular walk gave row 0 nb tiles and row nb - 1 one.
    const int cyc = pair_mode ? 0 : ((nb & 1) ? (nb + 1) / 2 : (rb0 < nb / 2 ? nb / 2 + 1 : nb / 2));
    // One tile of the walk.  FIRST (compile time): the workgroup's first tile - no previous tile's epilogue to interleave;
    // peeled out of the loop below, so that every other tile is the HASPREV form without a branch.
    auto tile_step = [&](auto first_tag, int ri, int I, int cnt, int tt) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int J = I + tt;
        if (J >= nb) J -= nb;
        int next_blk = -1;
        bool newrow = false;
        if (tt + 1 < cnt) next_blk = (J + 1 < nb) ? J + 1 : 0;
        else if (ri + 1 < nrb) { next_blk = rb1; newrow = true; }
        const int n_idx = (a_idx == b_idx) ? (a_idx + 1) % 3 : 3 - a_idx - b_idx;

        f32x4 st[NST];
        {   // unconditional (index-clamped) loads keep st[] in registers; on the last tile they re-read a panel
            const int lb = next_blk >= 0 ? next_blk : J;
            const f32x4* src = reinterpret_cast<const f32x4*>(xb + (long long)lb * PANEL);
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int f = tid + 256 * u, fc = f < N4 ? f : N4 - 1;
                st[u] = src[fc];
            }
        }

        const float* Ap = lds + a_idx * PANEL + (wm * 32 + l31) * HW + 4 * lh;
        const float* Bp = lds + b_idx * PANEL + (wn * 32 + l31) * HW + 4 * lh;
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        f32x4* dst = reinterpret_cast<f32x4*>(lds + n_idx * PANEL);
        if constexpr (AREG) gram_tile_ra<HW, !FIRST, NST>(areg, Bp, acc0, acc1, prev, ep, lh, st, dst, next_blk >= 0, tid);
        else gram_tile<HW, !FIRST, NST>(Ap, Bp, acc0, acc1, prev, ep, lh, st, dst, next_blk >= 0, tid);
        prev = acc0 + acc1;
        ep.i0 = I * 64 + wm * 32;
        ep.j0 = J * 64 + wn * 32;
        ep.offdiag = (I != J);
        // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for every epilogue store of this tile
        HK_LDS_BARRIER();
        if (next_blk >= 0) {
            if (newrow) {
                a_idx = n_idx; b_idx = n_idx;
                if (CENTER) {                                         // the second row block of the pair: its panel just landed
                    center_panel<HW>(lds + n_idx * PANEL, mub + rb1 * 64, tid);
                    HK_LDS_BARRIER();
                }
                if (AREG) areg.load(lds + a_idx * PANEL + (wm * 32 + l31) * HW + 4 * lh, lh);
            } else {
                b_idx = n_idx;
            }
        }
    };
    for (int ri = 0; ri < nrb; ++ri) {
        const int I = ri == 0 ? rb0 : rb1;
        const int cnt = pair_mode ? nb - I : cyc;
        int tt = 0;
        if (ri == 0) {
            tile_step(std::true_type{}, ri, I, cnt, 0);
            tt = 1;
        }
        for (; tt < cnt; ++tt) tile_step(std::false_type{}, ri, I, cnt, tt);
    }
#pragma unroll
    for (int s_ = 0; s_ < GramEpi<MODE>::NSTEP; ++s_) ep.step(prev, s_);
    if (MODE == 2) {        // this workgroup's share of |u|^2 (fixed order) -> mu[b][w]  (row stride SSQ_STRIDE)
        __shared__ float reds[4];
        const float tot = block_sum<4>(ep.ss, reds);
        if (tid == 0) mu[(long long)b * SSQ_STRIDE + w] = tot;
    }
}

// ----------------------------------------------------------------------------- backward
// this thread's 4 float4 of the 64x64 tiles (I,kb) of y, dy and (kb,I) of dy (row = f >> 4, col4 = (f & 15) * 4,
// f = tid + 256 u) and its share of the 64 x HW block kb of X
// MODE 0 BCNN: P = (dy + dy^T) / y * coef          (ry, rd, rt loaded)
// MODE 1 COV : P = (g + g^T) / M, X centred          (rd, rt loaded; rx -= mu[k])
// MODE 2 CBP : P = dG + dG^T gathered from dc         (rd computed; nothing to transpose)

template <int HW, int NSX, int MODE, bool LOAD_T = true>
__device__ __forceinline__ void bwd_load(f32x4 (&ry)[4], f32x4 (&rd)[4], f32x4 (&rt)[4], f32x4 (&rx)[NSX],
                                         const float* __restrict__ y, const float* __restrict__ dy,
                                         const float* __restrict__ xb, long long cc, int C, int I, int kb, int tid,
                                         const BwdExtra& ex, int b) {
    constexpr int XN4 = 64 * HW / 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = tid + 256 * u, r = f >> 4, c4 = (f & 15) * 4;
        const long long o1 = cc + (long long)(I * 64 + r) * C + kb * 64 + c4;
        const long long o2 = cc + (long long)(kb * 64 + r) * C + I * 64 + c4;
        if (MODE == 0 || MODE == 3) ry[u] = *reinterpret_cast<const f32x4*>(y + o1);
        if (MODE != 2) {
            rd[u] = *reinterpret_cast<const f32x4*>(dy + o1);
            if (LOAD_T) rt[u] = *reinterpret_cast<const f32x4*>(dy + o2);
        } else {
            const int i = I * 64 + r;
            const int h1i = ex.h1[i], h2i = ex.h2[i];
            const float s1i = ex.s1[i], s2i = ex.s2[i];
            const float* d = ex.dc + (long long)b * ex.D;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kb * 64 + c4 + t;
                int ba = h1i + ex.h2[k]; if (ba >= ex.D) ba -= ex.D;
                int bb = ex.h1[k] + h2i; if (bb >= ex.D) bb -= ex.D;
                rd[u][t] = s1i * ex.s2[k] * d[ba] + ex.s1[k] * s2i * d[bb];
            }
        }
    }
    const f32x4* xs = reinterpret_cast<const f32x4*>(xb + (long long)kb * 64 * HW);
#pragma unroll
    for (int u = 0; u < NSX; ++u) {
        const int f = tid + 256 * u, fc = f < XN4 ? f : XN4 - 1;
        rx[u] = xs[fc];
        if (MODE == 1) rx[u] -= ex.mu[(long long)b * C + kb * 64 + (4 * fc) / HW];
    }
}

template <int HW, int MODE>
__global__ __launch_bounds__(256, 2) void bcnn_bwd_panel_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ dy,
                                                                const float* __restrict__ inv_norm,
                                                                float* __restrict__ dx, float* __restrict__ tpart,
                                                                int C, int nb, int B, BwdExtra ex) {
    constexpr int NT = (HW + 15) / 16;          // 16-column output tiles
    constexpr int XN4 = 64 * HW / 4;
    constexpr int NSX = (XN4 + 255) / 256;
    constexpr int PP = 68;                      // P tile pitch
    constexpr int TP = 65;                      // dy^T scratch pitch
    constexpr int XREG = (64 * HW + 16) > (64 * TP) ? (64 * HW + 16) : (64 * TP);
    __shared__ __attribute__((aligned(16))) float lds[64 * PP + XREG];
    float* sP = lds;
    float* sX = lds + 64 * PP;

    int b, I;
    if (!xcd_map(blockIdx.x, B, nb, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const long long cc = (long long)b * C * C;
    const float* xb = x + (long long)b * C * HW;
    float coef = 1.0f / (float)HW;                         // COV
    if (MODE == 0 || MODE == 3) {
        const float in = inv_norm[b];
        coef = in * in / (2.0f * (float)HW);
    }
    const float t2 = MODE == 3 ? 2.0f * bwd_t_of(ex, b) : 0.f;

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tacc = 0.f;

    f32x4 ry[4], rd[4], rt[4], rx[NSX];
    bwd_load<HW, NSX, MODE>(ry, rd, rt, rx, y, dy, xb, cc, C, I, 0, tid, ex, b);
    for (int kb = 0; kb < nb; ++kb) {
        __syncthreads();                                   // previous MFMA phase finished with sP / sX
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                  // dy(K,I) transposed into the scratch: T[i][k] = dy[k][i]
                const int f = tid + 256 * u, r = f >> 4, c4 = (f & 15) * 4;
                sX[(c4 + 0) * TP + r] = rt[u][0];
                sX[(c4 + 1) * TP + r] = rt[u][1];
                sX[(c4 + 2) * TP + r] = rt[u][2];
                sX[(c4 + 3) * TP + r] = rt[u][3];
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                      // P tile
            const int f = tid + 256 * u, r = f >> 4, c4 = (f & 15) * 4;
            const float* tp = sX + r * TP + c4;
            f32x4 p;
            if (MODE == 0) {
                // v_rcp_f32 (1 ulp) instead of an IEEE division: 16 of them per thread per K-block sit on the
                // critical path between two MFMA phases; the parity budget is 1e-4
                p[0] = (rd[u][0] + tp[0]) * (__builtin_amdgcn_rcpf(ry[u][0]) * coef);
                p[1] = (rd[u][1] + tp[1]) * (__builtin_amdgcn_rcpf(ry[u][1]) * coef);
                p[2] = (rd[u][2] + tp[2]) * (__builtin_amdgcn_rcpf(ry[u][2]) * coef);
                p[3] = (rd[u][3] + tp[3]) * (__builtin_amdgcn_rcpf(ry[u][3]) * coef);
                tacc += (ry[u][0] * rd[u][0] + ry[u][1] * rd[u][1]) + (ry[u][2] * rd[u][2] + ry[u][3] * rd[u][3]);
            } else if (MODE == 1) {
                p[0] = (rd[u][0] + tp[0]) * coef;
                p[1] = (rd[u][1] + tp[1]) * coef;
                p[2] = (rd[u][2] + tp[2]) * coef;
                p[3] = (rd[u][3] + tp[3]) * coef;
            } else if (MODE == 3) {                        // signed sqrt: (dy_ij + dy_ji - 2 t y_ij) / |y_ij|, 0 at y = 0
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float yv = ry[u][t];
                    p[t] = yv == 0.f ? 0.f : (rd[u][t] + tp[t] - t2 * yv) * (__builtin_amdgcn_rcpf(fabsf(yv)) * coef);
                }
            } else {
                p = rd[u];
            }
            *reinterpret_cast<f32x4*>(&sP[r * PP + c4]) = p;
        }
        __syncthreads();                                   // scratch reads done: X block may overwrite it
#pragma unroll
        for (int u = 0; u < NSX; ++u) {
            const int f = tid + 256 * u;
            if (f < XN4) reinterpret_cast<f32x4*>(sX)[f] = rx[u];
        }
        __syncthreads();
        // next K-block's operands: in flight during the MFMA phase (last iteration: harmless re-read)
        bwd_load<HW, NSX, MODE>(ry, rd, rt, rx, y, dy, xb, cc, C, I, (kb + 1 < nb ? kb + 1 : kb), tid, ex, b);

        const float* ap = sP + (wave * 16 + l15) * PP + 4 * lq;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 16 * s);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* bp = sX + (16 * s + 4 * lq + t) * HW + l15;
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bp[16 * n], acc[n], 0, 0, 0);
            }
        }
    }

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* dxb = dx + (long long)b * C * HW + (long long)(I * 64 + wave * 16 + lq * 4) * HW;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = 16 * n + l15;
        if (col < HW) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dxb[(long long)r * HW + col] = acc[n][r];
        }
    }
    if (MODE == 0) {
        __syncthreads();
        const float tsum = block_sum<4>(tacc, lds);
        if (tid == 0) tpart[(long long)b * nb + I] = tsum;
    }
}

// pair: -1 = the rule the pooling heads were tuned on (row blocks in pairs once there are more than 256 of them); 0 / 1 = the caller's choice
template <int HW, int MODE, bool CENTER>
static int gram_launch(const float* x, const float* inv_norm, float* y, int B, int C, float* mu, float alpha,
                       hipStream_t st, GramNormSrc ns = GramNormSrc{nullptr, nullptr, nullptr, 0, 0}, int pair = -1) {
    const int nb = C / 64;
    const int pair_mode = pair >= 0 ? pair : (((long long)B * nb > 256) ? 1 : 0);
    const int per = pair_mode ? (nb + 1) / 2 : nb;
    hipLaunchKernelGGL((bcnn_gram_panel_kernel<HW, MODE, CENTER>), dim3(xcd_grid(B, per)), dim3(256), 0, st, x, inv_norm,
                       y, C, nb, B, pair_mode, mu, alpha, ns);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

// The batch size the work-split decisions below see: the real one, or tuning().sched_b (tests: the large-batch forms on
// small inputs; results never depend on it)
static inline int sched_batch(int B) { return tuning().sched_b > 0 ? tuning().sched_b : B; }

// hk_bwd3c.h: block height (rb = 2: 128 rows, 1: 64, 0: do not use the kernel) and column split.  128-row blocks where
// they fill the chip, else 64-row blocks, else 64-row blocks with the column tiles divided between two workgroups
// (B = 16, C = 512: 256 workgroups; nothing to add up).  Measured at B = 16: 64-row blocks 38.5 us, 128-row 64, column
// split 26.6 (a channel split with float atomics onto a zeroed dX: 35.8, removed).
static inline void cbp_bwd3_shape(int Breal, int C, int& rb, int& nsp) {
    rb = 0; nsp = 1;
    const int nb = C / 64, B = sched_batch(Breal);
    const long long n128 = C % 128 == 0 ? (long long)B * (C / 128) : 0, n64 = (long long)B * nb;
    if (n128 >= 192) rb = 2;
    else if (n64 >= 192) rb = 1;
    else if (2 * n64 >= 128) { rb = 1; nsp = 2; }
}

// tuning().bwd_v: 0 automatic, 1 force the four-wave 64-row panel kernel (the reference form the tests compare with)
template <int HW, int MODE>
static int bwd_launch(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart,
                      int B, int C, const BwdExtra& ex, hipStream_t st) {
    const int nb = C / 64, Bs = sched_batch(B);
    const int v = tuning().bwd_v;
    // gram_bwd3_kernel (hk_bwd3.h) for the BCNN, signed-sqrt and covariance modes wherever its blocks fill the chip:
    // 128-row blocks when B C / 128 >= 192, else 64-row blocks when B C / 64 >= 192.  Measured at B = 64, 14 x 14: BCNN
    // C = 512 64.1 us against 79.1 on the panel kernel; covariance C = 256 20.2 against 27.9.
    if constexpr (MODE == 0 || MODE == 1 || MODE == 3) {
        const bool fill2 = C % 128 == 0 && (long long)Bs * (C / 128) >= 192;
        const bool fill1 = (long long)Bs * nb >= 192;
        if (v == 0 && (fill2 || fill1)) {
            int rc = HK_ERR_UNSUPPORTED;
            if (fill2) rc = bwd3_launch<HW, MODE, 2>(x, y, dy, inv_norm, dx, tpart, B, C, ex, st);
            if (rc == HK_ERR_UNSUPPORTED) rc = bwd3_launch<HW, MODE, 1>(x, y, dy, inv_norm, dx, tpart, B, C, ex, st);
            if (rc != HK_ERR_UNSUPPORTED) return rc;
        }
    }
    // compact bilinear: cbp_bwd3_kernel (hk_bwd3c.h: P generated from dc in LDS, X by LDS-DMA); shape by cbp_bwd3_shape
    if constexpr (MODE == 2) {
        int rb = 0, nsp = 1;
        if (v == 0) cbp_bwd3_shape(B, C, rb, nsp);
        if (rb) {
            int rc = HK_ERR_UNSUPPORTED;
            if (rb == 2) rc = cbp_bwd3_launch<HW, 2>(x, dx, B, C, ex, nsp, st);
            if (rc == HK_ERR_UNSUPPORTED) rc = cbp_bwd3_launch<HW, 1>(x, dx, B, C, ex, nsp, st);
            if (rc == HK_ERR_UNSUPPORTED && nsp == 2) rc = cbp_bwd3_launch<HW, 1>(x, dx, B, C, ex, 1, st);
            if (rc != HK_ERR_UNSUPPORTED) return rc;
        }
    }
    // small batches (the row blocks do not fill the chip) and forced: the four-wave panel kernel, two workgroups per CU
    hipLaunchKernelGGL((bcnn_bwd_panel_kernel<HW, MODE>), dim3(xcd_grid(B, nb)), dim3(256), 0, st, x, y, dy, inv_norm, dx,
                       tpart, C, nb, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

#define HK_HW_SWITCH(CALL)                      \
    switch (HW) {                               \
        case 196: return CALL(196);             \
        case 144: return CALL(144);             \
        case 100: return CALL(100);             \
        case 64: return CALL(64);               \
        default: return HK_ERR_UNSUPPORTED;     \
    }

// All return HK_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the generic GEMM).
int bcnn_fast_gram(const float* x, const float* inv_norm, float* y, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(y)) return HK_ERR_UNSUPPORTED;
#define CALL(H) gram_launch<H, 0, false>(x, inv_norm, y, B, C, nullptr, 1.f, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// the same with the norm formed in the kernel from the column-sum partials part [B][G][HW] (G = C / 64 groups);
// colsum [B][HW] and inv_norm [B] are WRITTEN
int bcnn_fast_gram_norm(const float* x, const float* part, int G, float* colsum, float* inv_norm, float* y, int B, int C,
                        int HW, hipStream_t st) {
    if (C % 64 != 0 || G != C / 64 || !aligned16(x) || !aligned16(y)) return HK_ERR_UNSUPPORTED;
    const GramNormSrc ns{part, colsum, inv_norm, G, part ? 0 : 1};       // no partials: the kernel adds up the columns itself
    if (!part && (size_t)G * 256 * 16 > (size_t)2 * 64 * HW * sizeof(float)) return HK_ERR_UNSUPPORTED;   // phase sums in two panel buffers
#define CALL(H) gram_launch<H, 0, false>(x, nullptr, y, B, C, nullptr, 1.f, st, ns)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// signed-sqrt Gram (BCNN.py:23-24): y = u = sign(g) sqrt(|g| + 1e-10), g = X X^T / HW, un-normalised; part [B][64]
// receives *nparts partial sums of u^2 per image
int gram_fast_ssqrt(const float* x, float* y, float* part, int* nparts, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(y)) return HK_ERR_UNSUPPORTED;
    const int nb = C / 64;
    const int per = ((long long)B * nb > 256) ? (nb + 1) / 2 : nb;
    if (per > SSQ_STRIDE) return HK_ERR_UNSUPPORTED;
    *nparts = per;
#define CALL(H) gram_launch<H, 2, false>(x, nullptr, y, B, C, part, 1.0f / (float)HW, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// G = alpha * X X^T (raw Gram, CBP; mu == nullptr) or alpha * (X - mu 1^T) X^T (covariance; mu [B, C] is WRITTEN: the
// channel means, computed by the kernel itself)
int gram_fast_raw(const float* x, float* mu, float alpha, float* g, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(g)) return HK_ERR_UNSUPPORTED;
    if (mu) {
#define CALL(H) gram_launch<H, 1, true>(x, nullptr, g, B, C, mu, alpha, st)
        HK_HW_SWITCH(CALL)
#undef CALL
    }
#define CALL(H) gram_launch<H, 1, false>(x, nullptr, g, B, C, nullptr, alpha, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// G = alpha X X^T for a caller outside the pooling heads (CIN's interaction matrix at 14 x 14 maps: C = 2048, 32 row blocks per
// sample).  The row blocks go singly or in pairs, whichever leaves the shorter longest queue of 64 x 64 tiles on the 256 CUs
// (one workgroup per CU: 150 KB of LDS): B = 20, C = 2048 -> 640 single blocks of <= 17 tiles in 3 rounds against 320 pairs of
// 33 tiles in 2.
int gram_fast_scaled(const float* x, float alpha, float* g, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(g)) return HK_ERR_UNSUPPORTED;
    const long long nb = C / 64;
    const long long single = ((B * nb + 255) / 256) * (nb / 2 + 1), paired = ((B * ((nb + 1) / 2) + 255) / 256) * (nb + 1);
    const int pair = paired < single ? 1 : 0;
#define CALL(H) gram_launch<H, 1, false>(x, nullptr, g, B, C, nullptr, alpha, st, GramNormSrc{nullptr, nullptr, nullptr, 0, 0}, pair)
    HK_HW_SWITCH(CALL)
#undef CALL
}

int bcnn_fast_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart, int B,
                  int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(y) || !aligned16(dy) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
#define CALL(H) bwd_launch<H, 0>(x, y, dy, inv_norm, dx, tpart, B, C, ex, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// The BCNN backward in ONE launch (hk_bwd3.h, TK 1): t = <y, dy> is known as sum_k ta[b][k] (tb[b][k] - tc[k]) and the
// rank-1 term is applied in the GEMM kernel's epilogue.  HK_ERR_UNSUPPORTED - nothing launched - where gram_bwd3_kernel
// does not run (other map sizes, batches whose row blocks do not fill the chip, the bwd_v knob): the caller then takes
// the two-launch route.
template <int HW>
static int bwd_fold_launch(const float* x, const float* y, const float* dy, const float* inv_norm, const float* colsum,
                           const float* ta, const float* tb, const float* tc, int tK, float* dx, int B, int C,
                           hipStream_t st) {
    const int nb = C / 64, Bs = sched_batch(B);
    const bool fill2 = C % 128 == 0 && (long long)Bs * (C / 128) >= 192;
    const bool fill1 = (long long)Bs * nb >= 192;
    if (tuning().bwd_v != 0 || !(fill2 || fill1)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.colsum = colsum;
    ex.ta = ta; ex.tb2 = tb; ex.tc = tc; ex.tK = tK;
    int rc = HK_ERR_UNSUPPORTED;
    if (fill2) rc = bwd3_launch<HW, 0, 2, 1>(x, y, dy, inv_norm, dx, nullptr, B, C, ex, st);
    if (rc == HK_ERR_UNSUPPORTED) rc = bwd3_launch<HW, 0, 1, 1>(x, y, dy, inv_norm, dx, nullptr, B, C, ex, st);
    return rc;
}

int bcnn_fast_bwd_fold(const float* x, const float* y, const float* dy, const float* inv_norm, const float* colsum,
                       const float* ta, const float* tb, const float* tc, int tK, float* dx, int B, int C, int HW,
                       hipStream_t st) {
    if (C % 64 != 0 || !ta || !tb || !aligned16(x) || !aligned16(y) || !aligned16(dy) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
#define CALL(H) bwd_fold_launch<H>(x, y, dy, inv_norm, colsum, ta, tb, tc, tK, dx, B, C, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// signed-sqrt variant with t = <y, dy> handed over as a dot product (the classifier's backward knows it: hk_bwd3.h TK 1):
// no pass over y and dy for the partial sums.  t_inv2: y is the un-normalised u (inv_norm = the true 1 / |u|): the
// coefficient carries one factor inv less, t one more.  HK_ERR_UNSUPPORTED where gram_bwd3_kernel does not run.
template <int HW>
static int ssqrt_tdot_launch(const float* x, const float* y, const float* dy, const float* inv_norm, const float* ta,
                             const float* tb, const float* tc, int tK, int t_inv2, float* dx, int B, int C, hipStream_t st) {
    const int nb = C / 64, Bs = sched_batch(B);
    const bool fill2 = C % 128 == 0 && (long long)Bs * (C / 128) >= 192;
    const bool fill1 = (long long)Bs * nb >= 192;
    if (tuning().bwd_v != 0 || !(fill2 || fill1)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.ta = ta; ex.tb2 = tb; ex.tc = tc; ex.tK = tK; ex.t_inv2 = t_inv2;
    int rc = HK_ERR_UNSUPPORTED;
    if (fill2) rc = bwd3_launch<HW, 3, 2, 1>(x, y, dy, inv_norm, dx, nullptr, B, C, ex, st);
    if (rc == HK_ERR_UNSUPPORTED) rc = bwd3_launch<HW, 3, 1, 1>(x, y, dy, inv_norm, dx, nullptr, B, C, ex, st);
    return rc;
}
int bcnn_ssqrt_fast_bwd_tdot(const float* x, const float* y, const float* dy, const float* inv_norm, const float* ta,
                             const float* tb, const float* tc, int tK, int t_inv2, float* dx, int B, int C, int HW,
                             hipStream_t st) {
    if (C % 64 != 0 || !ta || !tb || !aligned16(x) || !aligned16(y) || !aligned16(dy) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
#define CALL(H) ssqrt_tdot_launch<H>(x, y, dy, inv_norm, ta, tb, tc, tK, t_inv2, dx, B, C, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// signed-sqrt variant (BCNN.py:23-24): P = (dy + dy^T - 2 t y) / |y| * inv^2 / (2M), t from its partial sums
int bcnn_ssqrt_fast_bwd(const float* x, const float* y, const float* dy, const float* inv_norm, const float* tpart, int nt,
                        float* dx, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(y) || !aligned16(dy) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.tb = tpart; ex.nt = nt;
#define CALL(H) bwd_launch<H, 3>(x, y, dy, inv_norm, dx, nullptr, B, C, ex, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// dX = (1/M) (g + g^T) (X - mu)
int cov_fast_bwd(const float* x, const float* mu, const float* g, float* dx, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(g) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.mu = mu;
#define CALL(H) bwd_launch<H, 1>(x, nullptr, g, nullptr, dx, nullptr, B, C, ex, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

template <int HW>
static int cbp_bwd3_try(const float* x, float* dx, int B, int C, const BwdExtra& ex, int rb, int nsp, hipStream_t st) {
    int rc = HK_ERR_UNSUPPORTED;
    if (rb == 2) rc = cbp_bwd3_launch<HW, 2>(x, dx, B, C, ex, nsp, st);
    if (rc == HK_ERR_UNSUPPORTED) rc = cbp_bwd3_launch<HW, 1>(x, dx, B, C, ex, nsp, st);
    if (rc == HK_ERR_UNSUPPORTED && nsp == 2) rc = cbp_bwd3_launch<HW, 1>(x, dx, B, C, ex, 1, st);   // odd tile count
    return rc;
}

// The same with dc computed inside the GEMM kernel from the forward's saved state (hk_bwd3c.h).  HK_ERR_UNSUPPORTED -
// nothing launched - when the shape / batch would not take that kernel: the caller then forms dc itself.
int cbp_fast_bwd_fused(const float* x, const int* h1, const int* h2, const float* s1, const float* s2, const float* y,
                       const float* dy, const float* c_raw, const float* inv_norm, int D, float* dx, int B, int C, int HW,
                       hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    if (tuning().bwd_v != 0) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.h1 = h1; ex.h2 = h2; ex.s1 = s1; ex.s2 = s2; ex.dc = nullptr; ex.D = D;
    ex.cy = y; ex.cdy = dy; ex.ccraw = c_raw; ex.cinv = inv_norm;
    int rb, nsp;
    cbp_bwd3_shape(B, C, rb, nsp);
    if (!rb) return HK_ERR_UNSUPPORTED;
#define CALL(H) cbp_bwd3_try<H>(x, dx, B, C, ex, rb, nsp, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

// dX = (dG + dG^T) X,  dG_ij = s1_i s2_j dc[(h1_i + h2_j) mod D]
int cbp_fast_bwd(const float* x, const int* h1, const int* h2, const float* s1, const float* s2, const float* dc, int D,
                 float* dx, int B, int C, int HW, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    BwdExtra ex = {};
    ex.h1 = h1; ex.h2 = h2; ex.s1 = s1; ex.s2 = s2; ex.dc = dc; ex.D = D;
#define CALL(H) bwd_launch<H, 2>(x, nullptr, nullptr, nullptr, dx, nullptr, B, C, ex, st)
    HK_HW_SWITCH(CALL)
#undef CALL
}

}  // namespace hk

