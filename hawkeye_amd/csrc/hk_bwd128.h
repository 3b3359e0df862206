// Gram backward, second structure:  dX[I rows] = sum_K P(I,K) X(K)  for 128-row blocks I, one 8-wave workgroup per CU.
//
// Why (profiles/r1c_sq_wait_counters.csv, BENCH_r01, profiles/r2_bwd_lab.json): the 64-row kernel
// (bcnn_bwd_panel_kernel) sits at 85-88 us = 0.48-0.49 of the fp32 MFMA peak.  Cycle stamps inside it (tools/bwd_lab.py)
// give 18.3k cycles per K-block of which 11.2k are the MFMA phase (6.7k when alone on the SIMD) and 7.1k are phases
// that issue no MFMA - transposing scatter of dy(K,I) 0.9k, P-tile build 1.6k, X store 1.0k, issuing the next block's
// 25 loads 2.7k, four barriers - more than the partner workgroup's MFMA phase can cover.  This kernel:
//   * no P tile: LDS holds the RAW tiles  S1 = dy(I,K) [i][k],  S2 = dy(K,I) [k][i],  W = coef / y(I,K) [i][k]  exactly
//     as they lie in HBM (coalesced 16-byte loads, 16-byte LDS stores, no transposition), and the MFMA A operand is
//     formed when the fragment is read:  a = (S1[i][k] + S2[k][i]) * W[i][k]  - for the 16x16x4 A layout lanes run
//     along i, so the "transposed" read of S2 is a conflict-free ds_read_b32 of consecutive addresses;
//   * each wave owns 32 rows x 7 (6) column tiles: an X fragment feeds two MFMAs, half the LDS reads per MFMA;
//   * 128-row blocks: X is staged by 4 workgroups per image instead of 8 (the vector-memory pipe of a CU moves
//     73 KB per K-block instead of 98 KB per 64 channels);
//   * K-blocks of 32 channels, two LDS stages (2 x 77 KB), ONE barrier per K-block; 8 waves = two per SIMD, the next
//     block's loads / LDS stores cut into four parts each and placed between the eight MFMA groups of a block, operand
//     fragments double-buffered one group ahead.
// Measured (B = 64, C = 512, 14x14; same box, alternating): 64-row 85.3 us, this kernel 77.5 us; with the staging
// removed from the loop 66.2 us, with the per-group fragment reads removed as well 62.7 us (timing-only builds) - the
// MFMA stream + 16 barriers + prologue / epilogue of this tiling is 0.66 of the peak, the staging costs another 0.12.
// A first version with ONE wave per SIMD (256 threads) ran 94-111 us: every staging instruction sat in the MFMA issue
// stream (15.1k cycles per K-block for 6.7k cycles of MFMAs).
// Same arithmetic per element as the 64-row kernel ((dy + dy^T) * (rcp(y) * coef)): bit-identical dX.
// MODE 0 BCNN   P = (dy + dy^T) / y * inv^2 / (2M)          MODE 1 COV   P = (g + g^T) / M, X centred
// MODE 2 CBP    P = dG + dG^T gathered from dc               MODE 3 signed-sqrt BCNN (BCNN.py:23-24): see bcnn_pool.hip
#pragma once
#include "hk_common.h"

namespace hk {

// HK_LAB builds only (make lab -> libhawkeye_hip_lab.so, tools/bwd_lab.py): cycle stamps of wave 0 of the first 64
// workgroups at the phase boundaries of every K-block.  Compiled out of the product library.
#ifdef HK_LAB
extern __device__ long long* g_lab_stamps;               // [64 workgroups][32 K-blocks][8 stamps]
#define HK_STAMP(kb_, slot_)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 64 && g_lab_stamps)                                      \
            g_lab_stamps[((long long)blockIdx.x * 32 + (kb_)) * 8 + (slot_)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HK_STAMP(kb_, slot_) do { } while (0)
#endif

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
    int dc_lds;          // CBP, 128-row kernel: the sample's dc vector and the hash / sign tables are copied to LDS once
    // CBP, hk_bwd3c.h with dc == nullptr: dc is computed by the kernel itself from the saved forward state
    const float* cy;     // [B][D]  y
    const float* cdy;    // [B][D]  dL/dy
    const float* ccraw;  // [B][D]  bins before the signed square root
    const float* cinv;   // [B]     1 / max(|u|, 1e-12)
};

// t = <y, dy> of sample b from its partial sums (every workgroup adds them itself, fixed order)
__device__ __forceinline__ float bwd_t_of(const BwdExtra& ex, int b) {
    float t = 0.f;
    for (int c = 0; c < ex.nt; ++c) t += ex.tb[(long long)b * ex.nt + c];
    return t;
}

// 512 threads = 8 waves = TWO waves per SIMD: wave w owns rows 32 (w & 3) .. + 31 and the column tiles of half w >> 2
// (7 + 6 of the 13 tiles at HW = 196), so both waves of a SIMD run the same stream half a tile-group apart and one's
// staging / fragment reads issue in the shadow of the other's MFMAs (a single wave per SIMD exposed all of them: the lab
// build measured 15.1k cycles per K-block for 6.7k cycles of MFMAs, tools/bwd_lab.py).  Global loads run TWO K-blocks
// ahead through two register sets; the loads and the LDS stores of a K-block are cut into four parts each and placed
// between the eight MFMA groups of the block.
// LABV (HK_LAB builds, timing only - results are wrong): 1 = no staging inside the loop (MFMA + fragment stream alone),
// 2 = staging but the fragments are read once per K-block only (no per-group LDS reads), 3 = both removed
// RB: 16-row blocks per wave = 2 (128-row blocks, the structure described above) or 1 (64-row blocks with the same eight
// waves: for shapes whose 128-row blocks would not fill the chip, e.g. the covariance at C = 256, B = 64 - the 4-wave
// panel kernel holds ONE wave per SIMD there and nothing covers its staging phases)
template <int HW, int MODE, int LABV = 0, int RB = 2>
__global__ __launch_bounds__(512, 2) void bcnn_bwd128_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                             float* __restrict__ tpart, int C, int nI, int B,
                                                             BwdExtra ex) {
    constexpr int NT = (HW + 15) / 16;          // 16-column output tiles
    constexpr int NH = (NT + 1) / 2;            // tiles of the first column half (the second has NT - NH)
    constexpr int KB = 32;                      // channels per K-block
    constexpr int P1 = KB + 4;                  // pitch of the [i][k] tiles (ds_read_b128: pitch / 4 odd)
    constexpr int IB = 64 * RB;                 // rows of a block
    constexpr int P2 = IB + 4;                  // pitch of the [k][i] tile
    constexpr bool HAS_W = MODE == 0 || MODE == 3;
    constexpr bool HAS_S2 = MODE != 2;
    constexpr int S1_SZ = IB * P1, W_SZ = HAS_W ? IB * P1 : 0, S2_SZ = HAS_S2 ? KB * P2 : 0;
    constexpr int XN4 = KB * HW / 4;            // float4 of one X block
    constexpr int NSX = (XN4 + 511) / 512;      // (<= 4)
    // floats, rounded to 64 B, + 16: the last column tile reads columns HW .. 16 NT - 1 of every channel row (never
    // stored); behind the last row of the block that is up to 12 floats past the X block
    constexpr int X_SZ = (XN4 + 3) / 4 * 16 + 16;
    constexpr int STAGE = S1_SZ + W_SZ + S2_SZ + X_SZ;
    static_assert(NSX <= 4, "X block staging assumes at most four 16-byte vectors per thread");
    HK_DYN_LDS16(lds);

    int b, I;
    if (!xcd_map(blockIdx.x, B, nI, b, I)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wrow = (wave & 3) * (16 * RB);
    const int half = __builtin_amdgcn_readfirstlane(wave) >> 2;   // wave-uniform: the tile guards below must be scalar branches
    const int nt0 = half * NH, nloc = half ? NT - NH : NH;      // this wave's column tiles: nt0 .. nt0 + nloc - 1
    const long long cc = (long long)b * C * C;
    const float* xb = x + (long long)b * C * HW;
    const int nkb = C / KB;                                     // even (C % 64 == 0)
    float coef = 1.0f / (float)HW;                              // COV
    if (HAS_W) {
        const float in = inv_norm[b];
        coef = in * in / (2.0f * (float)HW);
    }
    const float t2 = MODE == 3 ? 2.0f * bwd_t_of(ex, b) : 0.f;

    f32x4 acc[RB][NH];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tacc = 0.f;

    // staging registers of the NEXT K-block (requested during the first half of the current block, stored to the other
    // LDS stage during its second half); named members: an indexed array that is loaded and stored in different
    // blocks is not promoted to registers
    struct Regs {
        f32x4 y0, y1, d0, d1, t0, t1, x0, x1, x2, x3;
        float m0, m1, m2, m3;                                   // COV: channel mean of each staged X vector
    };
    Regs R0;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    R0.y0 = R0.y1 = R0.d0 = R0.d1 = R0.t0 = R0.t1 = R0.x0 = R0.x1 = R0.x2 = R0.x3 = z4;
    R0.m0 = R0.m1 = R0.m2 = R0.m3 = 0.f;

    const int r1 = tid >> 3, c1 = 4 * (tid & 7);           // [i][k] tiles: row r1 + 64 u (u < RB), k offset c1
    // [k][i] tile: RB = 2: k row r2 + 16 u, 32 float4 per row; RB = 1: k row r2 (32 rows at once), 16 float4 per row
    const int r2 = RB == 2 ? tid >> 5 : tid >> 4, c2 = RB == 2 ? 4 * (tid & 31) : 4 * (tid & 15);

    // addresses = wave-uniform base (advances with the K-block, lives in SGPRs) + a loop-invariant 32-bit offset per
    // thread: ten 64-bit per-thread pointers would not fit next to two staging register sets
    const int o1[2] = {r1 * C + c1, (r1 + 64) * C + c1};
    const int o2[2] = {r2 * C + c2, (r2 + 16) * C + c2};
    const float* ybase = y + cc + (long long)I * IB * C;       // + kb * KB
    const float* dbase = dy + cc + (long long)I * IB * C;      // + kb * KB
    const float* tbase = dy + cc + I * IB;                     // + kb * KB * C
    auto ld1 = [&](const float* base, int kb, int u) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(base + kb * KB + o1[u]);
    };
    auto ld2 = [&](int kb, int u) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(tbase + (long long)kb * KB * C + o2[u]);
    };
    // CBP: P(I,K) from the dc vector (CBCNN.py backward).  Every element costs two gathers from dc and the hashes /
    // signs of its column: from global memory that was 32 dependent loads per thread and K-block (L2 hits, but the
    // vector-memory pipe and their latency sat in front of the LDS stores).  The sample's dc (D floats) and the four
    // tables (C entries each) are copied to LDS once per workgroup behind the two stages when they fit (ex.dc_lds);
    // the row's own hash / sign values are loop-invariant registers.
    const float* dcl = lds + 2 * STAGE;                        // [D], then h1, h2 (int), s1, s2 [C]
    int hi1[2] = {0, 0}, hi2[2] = {0, 0};
    float si1[2] = {0.f, 0.f}, si2[2] = {0.f, 0.f};
    if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = I * IB + r1 + 64 * (u < RB ? u : 0);
            hi1[u] = ex.h1[i]; hi2[u] = ex.h2[i]; si1[u] = ex.s1[i]; si2[u] = ex.s2[i];
        }
        if (ex.dc_lds) {
            float* dw = lds + 2 * STAGE;
            const float* dcb = ex.dc + (long long)b * ex.D;
            for (int e = tid; e < ex.D; e += 512) dw[e] = dcb[e];
            int* tw = reinterpret_cast<int*>(dw + ex.D);
            for (int e = tid; e < C; e += 512) {
                tw[e] = ex.h1[e];
                tw[C + e] = ex.h2[e];
                reinterpret_cast<float*>(tw)[2 * C + e] = ex.s1[e];
                reinterpret_cast<float*>(tw)[3 * C + e] = ex.s2[e];
            }
            __syncthreads();
        }
    }
    auto gather = [&](int kb, int u) -> f32x4 {
        const int h1i = hi1[u], h2i = hi2[u];
        const float s1i = si1[u], s2i = si2[u];
        f32x4 p;
        if (ex.dc_lds) {
            const int* th1 = reinterpret_cast<const int*>(dcl + ex.D);
            const int* th2 = th1 + C;
            const float* ts1 = reinterpret_cast<const float*>(th1) + 2 * C;
            const float* ts2 = ts1 + C;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kb * KB + c1 + t;
                int ba = h1i + th2[k]; if (ba >= ex.D) ba -= ex.D;
                int bb = th1[k] + h2i; if (bb >= ex.D) bb -= ex.D;
                p[t] = s1i * ts2[k] * dcl[ba] + ts1[k] * s2i * dcl[bb];
            }
        } else {
            const float* dcb = ex.dc + (long long)b * ex.D;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kb * KB + c1 + t;
                int ba = h1i + ex.h2[k]; if (ba >= ex.D) ba -= ex.D;
                int bb = ex.h1[k] + h2i; if (bb >= ex.D) bb -= ex.D;
                p[t] = s1i * ex.s2[k] * dcb[ba] + ex.s1[k] * s2i * dcb[bb];
            }
        }
        return p;
    };
    int ox[4];                                                  // clamped float4 index of this thread's X vectors
#pragma unroll
    for (int u = 0; u < 4; ++u) ox[u] = (tid + 512 * u < XN4) ? tid + 512 * u : XN4 - 1;
    auto ldx = [&](int kb, int u, float& mean) -> f32x4 {
        mean = MODE == 1 ? ex.mu[(long long)b * C + kb * KB + (4 * ox[u]) / HW] : 0.f;
        return reinterpret_cast<const f32x4*>(xb + (long long)kb * KB * HW)[ox[u]];
    };
    // the loads of one K-block in four parts (placed between MFMA groups)
#define HK_BW_GLOAD(R, kb, part)                                                                               \
    do {                                                                                                       \
        if ((part) == 0) { if (HAS_W) { R.y0 = ld1(ybase, kb, 0); if (RB == 2) R.y1 = ld1(ybase, kb, 1); } }    \
        if ((part) == 1) { if (MODE != 2) { R.d0 = ld1(dbase, kb, 0); if (RB == 2) R.d1 = ld1(dbase, kb, 1); } \
                           else { R.d0 = gather(kb, 0); if (RB == 2) R.d1 = gather(kb, 1); } }                 \
        if ((part) == 2) { if (HAS_S2) { R.t0 = ld2(kb, 0); if (RB == 2) R.t1 = ld2(kb, 1); } }                 \
        if ((part) == 3) { R.x0 = ldx(kb, 0, R.m0); R.x1 = ldx(kb, 1, R.m1);                                   \
                           if (NSX > 2) R.x2 = ldx(kb, 2, R.m2);                                               \
                           if (NSX > 3) R.x3 = ldx(kb, 3, R.m3); }                                             \
    } while (0)

    // W = coef / y (v_rcp_f32, 1 ulp: parity budget 1e-4);  signed sqrt: |y| in the denominator, 0 where y == 0
    auto wof = [&](f32x4 yv) -> f32x4 {
        f32x4 w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 3) w[t] = yv[t] == 0.f ? 0.f : __builtin_amdgcn_rcpf(fabsf(yv[t])) * coef;
            else w[t] = __builtin_amdgcn_rcpf(yv[t]) * coef;
        }
        return w;
    };
    auto sst1 = [&](float* S1, float* Wt, int u, f32x4 yv, f32x4 dv) {
        const int o = (r1 + 64 * u) * P1 + c1;
        if (MODE == 0) tacc += (yv[0] * dv[0] + yv[1] * dv[1]) + (yv[2] * dv[2] + yv[3] * dv[3]);
        if (MODE == 1) dv *= coef;
        if (MODE == 3) dv -= t2 * yv;                          // (dy_ij + dy_ji - 2 t y_ij) / |y_ij|: the whole -2 t y term here
        *reinterpret_cast<f32x4*>(S1 + o) = dv;
        if (HAS_W) *reinterpret_cast<f32x4*>(Wt + o) = wof(yv);
    };
    auto sstx = [&](float* X_, int u, f32x4 v, float mean) {
        const int f = tid + 512 * u;
        // the mean is subtracted HERE, not after the load: touching the value there parks the wave on it
        if (f < XN4) reinterpret_cast<f32x4*>(X_)[f] = MODE == 1 ? v - mean : v;
    };
    // the LDS stores of one K-block (registers R -> stage st), in four parts
#define HK_BW_SSTORE(R, st, part)                                                                              \
    do {                                                                                                       \
        float* S1_ = lds + (st) * STAGE;                                                                       \
        float* W_ = S1_ + S1_SZ;                                                                               \
        float* S2_ = W_ + W_SZ;                                                                                \
        float* X_ = S2_ + S2_SZ;                                                                               \
        if ((part) == 0) sst1(S1_, W_, 0, R.y0, R.d0);                                                         \
        if ((part) == 1 && RB == 2) sst1(S1_, W_, 1, R.y1, R.d1);                                              \
        if ((part) == 2 && HAS_S2) {                                                                           \
            const float sc_ = MODE == 1 ? coef : 1.0f;                                                         \
            *reinterpret_cast<f32x4*>(S2_ + (r2 + 0) * P2 + c2) = R.t0 * sc_;                                  \
            if (RB == 2) *reinterpret_cast<f32x4*>(S2_ + (r2 + 16) * P2 + c2) = R.t1 * sc_;                    \
        }                                                                                                      \
        if ((part) == 3) { sstx(X_, 0, R.x0, R.m0); sstx(X_, 1, R.x1, R.m1);                                   \
                           if (NSX > 2) sstx(X_, 2, R.x2, R.m2);                                               \
                           if (NSX > 3) sstx(X_, 3, R.x3, R.m3); }                                             \
    } while (0)

    // A fragments of the wave's two 16-row blocks for k = 16 s + 4 lq + t, formed from the raw tiles
#define HK_BW_AFRAG(A_, s_)                                                                                    \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                       \
            const int row_ = wrow + i * 16 + l15;                                                              \
            const f32x4 d1_ = *reinterpret_cast<const f32x4*>(S1 + row_ * P1 + 16 * (s_) + 4 * lq);            \
            f32x4 wv_ = (f32x4){1.f, 1.f, 1.f, 1.f};                                                           \
            if (HAS_W) wv_ = *reinterpret_cast<const f32x4*>(Wt + row_ * P1 + 16 * (s_) + 4 * lq);             \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
                float v_ = d1_[t];                                                                             \
                if (HAS_S2) v_ += S2[(16 * (s_) + 4 * lq + t) * P2 + row_];                                    \
                A_[i][t] = HAS_W ? v_ * wv_[t] : v_;                                                           \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#define HK_BW_BFRAG(B_, s_, t_)                                                                                \
    do {                                                                                                       \
        if ((LABV & 2) && ((s_) != 0 || (t_) > 1)) break;                                                      \
        const float* bp_ = X + (16 * (s_) + 4 * lq + (t_)) * HW + 16 * nt0 + l15;                              \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) B_[n] = (n < nloc) ? bp_[16 * n] : 0.f;                 \
    } while (0)
#define HK_BW_MFMA(A_, B_, t_)                                                                                 \
    do {                                                                                                       \
        _Pragma("unroll") for (int n = 0; n < NH; ++n) {                                                       \
            if (n < NH - 1 || n < nloc) {                                                                      \
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[0][t_], B_[n], acc[0][n], 0, 0, 0);        \
                if (RB == 2) acc[RB - 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[RB - 1][t_], B_[n], acc[RB - 1][n], 0, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
    // One K-block: eight MFMA groups (s = 0, 1; t = 0..3).  The operand fragments of a group are read while the
    // previous group's MFMAs run (explicit double buffering: bA / bB); the loads of K-block kb_ + 1 are requested in four
    // parts behind groups 0-3 and stored to the other LDS stage in four parts behind groups 4-7 (each part has four
    // groups = >3000 matrix-pipe cycles to arrive).  MORE_ is compile-time: no branch around the loads.
#define HK_BW_KBLOCK(kb_, RL, RS, LOAD_, STORE_)                                                               \
    do {                                                                                                       \
        const int cur = (kb_) & 1;                                                                             \
        const float* S1 = lds + cur * STAGE;                                                                   \
        const float* Wt = S1 + S1_SZ;                                                                          \
        const float* S2 = Wt + W_SZ;                                                                           \
        const float* X = S2 + S2_SZ;                                                                           \
        HK_STAMP(kb_, 0);                                                                                      \
        float a0[RB][4], a1[RB][4], bA[NH], bB[NH];                                                            \
        HK_BW_AFRAG(a0, 0);                                                                                    \
        HK_BW_BFRAG(bA, 0, 0);                                                                                 \
        HK_STAMP(kb_, 1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bB, 0, 1); HK_BW_MFMA(a0, bA, 0); if (LOAD_) HK_BW_GLOAD(RL, (kb_) + 1, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bA, 0, 2); HK_BW_MFMA(a0, bB, 1); if (LOAD_) HK_BW_GLOAD(RL, (kb_) + 1, 1);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bB, 0, 3); HK_BW_MFMA(a0, bA, 2); if (LOAD_) HK_BW_GLOAD(RL, (kb_) + 1, 2);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (LABV & 2) { _Pragma("unroll") for (int t = 0; t < 4; ++t) { a1[0][t] = a0[0][t]; a1[RB - 1][t] = a0[RB - 1][t]; } }   \
        else HK_BW_AFRAG(a1, 1);                                                                               \
        HK_BW_BFRAG(bA, 1, 0); HK_BW_MFMA(a0, bB, 3); if (LOAD_) HK_BW_GLOAD(RL, (kb_) + 1, 3);                \
        HK_STAMP(kb_, 2);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bB, 1, 1); HK_BW_MFMA(a1, bA, 0); if (STORE_) HK_BW_SSTORE(RS, cur ^ 1, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bA, 1, 2); HK_BW_MFMA(a1, bB, 1); if (STORE_) HK_BW_SSTORE(RS, cur ^ 1, 1);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_BFRAG(bB, 1, 3); HK_BW_MFMA(a1, bA, 2); if (STORE_) HK_BW_SSTORE(RS, cur ^ 1, 2);                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        HK_BW_MFMA(a1, bB, 3); if (STORE_) HK_BW_SSTORE(RS, cur ^ 1, 3);                                       \
        HK_STAMP(kb_, 3);                                                                                      \
        __syncthreads();                                                                                       \
        HK_STAMP(kb_, 4);                                                                                      \
    } while (0)

    // prologue: K-block 0 staged
#pragma unroll
    for (int part = 0; part < 4; ++part) HK_BW_GLOAD(R0, 0, part);
#pragma unroll
    for (int part = 0; part < 4; ++part) HK_BW_SSTORE(R0, 0, part);
    __syncthreads();
    int kb = 0;
    for (; kb + 1 < nkb; ++kb) HK_BW_KBLOCK(kb, R0, R0, !(LABV & 1), !(LABV & 1));     // steady state
    HK_BW_KBLOCK(kb, R0, R0, false, false);                             // last block: nothing left to stage
#undef HK_BW_KBLOCK
#undef HK_BW_MFMA
#undef HK_BW_BFRAG
#undef HK_BW_AFRAG
#undef HK_BW_GLOAD
#undef HK_BW_SSTORE

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float* dxb = dx + (long long)b * C * HW + (long long)(I * IB + wrow + i * 16 + lq * 4) * HW;
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            const int col = 16 * (nt0 + n) + l15;
            if (n < nloc && col < HW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dxb[(long long)r * HW + col] = acc[i][n][r];
            }
        }
    }
    if (MODE == 0) {                                           // t partials: slot 2I (+ a zero in 2I + 1: the consumer
        __syncthreads();                                       // adds C / 64 slots per image)
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

template <int HW, int MODE, int RB = 2>
static inline size_t bwd128_lds_bytes() {
    constexpr bool HAS_W = MODE == 0 || MODE == 3;
    constexpr bool HAS_S2 = MODE != 2;
    constexpr int IB = 64 * RB;
    constexpr int stage = IB * 36 + (HAS_W ? IB * 36 : 0) + (HAS_S2 ? 32 * (IB + 4) : 0) + ((32 * HW / 4 + 3) / 4 * 16 + 16);
    return (size_t)2 * stage * sizeof(float);
}

// HK_ERR_UNSUPPORTED unless C % (64 RB) == 0 (the caller then takes the 4-wave 64-row kernel)
template <int HW, int MODE, int RB = 2>
static int bwd128_launch(const float* x, const float* y, const float* dy, const float* inv_norm, float* dx, float* tpart,
                         int B, int C, const BwdExtra& ex_in, hipStream_t st) {
    if (C % (64 * RB) != 0) return HK_ERR_UNSUPPORTED;
    size_t lds = bwd128_lds_bytes<HW, MODE, RB>();
    BwdExtra ex = ex_in;
    ex.dc_lds = 0;
    if (MODE == 2) {                                        // dc + tables behind the two stages when they fit
        const size_t extra = ((size_t)ex.D + 4 * (size_t)C) * sizeof(float);
        if (lds + extra <= 160 * 1024) { lds += extra; ex.dc_lds = 1; }
    }
    HK_ALLOW_BIG_LDS((&bcnn_bwd128_kernel<HW, MODE, 0, RB>), lds);
    const int nI = C / (64 * RB);
#ifdef HK_LAB
    if (MODE == 0 && RB == 2 && tuning().bwd_v >= 6 && tuning().bwd_v <= 8) {
        const int lv = tuning().bwd_v - 5;
        HK_ALLOW_BIG_LDS((&bcnn_bwd128_kernel<HW, 0, 1>), lds);
        HK_ALLOW_BIG_LDS((&bcnn_bwd128_kernel<HW, 0, 2>), lds);
        HK_ALLOW_BIG_LDS((&bcnn_bwd128_kernel<HW, 0, 3>), lds);
        if (lv == 1) hipLaunchKernelGGL((bcnn_bwd128_kernel<HW, 0, 1>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm, dx, tpart, C, nI, B, ex);
        if (lv == 2) hipLaunchKernelGGL((bcnn_bwd128_kernel<HW, 0, 2>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm, dx, tpart, C, nI, B, ex);
        if (lv == 3) hipLaunchKernelGGL((bcnn_bwd128_kernel<HW, 0, 3>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm, dx, tpart, C, nI, B, ex);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
#endif
    hipLaunchKernelGGL((bcnn_bwd128_kernel<HW, MODE, 0, RB>), dim3(xcd_grid(B, nI)), dim3(512), lds, st, x, y, dy, inv_norm,
                       dx, tpart, C, nI, B, ex);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
