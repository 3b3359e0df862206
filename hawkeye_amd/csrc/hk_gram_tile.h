// The 64x64 tile step of the panel-resident Gram kernels (bcnn_fast.hip: BCNN / covariance / raw Gram; cbp_fused.hip:
// the compact-bilinear Gram with the binning fused in) and the epilogue of the kernels that write the Gram out.
#pragma once
#include "hk_common.h"

namespace hk {

// MODE 0: BCNN  y = sqrt(acc / M + 1e-5) * inv_norm[b]        MODE 1: raw  y = alpha * acc  (CBP Gram, covariance)
// MODE 2: signed sqrt (BCNN.py:23-24)  u = sign(g) sqrt(|g| + 1e-10), g = alpha * acc, written un-normalised; the thread
//         adds up u^2 of what it writes (twice for a tile that is also written mirrored) in ss - the norm's partial sums
//         come out of the Gram kernel and the separate pass over the 67 MB of G is gone
template <int MODE>
struct GramEpi {
    float* yb;       // y + b*C*C
    int C;
    int i0, j0;      // top-left of this wave's 32x32 sub-tile
    float inv, inv_m;   // MODE 1, 2: inv = alpha
    int offdiag;
    int l31, lh;
    float ss;        // MODE 2
    float* stg;      // 512 floats of LDS private to this wave: the sub-tile leaves through it, half at a time
    f32x4 t0, t1;    // a half on its way from LDS to HBM
    __device__ __forceinline__ float value(float v) {
        // v_sqrt_f32 (1 ulp, argument >= 1e-5: no denormal/negative handling needed) - parity budget is 1e-4
        float z = MODE == 0 ? __builtin_amdgcn_sqrtf(fmaf(v, inv_m, 1e-5f)) * inv : v * inv;
        if (MODE == 2) {
            // (sign(0) = 0, like torch; v_sqrt_f32 like MODE 0 - 1 ulp, argument >= 1e-10: the IEEE sqrtf expansion was ~12
            //  VALU instructions per element next to the MFMA stream)
            z = z == 0.f ? 0.f : copysignf(__builtin_amdgcn_sqrtf(fabsf(z) + 1e-10f), z);
            ss = fmaf(offdiag ? 2.f * z : z, z, ss);
        }
        return z;
    }
    // The finished sub-tile p (C layout of the 32x32 MFMA: lane (l31, lh), register r = row (r & 3) + 8 (r >> 2) + 4 lh of
    // column l31) leaves in 13 steps spread over the next tile's MFMA steps.  Round 4: every store is a 16-byte store of
    // a whole 128-byte (direct rows) / 64-byte (mirrored rows) run, the tile turned through 2 KB of wave-private LDS a
    // half at a time - 8 store instructions per tile and wave instead of 16 four-byte stores + 4 sixteen-byte stores
    // that covered 32 bytes of a row each.  With four-byte stores the kernel could not go below 28 us at ANY map size
    // (67 MB of y at 2.4 TB/s: tools/probe/gram_hw.py); the values are the same bits.
    //   0-3  the 16 values (sqrt, scale)          4 / 6   rows 0-15 / 16-31 of the tile -> LDS [16][32]
    //   5 / 7  ... read back as 16-byte pieces    6 / 8   ... stored: 8 rows x 128 B per instruction
    //   8 / 10 columns 0-15 / 16-31 of the MIRRORED tile -> LDS [32][16]; 9 / 11 read back; 10 / 12 stored: 16 rows x 64 B
    __device__ __forceinline__ void step(f32x16& p, int s) {
        const int lane = l31 + 32 * lh;
        if (s < 4) {
#pragma unroll
            for (int r = 4 * s; r < 4 * s + 4; ++r) p[r] = value(p[r]);
            return;
        }
        if (s == 6 || s == 8) {                                   // the half read one step ago: rows 16 h .. of the tile
            const int h = (s - 6) >> 1;
            float* o = yb + (long long)(i0 + 16 * h + (lane >> 3)) * C + j0 + 4 * (lane & 7);
            *reinterpret_cast<f32x4*>(o) = t0;
            *reinterpret_cast<f32x4*>(o + 8ll * C) = t1;
        }
        if ((s == 10 || s == 12) && offdiag) {                    // mirrored: rows j0 + .., columns i0 + 16 h ..
            const int h = (s - 10) >> 1;                          // (the piece this lane read back is slot msw of its row)
            float* o = yb + (long long)(j0 + (lane >> 2)) * C + i0 + 16 * h + 4 * ((lane & 3) ^ ((lane >> 3) & 3));
            *reinterpret_cast<f32x4*>(o) = t0;
            *reinterpret_cast<f32x4*>(o + 16ll * C) = t1;
        }
        if (s == 4 || s == 6) {                                   // rows (r & 3) + 8 (r >> 2) + 4 lh - 16 h, h = registers 8 h ..
            const int h = (s - 4) >> 1;
            HK_WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < 8; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = p[8 * h + r];
            HK_WAVE_SYNC();
        }
        if (s == 5 || s == 7) {
            t0 = *reinterpret_cast<const f32x4*>(stg + (lane >> 3) * 32 + 4 * (lane & 7));
            t1 = *reinterpret_cast<const f32x4*>(stg + (8 + (lane >> 3)) * 32 + 4 * (lane & 7));
        }
        if ((s == 8 || s == 10) && offdiag) {                     // T[column l31][row 8 g + 4 lh + t - 16 h]
            // A row of T is 64 bytes = four 16-byte slots; a ds_write_b128 is served in groups of 8 consecutive lanes and
            // its banks repeat every 128 bytes, so with the slots in place lanes l31 = 0, 2, 4, 6 of a group landed on the
            // same four banks (4-way: 17 % of the kernel's LDS cycles were conflict cycles, round 4).  Slot k of row l31
            // is kept at k ^ ((l31 >> 1) & 3): the eight lanes of a group cover eight different slots of a bank row; the
            // linear read-back below stays conflict-free (a row's four slots are permuted among themselves) and the
            // store above undoes the permutation in its column offset.
            const int h = (s - 8) >> 1;
            const int sw = (l31 >> 1) & 3;
            HK_WAVE_SYNC();
#pragma unroll
            for (int g = 0; g < 2; ++g)
                *reinterpret_cast<f32x4*>(stg + l31 * 16 + 4 * ((2 * g + lh) ^ sw)) =
                    (f32x4){p[8 * h + 4 * g], p[8 * h + 4 * g + 1], p[8 * h + 4 * g + 2], p[8 * h + 4 * g + 3]};
            HK_WAVE_SYNC();
        }
        if ((s == 9 || s == 11) && offdiag) {
            t0 = *reinterpret_cast<const f32x4*>(stg + (lane >> 2) * 16 + 4 * (lane & 3));
            t1 = *reinterpret_cast<const f32x4*>(stg + (16 + (lane >> 2)) * 16 + 4 * (lane & 3));
        }
    }
    static constexpr int NSTEP = 13;
};

// One 64x64 tile step for this wave's 32x32 sub-tile.  K is split over TWO independent accumulator chains (the two
// k-pairs of every 8-wide step alternate between them): a dependent f32 MFMA chain tolerates no issue slot between
// its links (MI355X_MICROARCH.md: +43 cycles for the first extra state), and the epilogue of the previous tile is
// interleaved here; with two chains the matrix pipe always has the other chain's instruction to run.
// The staged next panel (st[], fetched before the loop) is written into its free LDS buffer from INSIDE the k loop,
// one 16-B store per step in the second half of the tile, so that neither the wait for the global loads nor the LDS
// write pass sits between two tiles: the tile boundary is a bare barrier.
template <int HW, bool HASPREV, int NST, class EPI>
__device__ __forceinline__ void gram_tile(const float* Ap, const float* Bp, f32x16& acc0, f32x16& acc1, f32x16& prev,
                                          EPI& ep, int lh, const f32x4 (&st)[NST], f32x4* dst, bool do_write,
                                          int tid) {
    constexpr int KS = HW / 8;
    constexpr int N4 = 16 * HW;
    constexpr int WS = (KS - NST - 1) > 0 ? (KS - NST - 1) : 0;   // first step that writes
    // operand fragments are fetched one step ahead: the 4 MFMAs of a step block the wave's issue for ~130 cycles
    // (dependent pairs), so reads issued after them would land too late for the next step
    f32x4 a = *reinterpret_cast<const f32x4*>(Ap);
    f32x4 q = *reinterpret_cast<const f32x4*>(Bp);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f32x4 an = a, qn = q;
        if (s + 1 < KS) {
            an = *reinterpret_cast<const f32x4*>(Ap + 8 * (s + 1));
            qn = *reinterpret_cast<const f32x4*>(Bp + 8 * (s + 1));
        }
        if (s >= WS && s - WS < NST) {
            const int f = tid + 256 * (s - WS);
            if (do_write && f < N4) dst[f] = st[s - WS];
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], q[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], q[3], acc1, 0, 0, 0);
        if (HASPREV) {
            if (s < EPI::NSTEP) ep.step(prev, s);
        }
        a = an;
        q = qn;
    }
    if (HW % 8 == 4) {   // k = 8*KS .. +3: lanes 0-31 take the first two, lanes 32-63 the last two
        // (ext-vector loads, not float2: a struct-typed LDS load carries alias info and the wait-count pass then
        // parks it behind every LDS-DMA in flight - vmcnt(0) - in kernels that prefetch by LDS-DMA)
        const f32x2 a = *reinterpret_cast<const f32x2*>(Ap + 8 * KS - 2 * lh);
        const f32x2 q = *reinterpret_cast<const f32x2*>(Bp + 8 * KS - 2 * lh);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc1, 0, 0, 0);
    }
    if (HASPREV) {
#pragma unroll
        for (int s = KS; s < EPI::NSTEP; ++s) ep.step(prev, s);
    }
}

// The same tile step with the A operand in REGISTERS (round 4).  A wave's A rows are the same for every tile of a row
// block: its 32 rows x the k it feeds to the MFMAs (lane (l31, lh): k = 8 s + 4 lh + t) are HW / 2 floats per lane,
// read from the LDS panel once per row block instead of once per tile - half of the fragment traffic.  Used where the
// panel's row pitch makes the fragment reads collide (HW = 64); elsewhere it measured no gain (bcnn_fast.hip).
template <int HW>
struct GramAReg {
    static constexpr int KS = HW / 8;
    f32x4 a[KS];
    f32x2 tail;
    __device__ __forceinline__ void load(const float* Ap, int lh) {      // Ap = panel + row * HW + 4 lh, as gram_tile takes it
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const f32x4*>(Ap + 8 * s);
        if (HW % 8 == 4) tail = *reinterpret_cast<const f32x2*>(Ap + 8 * KS - 2 * lh);
    }
};
template <int HW, bool HASPREV, int NST, class EPI>
__device__ __forceinline__ void gram_tile_ra(const GramAReg<HW>& ar, const float* Bp, f32x16& acc0, f32x16& acc1,
                                             f32x16& prev, EPI& ep, int lh, const f32x4 (&st)[NST], f32x4* dst,
                                             bool do_write, int tid) {
    constexpr int KS = HW / 8;
    constexpr int N4 = 16 * HW;
    constexpr int WS = (KS - NST - 1) > 0 ? (KS - NST - 1) : 0;   // first step that writes
    f32x4 q = *reinterpret_cast<const f32x4*>(Bp);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f32x4 qn = q;
        if (s + 1 < KS) qn = *reinterpret_cast<const f32x4*>(Bp + 8 * (s + 1));
        if (s >= WS && s - WS < NST) {
            const int f = tid + 256 * (s - WS);
            if (do_write && f < N4) dst[f] = st[s - WS];
        }
        const f32x4 a = ar.a[s];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], q[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], q[3], acc1, 0, 0, 0);
        if (HASPREV) {
            if (s < EPI::NSTEP) ep.step(prev, s);
        }
        q = qn;
    }
    if (HW % 8 == 4) {   // k = 8*KS .. +3: lanes 0-31 take the first two, lanes 32-63 the last two
        const f32x2 qt = *reinterpret_cast<const f32x2*>(Bp + 8 * KS - 2 * lh);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar.tail[0], qt[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar.tail[1], qt[1], acc1, 0, 0, 0);
    }
    if (HASPREV) {
#pragma unroll
        for (int s = KS; s < EPI::NSTEP; ++s) ep.step(prev, s);
    }
}

}  // namespace hk
