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
    __device__ __forceinline__ float direct(float v, int r) {
        // v_sqrt_f32 (1 ulp, argument >= 1e-5: no denormal/negative handling needed) - parity budget is 1e-4
        float z = MODE == 0 ? __builtin_amdgcn_sqrtf(fmaf(v, inv_m, 1e-5f)) * inv : v * inv;
        if (MODE == 2) {
            z = z == 0.f ? 0.f : copysignf(sqrtf(fabsf(z) + 1e-10f), z);      // (sign(0) = 0, like torch)
            ss = fmaf(offdiag ? 2.f * z : z, z, ss);
        }
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        yb[(long long)i * C + j0 + l31] = z;
        return z;
    }
    __device__ __forceinline__ void mirror(const f32x16& p, int g) {   // rows 8g+4lh .. +3 of column l31 -> y[j][i..i+3]
        if (!offdiag) return;
        const float4 q = make_float4(p[4 * g], p[4 * g + 1], p[4 * g + 2], p[4 * g + 3]);
        *reinterpret_cast<float4*>(&yb[(long long)(j0 + l31) * C + i0 + 8 * g + 4 * lh]) = q;
    }
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
            if (s < 16) prev[s] = ep.direct(prev[s], s);
            else if (s < 20) ep.mirror(prev, s - 16);
        }
        a = an;
        q = qn;
    }
    if (HW % 8 == 4) {   // k = 8*KS .. +3: lanes 0-31 take the first two, lanes 32-63 the last two
        const float2 a = *reinterpret_cast<const float2*>(Ap + 8 * KS - 2 * lh);
        const float2 q = *reinterpret_cast<const float2*>(Bp + 8 * KS - 2 * lh);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q.y, acc1, 0, 0, 0);
    }
    if (HASPREV) {
#pragma unroll
        for (int s = KS; s < 20; ++s) {
            if (s < 16) prev[s] = ep.direct(prev[s], s);
            else ep.mirror(prev, s - 16);
        }
    }
}

}  // namespace hk
