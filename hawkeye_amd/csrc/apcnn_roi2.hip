// ROI-refinement backward for maps up to 64x64 (get_roi_crop_feat's autograd backward, model/methods/APCNN.py:478-531):
//   dX[crop pixel] = rate * sum over the output pixels that sampled it of  ly * lx * dY      (bilinear resize, transposed)
// History on the AP-CNN shape (16 x 512 maps of 56x56 = 205 MB read + written; HBM time ~35 us):
//   roi_crop_bwd_tab_kernel (apcnn.hip)   tables per workgroup, taps read from global memory                  187-204 us
//   LDS-staged map, 4 maps per workgroup   same loops on LDS                                                   144 us
//   uniform KY x KX window, 4 pixels in flight, geometry once per 8 maps                                       132 us
// Cycle stamps inside that third version (tools/roi_lab.py, profiles/r2_roi_lab.json) showed where the time went: the
// gather of one map took ~8-16k cycles for a few hundred LDS reads per thread - one exposed LDS latency per tap (the tap
// loops have data-dependent trip counts, every read sits in its own basic block behind an s_waitcnt), all H x W pixels
// of the map walked the loops although only the crop receives gradient, and the per-workgroup prologue (tables by
// serial scans, 13 integer divisions per thread) was another ~16k cycles.  This version:
//   * only CROP pixels are gathered: slot k of a thread is crop pixel q = tid + 256 k (row q / cw, column q % cw);
//     the rest of the map is zero and stays zero in an LDS image of the output map (zeroed once per workgroup), which
//     is written to HBM with 16-byte stores after every map;
//   * the tap window has the SAME size for every pixel of the workgroup and, when it is at most 3 x 3 or 4 x 4 (crops of
//     at least ~28 / ~19 pixels a side of a 56-pixel map), the loops are compile-time: a pixel's KW + KW table weights
//     live in registers for all maps of the workgroup and its KW * KW taps are unconditional LDS reads, all in flight
//     together.  The window is shifted to stay inside the map, taps outside a pixel's true range carry table weight 0
//     and read finite map values, so dX is bit-identical to the table kernel (x + 0 * finite = x).  Small crops (up
//     to 512 pixels) have windows up to 8 x 8 the same way, two pixels per thread, mid-size crops (up to 1536 pixels)
//     windows up to 6 x 6 with six pixels per thread; anything else takes the table loops - separable for small crops
//     (a 12 x 12 instance was tried: its 288 reads in flight push the whole kernel to one workgroup per CU);
//   * the tables' non-zero ranges come from LDS atomics while the tables are filled (no scans), pixel coordinates from
//     a multiply-shift instead of integer division;
//   * 8-32 maps per workgroup (chosen so that the grid is at most two workgroups per CU: one round, no tail), each map
//     staged into LDS with 16-byte loads, the next map's loads issued before the current map's gather.
#include "hk_roi.h"
#include "../../include/hawkeye_hip.h"

namespace hk {


// PPT (template): crop pixels per thread, ceil(H * W / 256) rounded up to an instance - 4 (maps up to 32 x 32), 13 (56 x 56:
// 255 registers, two workgroups per CU), 16 (64 x 64: one workgroup per CU)

constexpr int ROI_TROW_CW = 44;     // widest crop of the table path: a window of more than 4 taps <=> scale > 1.5 <=> cw < W / 1.5 <= 43

struct RoiShared {
    CropGeom g;
    int ylo[64], yhi[64], xlo[64], xhi[64];
    int kmax[2];
};

// One workgroup: image b, maps c0 .. c0 + nmaps - 1.  KW = 3 / 4: compile-time KW x KW window with register weights
// (KY = KX = KW on entry); 0: table loops over the KY x KX window
// trow (KW = 0 only): [H][cw] floats of LDS for the row sums of the separable table path
template <int KW, int ROI2_PPT>
__device__ __forceinline__ void roi_bwd_maps(const float* __restrict__ dy, float* __restrict__ dx, const RoiShared& sh,
                                             const float* wy, const float* wx, float* smap, float* omap, int b, int C,
                                             int c0, int nmaps, int H, int W, int KY, int KX, int training,
                                             float* trow = nullptr) {
    constexpr bool REGW = KW > 0;
    constexpr int KR = REGW ? KW : 1;
    const int tid = threadIdx.x, hw = H * W;
    const CropGeom g = sh.g;
    const int ncrop = g.ch * g.cw;                         // (> 0 here)
    const int nk = __builtin_amdgcn_readfirstlane((ncrop + 255) >> 8);   // slots in use (uniform: scalar branches below)
    const int inv = ((1 << 20) + g.cw - 1) / g.cw;         // q / cw = (q * inv) >> 20, exact for q < 4096, cw <= 64
    // table path: separable (row sums once, then KY terms per pixel) for SMALL crops, whose windows are the widest and
    // whose pixels leave most threads idle - 9 x 11: 196 -> 57 us; on larger crops with 5 .. 8 taps and on strips the
    // per-pixel loops below measured faster (the AP-CNN step's own crop: 78 us against 100-122 us separable)
    const bool sepr = !REGW && ncrop <= 512 && g.cw <= ROI_TROW_CW;

    // geometry of slot k: offset of the window origin in the staged map (< 0: the pixel receives no gradient), offset of
    // the pixel in the output map, and either the weights (REGW) or the offsets of the window origin in the tables
    int smo[ROI2_PPT], omo[ROI2_PPT];
    int wyo[REGW ? 1 : ROI2_PPT], wxo[REGW ? 1 : ROI2_PPT];
    int toy[REGW ? 1 : ROI2_PPT], trx[REGW ? 1 : ROI2_PPT];  // table path: the window's first output row, the crop column
    float rwy[REGW ? ROI2_PPT : 1][KR], rwx[REGW ? ROI2_PPT : 1][KR];
#pragma unroll
    for (int k = 0; k < ROI2_PPT; ++k) {
        const int q = tid + 256 * k;
        int wyo_k = 0, wxo_k = 0;
        smo[k] = -1;
        omo[k] = 0;
        if (q < ncrop) {
            const int ry = (q * inv) >> 20, rx = q - ry * g.cw;
            const int iy = ry + g.y1, ix = rx + g.x1;
            const bool dropped = training && iy >= g.dy1 && iy < g.dy2 && ix >= g.dx1 && ix < g.dx2;
            if (!dropped && sh.yhi[ry] >= sh.ylo[ry] && sh.xhi[rx] >= sh.xlo[rx]) {
                int oy0 = sh.ylo[ry], ox0 = sh.xlo[rx];
                if (oy0 > H - KY) oy0 = H - KY;              // keep the window inside the map: the taps added on the
                if (ox0 > W - KX) ox0 = W - KX;              // low side have table weight 0
                wyo_k = ry * 65 + oy0;
                wxo_k = rx * 65 + ox0;
                smo[k] = oy0 * W + ox0;
                omo[k] = iy * W + ix;
            }
        }
        if (REGW) {                                         // (entries behind the true range are table zeros)
#pragma unroll
            for (int a = 0; a < KR; ++a) {
                rwy[k][a] = wy[wyo_k + a];
                rwx[k][a] = wx[wxo_k + a];
            }
        } else {
            wyo[k] = wyo_k;
            wxo[k] = wxo_k;
            toy[k] = wyo_k % 65;
            trx[k] = wxo_k / 65;
        }
    }

    const bool vec = (hw % 4 == 0) && ((((uintptr_t)dy) & 15) == 0) && ((((uintptr_t)dx) & 15) == 0);
    const int n4 = hw / 4;
    // staged next map (vec): 4 x 256 float4 >= 64 x 64 / 4.  Named registers - an array that is loaded in one block and
    // stored in another is not promoted to registers (it went to scratch: 80 bytes per lane, a round trip per map)
    float4 st0, st1, st2, st3;
    st0 = st1 = st2 = st3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int f0 = tid < n4 ? tid : n4 - 1, f1 = tid + 256 < n4 ? tid + 256 : n4 - 1;
    const int f2 = tid + 512 < n4 ? tid + 512 : n4 - 1, f3 = tid + 768 < n4 ? tid + 768 : n4 - 1;
    auto issue = [&](int cc) {
        const float4* gp4 = reinterpret_cast<const float4*>(dy + ((long long)b * C + c0 + cc) * hw);
        st0 = gp4[f0]; st1 = gp4[f1]; st2 = gp4[f2]; st3 = gp4[f3];
    };
    if (vec) issue(0);
    for (int cc = 0; cc < nmaps; ++cc) {
        const float* gp = dy + ((long long)b * C + c0 + cc) * hw;
        float* dp = dx + ((long long)b * C + c0 + cc) * hw;
        __syncthreads();                                   // previous map: gathered by everyone, output image written out
        if (vec) {
            float4* s4 = reinterpret_cast<float4*>(smap);
            if (tid < n4) s4[tid] = st0;
            if (tid + 256 < n4) s4[tid + 256] = st1;
            if (tid + 512 < n4) s4[tid + 512] = st2;
            if (tid + 768 < n4) s4[tid + 768] = st3;
        } else {
            for (int p = tid; p < hw; p += 256) smap[p] = gp[p];
        }
        __syncthreads();
        if (vec && cc + 1 < nmaps) issue(cc + 1);          // in flight during the gather below
        if (REGW) {
            // two slots per (uniform) block: both slots' KW * KW reads are in flight before the first is used
#pragma unroll
            for (int k0 = 0; k0 < ROI2_PPT; k0 += 2) {
                if (k0 < nk) {
                    float res[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = k0 + j < ROI2_PPT ? k0 + j : k0;
                        const int so = smo[k] < 0 ? 0 : smo[k];
                        float acc = 0.f;
#pragma unroll
                        for (int a = 0; a < KR; ++a) {
                            float rowacc = 0.f;
#pragma unroll
                            for (int bq = 0; bq < KR; ++bq) rowacc = fmaf(rwx[k][bq], smap[so + a * W + bq], rowacc);
                            acc = fmaf(rwy[k][a], rowacc, acc); // (explicit fmaf: bit-identical to the table kernel)
                        }
                        res[j] = acc * g.rate;
                    }
                    if (smo[k0] >= 0) omap[omo[k0]] = res[0];
                    if (k0 + 1 < ROI2_PPT && smo[k0 + 1 < ROI2_PPT ? k0 + 1 : k0] >= 0) omap[omo[k0 + 1 < ROI2_PPT ? k0 + 1 : k0]] = res[1];
                }
            }
        } else {
            // Table path (windows wider than 8 taps, or wider than 4 on a large crop), SEPARABLE: a pixel's gradient is
            //     rate * sum_a wy[ry][oy0 + a] * ( sum_b wx[rx][ox0 + b] * dY[oy0 + a][ox0 + b] )
            // and the inner row sum depends on (output row, crop column) only - it used to be recomputed by every pixel
            // of the column (KY x KX dependent LDS round trips per pixel on ch x cw of the 256 threads: 196-424 us for a
            // 9 x 11 crop).  Pass 1: all threads form the H x cw row sums once; pass 2: KY terms per pixel.  The same
            // fmaf chains in the same order as before: bit-identical.
            if (sepr) {
                // (trow holds ROI_TROW_CW columns; sepr implies cw <= ROI_TROW_CW, the loop body runs once - it is written
                //  for column chunks so that the bound can be lifted)
                for (int cx0 = 0; cx0 < g.cw; cx0 += ROI_TROW_CW) {
                    const int cwc = g.cw - cx0 < ROI_TROW_CW ? g.cw - cx0 : ROI_TROW_CW;
                    const int invc = ((1 << 20) + cwc - 1) / cwc;
                    const int ntr = H * cwc;
                    if (cx0 > 0) __syncthreads();              // the previous chunk's row sums have been consumed
                    // four row sums / four pixels at a time: independent fmaf chains, their LDS reads in flight together
                    for (int e0 = tid; e0 < ntr; e0 += 1024) {
                        const float* wrow[4];
                        const float* srow[4];
                        float rowacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int e = e0 + 256 * j;
                            e = e < ntr ? e : ntr - 1;
                            const int oy = (e * invc) >> 20, rx = cx0 + e - oy * cwc;
                            int ox0 = sh.xhi[rx] >= sh.xlo[rx] ? sh.xlo[rx] : 0;        // (no range: all-zero table row)
                            if (ox0 > W - KX) ox0 = W - KX;
                            wrow[j] = wx + rx * 65 + ox0;
                            srow[j] = smap + oy * W + ox0;
                        }
                        for (int bq = 0; bq < KX; ++bq) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) rowacc[j] = fmaf(wrow[j][bq], srow[j][bq], rowacc[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (e0 + 256 * j < ntr) trow[e0 + 256 * j] = rowacc[j];
                    }
                    __syncthreads();
#pragma unroll
                    for (int k0 = 0; k0 < ROI2_PPT; k0 += 4) {
                        if (k0 < nk) {                           // uniform
                            const float* wcol[4];
                            const float* tcol[4];
                            float acc[4] = {0.f, 0.f, 0.f, 0.f};
                            bool on[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int k = k0 + j < ROI2_PPT ? k0 + j : k0;   // (PPT = 13: slots 13..15 do not exist)
                                on[j] = k0 + j < ROI2_PPT && smo[k] >= 0 && trx[k] >= cx0 && trx[k] < cx0 + cwc;
                                wcol[j] = wy + (on[j] ? wyo[k] : 0);
                                tcol[j] = trow + (on[j] ? toy[k] * cwc + (trx[k] - cx0) : 0);
                            }
                            for (int a = 0; a < KY; ++a) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[j] = fmaf(wcol[j][a], tcol[j][a * cwc], acc[j]);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int k = k0 + j < ROI2_PPT ? k0 + j : k0;
                                if (on[j]) omap[omo[k]] = acc[j] * g.rate;
                            }
                        }
                    }
                }
            } else {
                // larger crops with 5 .. 8 taps, strips: per pixel KY x KX taps, four pixels in flight
#pragma unroll
                for (int k0 = 0; k0 < ROI2_PPT; k0 += 4) {
                    if (k0 < nk) {                               // uniform
                        float acc[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int a = 0; a < KY; ++a) {
                            float rowacc[4] = {0.f, 0.f, 0.f, 0.f};
                            for (int bq = 0; bq < KX; ++bq) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int k = k0 + j < ROI2_PPT ? k0 + j : k0;   // (PPT = 13: slots 13..15 do not exist)
                                    const int so = smo[k] < 0 ? 0 : smo[k];
                                    rowacc[j] = fmaf(wx[wxo[k] + bq], smap[so + a * W + bq], rowacc[j]);
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int k = k0 + j < ROI2_PPT ? k0 + j : k0;
                                acc[j] = fmaf(wy[wyo[k] + a], rowacc[j], acc[j]);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = k0 + j < ROI2_PPT ? k0 + j : k0;
                            if (k0 + j < ROI2_PPT && smo[k] >= 0) omap[omo[k]] = acc[j] * g.rate;
                        }
                    }
                }
            }
        }
        __syncthreads();                                   // output image complete
        if (vec) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = tid + 256 * u;
                if (f < n4) reinterpret_cast<float4*>(dp)[f] = reinterpret_cast<const float4*>(omap)[f];
            }
        } else {
            for (int p = tid; p < hw; p += 256) dp[p] = omap[p];
        }
    }
}

template <int PPT>
__global__ __launch_bounds__(256) void roi_crop_bwd_tab3_kernel(const float* __restrict__ dy, const float* __restrict__ box,
                                                                const float* __restrict__ drop, float* __restrict__ dx,
                                                                int C, int H, int W, int training, int cpb) {
    HK_DYN_LDS16(lds);                                     // wy, wx [64 * 65 + 64], smap, omap [64 * 64], trow [64 * 44]: 77.5 KB
    __shared__ RoiShared sh;
    float* wy = lds;
    float* wx = wy + 64 * 65 + 64;
    float* smap = wx + 64 * 65 + 64;
    float* omap = smap + 64 * 64;
    float* trow = omap + 64 * 64;                          // (table path only: crops up to ROI_TROW_CW columns)
    const int b = blockIdx.y, c0 = blockIdx.x * cpb, tid = threadIdx.x;
    const int hw = H * W;
    if (tid == 0) {
        sh.g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
        sh.kmax[0] = sh.kmax[1] = 1;
    }
    {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < (64 * 65 + 64) / 4; e += 256) {
            reinterpret_cast<float4*>(wy)[e] = z;
            reinterpret_cast<float4*>(wx)[e] = z;
        }
        for (int e = tid; e < 64 * 64 / 4; e += 256) reinterpret_cast<float4*>(omap)[e] = z;
        if (tid < 64) { sh.ylo[tid] = H; sh.yhi[tid] = -1; sh.xlo[tid] = W; sh.xhi[tid] = -1; }
    }
    __syncthreads();
    const CropGeom g = sh.g;
    const int nmaps = (C - c0) < cpb ? (C - c0) : cpb;
    if (g.ch > 0 && g.cw > 0) {
        // table rows = source pixels of the crop, columns = output pixels; a row's non-zero range by LDS atomics
        if (tid < H) {                                   // output row tid contributes to source rows i0, i1
            int i0, i1; float l0, l1;
            src_index(g.sh, tid, g.ch, i0, i1, l0, l1);
            wy[i0 * 65 + tid] += l0;
            wy[i1 * 65 + tid] += l1;
            if (l0 != 0.f) { atomicMin(&sh.ylo[i0], tid); atomicMax(&sh.yhi[i0], tid); }
            if (l1 != 0.f) { atomicMin(&sh.ylo[i1], tid); atomicMax(&sh.yhi[i1], tid); }
        } else if (tid >= 64 && tid - 64 < W) {
            const int ox = tid - 64;
            int i0, i1; float l0, l1;
            src_index(g.sw, ox, g.cw, i0, i1, l0, l1);
            wx[i0 * 65 + ox] += l0;
            wx[i1 * 65 + ox] += l1;
            if (l0 != 0.f) { atomicMin(&sh.xlo[i0], ox); atomicMax(&sh.xhi[i0], ox); }
            if (l1 != 0.f) { atomicMin(&sh.xlo[i1], ox); atomicMax(&sh.xhi[i1], ox); }
        }
        __syncthreads();
        if (tid < 64) atomicMax(&sh.kmax[0], sh.yhi[tid] - sh.ylo[tid] + 1);      // largest tap range of any table row
        else if (tid < 128) atomicMax(&sh.kmax[1], sh.xhi[tid - 64] - sh.xlo[tid - 64] + 1);
        __syncthreads();
        int KY = sh.kmax[0], KX = sh.kmax[1];
        KY = KY < H ? KY : H;
        KX = KX < W ? KX : W;
        // (a map smaller than the compile-time window cannot take it: the window has to fit inside the map)
        if (KY <= 3 && KX <= 3 && H >= 3 && W >= 3)
            roi_bwd_maps<3, PPT>(dy, dx, sh, wy, wx, smap, omap, b, C, c0, nmaps, H, W, 3, 3, training);
        else if (KY <= 4 && KX <= 4 && H >= 4 && W >= 4)
            roi_bwd_maps<4, PPT>(dy, dx, sh, wy, wx, smap, omap, b, C, c0, nmaps, H, W, 4, 4, training);
        else if (KY <= 8 && KX <= 8 && H >= 8 && W >= 8 && g.ch * g.cw <= 512)
            // small crops: windows up to 8 x 8, but at most two pixels per thread
            roi_bwd_maps<8, 2>(dy, dx, sh, wy, wx, smap, omap, b, C, c0, nmaps, H, W, 8, 8, training);
        else if (KY <= 6 && KX <= 6 && H >= 6 && W >= 6 && g.ch * g.cw <= 1536)
            // mid-size crops (5 or 6 taps: 19 .. 37 pixels a side of a 56-pixel map), up to six pixels per thread
            roi_bwd_maps<6, 6>(dy, dx, sh, wy, wx, smap, omap, b, C, c0, nmaps, H, W, 6, 6, training);
        else      // (a window wider than 4 taps means a crop narrower than W / 1.5: its columns fit trow)
            roi_bwd_maps<0, PPT>(dy, dx, sh, wy, wx, smap, omap, b, C, c0, nmaps, H, W, KY, KX, training, trow);
    } else {                                               // empty crop: the whole map is zero
        for (int cc = 0; cc < nmaps; ++cc) {
            float* dp = dx + ((long long)b * C + c0 + cc) * hw;
            for (int p = tid; p < hw; p += 256) dp[p] = 0.f;
        }
    }
}

int roi_crop_bwd_v2(const float* dy, const float* box, const float* drop, float* dx, int B, int C, int H, int W,
                    int training, hipStream_t st) {
    if (H > 64 || W > 64) return HK_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(2 * (64 * 65 + 64) + 2 * 64 * 64 + 64 * ROI_TROW_CW) * sizeof(float);
    HK_ALLOW_BIG_LDS(&roi_crop_bwd_tab3_kernel<4>, lds);
    HK_ALLOW_BIG_LDS(&roi_crop_bwd_tab3_kernel<13>, lds);
    HK_ALLOW_BIG_LDS(&roi_crop_bwd_tab3_kernel<16>, lds);
    // maps per workgroup: the kernel holds 2 workgroups per CU (registers, LDS), so up to 512 run at once; a grid just
    // above that leaves a mostly idle second round
    int cpb = 8;
    while (cpb < 32 && (long long)B * ((C + cpb - 1) / cpb) > 512) cpb *= 2;
    const dim3 grid((C + cpb - 1) / cpb, B);
    if (H * W <= 256 * 4)
        hipLaunchKernelGGL(roi_crop_bwd_tab3_kernel<4>, grid, dim3(256), lds, st, dy, box, drop, dx, C, H, W, training, cpb);
    else if (H * W <= 256 * 13)
        hipLaunchKernelGGL(roi_crop_bwd_tab3_kernel<13>, grid, dim3(256), lds, st, dy, box, drop, dx, C, H, W, training, cpb);
    else
        hipLaunchKernelGGL(roi_crop_bwd_tab3_kernel<16>, grid, dim3(256), lds, st, dy, box, drop, dx, C, H, W, training, cpb);
    HK_LAUNCH_CHECK();
    return HK_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Forward (crop + bilinear resize to H x W + drop mask), same recipe: 8-32 maps per workgroup, the map staged in LDS with
// 16-byte loads one map ahead, an output pixel's geometry (two row offsets, column step, four weights, drop bits) in
// registers for all maps of the workgroup, its four taps unconditional LDS reads.  The round-1 kernel
// (roi_crop_fwd_tab_kernel: one map per workgroup, four 4-byte global gathers and an integer division per pixel) ran
// 73-75 us on the AP-CNN shape.  Same expression, explicit fmaf in both: bit-identical.
template <int PPT>
__global__ __launch_bounds__(256) void roi_crop_fwd_tab2_kernel(const float* __restrict__ x, const float* __restrict__ box,
                                                                const float* __restrict__ drop, float* __restrict__ y,
                                                                int C, int H, int W, int training, int cpb) {
    __shared__ CropGeom gs;
    __shared__ int ti0[2][64], ti1[2][64];                 // [0] rows, [1] columns: source taps of an output coordinate
    __shared__ float tl0[2][64], tl1[2][64];
    __shared__ __attribute__((aligned(16))) float smap[64 * 64];
    const int b = blockIdx.y, c0 = blockIdx.x * cpb, tid = threadIdx.x;
    const int hw = H * W;
    if (tid == 0) gs = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    __syncthreads();
    const CropGeom g = gs;
    const int nmaps = (C - c0) < cpb ? (C - c0) : cpb;
    if (g.cw <= 0 || g.ch <= 0) {                          // empty crop: zeros
        for (int cc = 0; cc < nmaps; ++cc) {
            float* yp = y + ((long long)b * C + c0 + cc) * hw;
            for (int p = tid; p < hw; p += 256) yp[p] = 0.f;
        }
        return;
    }
    if (tid < H) src_index(g.sh, tid, g.ch, ti0[0][tid], ti1[0][tid], tl0[0][tid], tl1[0][tid]);
    else if (tid >= 64 && tid - 64 < W) src_index(g.sw, tid - 64, g.cw, ti0[1][tid - 64], ti1[1][tid - 64], tl0[1][tid - 64], tl1[1][tid - 64]);
    __syncthreads();

    // slot k = output pixel o = tid + 256 k
    const int inv = ((1 << 20) + W - 1) / W;               // o / W = (o * inv) >> 20, exact for o < 4096, W <= 64
    int so0[PPT];                                           // offset of tap (0, 0) in the map
    unsigned pk[PPT];                                       // bit 0: column step (0 / 1), bit 1: row step (0 / W), bits 2-5: tap (i, j) not dropped
    float la0[PPT], la1[PPT], lb0[PPT], lb1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        int o = tid + 256 * k;
        if (o >= hw) o = hw - 1;
        const int oy = (o * inv) >> 20, ox = o - oy * W;
        const int ys0 = g.y1 + ti0[0][oy], ys1 = g.y1 + ti1[0][oy], xs0 = g.x1 + ti0[1][ox], xs1 = g.x1 + ti1[1][ox];
        la0[k] = tl0[0][oy]; la1[k] = tl1[0][oy]; lb0[k] = tl0[1][ox]; lb1[k] = tl1[1][ox];
        so0[k] = ys0 * W + xs0;
        unsigned m = 15u;
        if (training) {
            const bool dy0 = ys0 >= g.dy1 && ys0 < g.dy2, dy1 = ys1 >= g.dy1 && ys1 < g.dy2;
            const bool dx0 = xs0 >= g.dx1 && xs0 < g.dx2, dx1 = xs1 >= g.dx1 && xs1 < g.dx2;
            m = (dy0 && dx0 ? 0u : 1u) | (dy0 && dx1 ? 0u : 2u) | (dy1 && dx0 ? 0u : 4u) | (dy1 && dx1 ? 0u : 8u);
        }
        pk[k] = (unsigned)(xs1 - xs0) | ((unsigned)(ys1 - ys0) << 1) | (m << 2);
    }

    const bool vec = (hw % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    const int n4 = hw / 4;
    float4 st0, st1, st2, st3;                             // named: see the backward
    st0 = st1 = st2 = st3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int f0 = tid < n4 ? tid : n4 - 1, f1 = tid + 256 < n4 ? tid + 256 : n4 - 1;
    const int f2 = tid + 512 < n4 ? tid + 512 : n4 - 1, f3 = tid + 768 < n4 ? tid + 768 : n4 - 1;
    auto issue = [&](int cc) {
        const float4* gp4 = reinterpret_cast<const float4*>(x + ((long long)b * C + c0 + cc) * hw);
        st0 = gp4[f0]; st1 = gp4[f1]; st2 = gp4[f2]; st3 = gp4[f3];
    };
    if (vec) issue(0);
    for (int cc = 0; cc < nmaps; ++cc) {
        const float* xp = x + ((long long)b * C + c0 + cc) * hw;
        float* yp = y + ((long long)b * C + c0 + cc) * hw;
        __syncthreads();                                   // previous map no longer read
        if (vec) {
            float4* s4 = reinterpret_cast<float4*>(smap);
            if (tid < n4) s4[tid] = st0;
            if (tid + 256 < n4) s4[tid + 256] = st1;
            if (tid + 512 < n4) s4[tid + 512] = st2;
            if (tid + 768 < n4) s4[tid + 768] = st3;
        } else {
            for (int p = tid; p < hw; p += 256) smap[p] = xp[p];
        }
        __syncthreads();
        if (vec && cc + 1 < nmaps) issue(cc + 1);          // in flight during the gather below
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int o = tid + 256 * k;
            const int dxs = pk[k] & 1u, so1 = so0[k] + ((pk[k] & 2u) ? W : 0);
            float v00 = smap[so0[k]], v01 = smap[so0[k] + dxs], v10 = smap[so1], v11 = smap[so1 + dxs];
            if (training) {
                v00 = ((pk[k] & 4u) ? v00 : 0.f) * g.rate;
                v01 = ((pk[k] & 8u) ? v01 : 0.f) * g.rate;
                v10 = ((pk[k] & 16u) ? v10 : 0.f) * g.rate;
                v11 = ((pk[k] & 32u) ? v11 : 0.f) * g.rate;
            }
            const float r0 = fmaf(lb1[k], v01, lb0[k] * v00), r1 = fmaf(lb1[k], v11, lb0[k] * v10);
            if (o < hw) yp[o] = fmaf(la1[k], r1, la0[k] * r0);
        }
    }
}

int roi_crop_fwd_v2(const float* x, const float* box, const float* drop, float* y, int B, int C, int H, int W, int training,
                    hipStream_t st) {
    if (H > 64 || W > 64) return HK_ERR_UNSUPPORTED;
    int cpb = 8;                                            // (190 registers at 56 x 56: two workgroups per CU, 512 run at once)
    while (cpb < 32 && (long long)B * ((C + cpb - 1) / cpb) > 512) cpb *= 2;
    const dim3 grid((C + cpb - 1) / cpb, B);
    if (H * W <= 256 * 4)
        hipLaunchKernelGGL(roi_crop_fwd_tab2_kernel<4>, grid, dim3(256), 0, st, x, box, drop, y, C, H, W, training, cpb);
    else if (H * W <= 256 * 13)
        hipLaunchKernelGGL(roi_crop_fwd_tab2_kernel<13>, grid, dim3(256), 0, st, x, box, drop, y, C, H, W, training, cpb);
    else
        hipLaunchKernelGGL(roi_crop_fwd_tab2_kernel<16>, grid, dim3(256), 0, st, x, box, drop, y, C, H, W, training, cpb);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk

