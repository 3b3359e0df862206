// ROI-refinement backward for maps up to 64x64 (get_roi_crop_feat's autograd backward, model/methods/APCNN.py:478-531):
//   dX[crop pixel] = rate * sum over the output pixels that sampled it of  ly * lx * dY      (bilinear resize, transposed)
// History on the AP-CNN shape (16 x 512 maps of 56x56 = 205 MB read + written; HBM time ~35 us):
//   roi_crop_bwd_tab_kernel (apcnn.hip)   tables per workgroup, taps read from global memory          187-201 us
//   LDS-staged map, 4 maps per workgroup   same loops on LDS                                           144 us
// Both are LATENCY bound, not bandwidth bound: every source pixel runs a doubly nested loop with data-dependent trip
// counts (its own tap range), one dependent LDS/global read per tap, one pixel at a time per thread, and the index
// arithmetic of all 3136 pixels is redone for every channel map.  This kernel keeps the taps and their order
// (bit-identical dX) and removes the serialisation:
//   * the tap window has the SAME size KY x KX for every pixel of the workgroup (the largest range of the tables;
//     windows are shifted to stay inside the map, the extra taps carry table weight 0), so the loops are uniform and
//     FOUR pixels per thread run through them together - four independent accumulation chains instead of one;
//   * a pixel's geometry (table rows, window origin, inside-crop / dropped flags) is computed once per workgroup and
//     kept in registers for all ROI2_CPB = 8 channel maps;
//   * each map is staged into LDS with 16-byte loads, the next map's loads are issued before the current map's gather.
#include "hk_roi.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

constexpr int ROI2_CPB = 8;      // channel maps per workgroup
constexpr int ROI2_PPT = 13;     // pixels per thread: ceil(64 * 64 / 256) would be 16; 56 x 56 needs 13

__global__ __launch_bounds__(256) void roi_crop_bwd_tab3_kernel(const float* __restrict__ dy, const float* __restrict__ box,
                                                                const float* __restrict__ drop, float* __restrict__ dx,
                                                                int C, int H, int W, int training) {
    __shared__ CropGeom g;
    __shared__ float wy[64 * 65 + 64], wx[64 * 65 + 64];
    __shared__ __attribute__((aligned(16))) float smap[64 * 64];
    __shared__ int ylo[64], yhi[64], xlo[64], xhi[64];
    const int b = blockIdx.y, c0 = blockIdx.x * ROI2_CPB, tid = threadIdx.x;
    if (tid == 0) g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    for (int e = tid; e < 64 * 65 + 64; e += 256) { wy[e] = 0.f; wx[e] = 0.f; }
    __syncthreads();
    if (g.ch > 0 && g.cw > 0) {
        if (tid < H) {                                   // output row tid contributes to source rows i0, i1
            int i0, i1; float l0, l1;
            src_index(g.sh, tid, g.ch, i0, i1, l0, l1);
            wy[i0 * 65 + tid] += l0;
            wy[i1 * 65 + tid] += l1;
        } else if (tid >= 64 && tid - 64 < W) {
            const int ox = tid - 64;
            int i0, i1; float l0, l1;
            src_index(g.sw, ox, g.cw, i0, i1, l0, l1);
            wx[i0 * 65 + ox] += l0;
            wx[i1 * 65 + ox] += l1;
        }
    }
    __syncthreads();
    if (tid < 64) {                                       // non-zero range of each table row
        int lo = H, hi = -1;
        for (int o = 0; o < H; ++o)
            if (wy[tid * 65 + o] != 0.f) { lo = o < lo ? o : lo; hi = o; }
        ylo[tid] = lo; yhi[tid] = hi;
    } else if (tid < 128) {
        const int r = tid - 64;
        int lo = W, hi = -1;
        for (int o = 0; o < W; ++o)
            if (wx[r * 65 + o] != 0.f) { lo = o < lo ? o : lo; hi = o; }
        xlo[r] = lo; xhi[r] = hi;
    }
    __syncthreads();
    int KY = 1, KX = 1;                                   // largest tap range of any table row (every thread for itself)
    for (int r = 0; r < 64; ++r) {
        const int ny = yhi[r] - ylo[r] + 1, nx = xhi[r] - xlo[r] + 1;
        KY = ny > KY ? ny : KY;
        KX = nx > KX ? nx : KX;
    }
    KY = KY < H ? KY : H;
    KX = KX < W ? KX : W;
    const int hw = H * W;

    // geometry of this thread's pixels p = tid + 256 k : offsets of the window origin in the tables and in the map
    int wyo[ROI2_PPT], wxo[ROI2_PPT], smo[ROI2_PPT];      // < 0 in smo: the pixel receives no gradient
#pragma unroll
    for (int k = 0; k < ROI2_PPT; ++k) {
        const int p = tid + 256 * k;
        wyo[k] = wxo[k] = 0;
        smo[k] = -1;
        if (p < hw) {
            const int iy = p / W, ix = p % W;
            const int ry = iy - g.y1, rx = ix - g.x1;
            if (g.cw > 0 && g.ch > 0 && ry >= 0 && ry < g.ch && rx >= 0 && rx < g.cw) {
                const bool dropped = training && iy >= g.dy1 && iy < g.dy2 && ix >= g.dx1 && ix < g.dx2;
                if (!dropped && yhi[ry] >= ylo[ry] && xhi[rx] >= xlo[rx]) {
                    int oy0 = ylo[ry], ox0 = xlo[rx];
                    if (oy0 > H - KY) oy0 = H - KY;          // keep the window inside the map: the taps added on the
                    if (ox0 > W - KX) ox0 = W - KX;          // low side have table weight 0
                    wyo[k] = ry * 65 + oy0;
                    wxo[k] = rx * 65 + ox0;
                    smo[k] = oy0 * W + ox0;
                }
            }
        }
    }

    const bool vec = (hw % 4 == 0) && ((((uintptr_t)dy) & 15) == 0);
    const int n4 = hw / 4;
    float4 st[4];                                          // staged next map (vec) - 4 x 256 float4 >= 64 x 64 / 4
    auto issue = [&](int cc) {
        const float4* gp4 = reinterpret_cast<const float4*>(dy + ((long long)b * C + c0 + cc) * hw);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = tid + 256 * u;
            st[u] = gp4[f < n4 ? f : n4 - 1];
        }
    };
    const int nmaps = (C - c0) < ROI2_CPB ? (C - c0) : ROI2_CPB;
    if (vec) issue(0);
    for (int cc = 0; cc < nmaps; ++cc) {
        const float* gp = dy + ((long long)b * C + c0 + cc) * hw;
        float* dp = dx + ((long long)b * C + c0 + cc) * hw;
        __syncthreads();                                   // previous map no longer read
        if (vec) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = tid + 256 * u;
                if (f < n4) reinterpret_cast<float4*>(smap)[f] = st[u];
            }
        } else {
            for (int p = tid; p < hw; p += 256) smap[p] = gp[p];
        }
        __syncthreads();
        if (vec && cc + 1 < nmaps) issue(cc + 1);          // in flight during the gather below
#pragma unroll
        for (int k0 = 0; k0 < ROI2_PPT; k0 += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int a = 0; a < KY; ++a) {
                float rowacc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int bq = 0; bq < KX; ++bq) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k0 + j < ROI2_PPT) {
                            const int so = smo[k0 + j] < 0 ? 0 : smo[k0 + j];
                            rowacc[j] += wx[wxo[k0 + j] + bq] * smap[so + a * W + bq];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ROI2_PPT) acc[j] += wy[wyo[k0 + j] + a] * rowacc[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j < ROI2_PPT) {
                    const int p = tid + 256 * (k0 + j);
                    if (p < hw) dp[p] = smo[k0 + j] < 0 ? 0.f : acc[j] * g.rate;
                }
            }
        }
    }
}

int roi_crop_bwd_v2(const float* dy, const float* box, const float* drop, float* dx, int B, int C, int H, int W,
                    int training, hipStream_t st) {
    if (H > 64 || W > 64 || H * W > 256 * ROI2_PPT) return HK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roi_crop_bwd_tab3_kernel, dim3((C + ROI2_CPB - 1) / ROI2_CPB, B), dim3(256), 0, st, dy, box, drop, dx,
                       C, H, W, training);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
