// Variant of the table-driven ROI-refinement backward (HK_ROI_BWD=2, not yet timed on the GPU; in its own translation
// unit so that the GPU-validated kernels of apcnn.hip keep their exact code): the tap tables are built once per
// workgroup and serve ROI_BWD_CPB channel maps, and each dY map (<= 64x64 floats) is staged into LDS with coalesced
// loads before the gather, so the doubly-bounded tap loops read LDS instead of issuing dependent global loads.
// Same taps in the same order as roi_crop_bwd_tab_kernel: bit-identical dX.  (Round 1 measured the plain kernel at
// 187 us for 16 x 512 x 56 x 56 = 1.1 TB/s: per-workgroup table construction plus latency-bound gathers.)
#include "hk_roi.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

constexpr int ROI_BWD_CPB = 4;

__global__ __launch_bounds__(256) void roi_crop_bwd_tab2_kernel(const float* __restrict__ dy, const float* __restrict__ box,
                                                                const float* __restrict__ drop, float* __restrict__ dx,
                                                                int C, int H, int W, int training) {
    __shared__ CropGeom g;
    __shared__ float wy[64 * 65], wx[64 * 65];
    __shared__ __attribute__((aligned(16))) float smap[64 * 64];
    __shared__ int ylo[64], yhi[64], xlo[64], xhi[64];
    const int b = blockIdx.y, c0 = blockIdx.x * ROI_BWD_CPB, tid = threadIdx.x;
    if (tid == 0) g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    for (int e = tid; e < 64 * 65; e += 256) { wy[e] = 0.f; wx[e] = 0.f; }
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
    const int hw = H * W;
    for (int cc = 0; cc < ROI_BWD_CPB && c0 + cc < C; ++cc) {
        const float* gp = dy + ((long long)b * C + c0 + cc) * hw;
        float* dp = dx + ((long long)b * C + c0 + cc) * hw;
        __syncthreads();                                   // tables ready / previous map no longer read
        for (int p = tid; p < hw; p += 256) smap[p] = gp[p];
        __syncthreads();
        for (int p = tid; p < hw; p += 256) {
            const int iy = p / W, ix = p % W;
            const int ry = iy - g.y1, rx = ix - g.x1;
            float acc = 0.f;
            if (g.cw > 0 && g.ch > 0 && ry >= 0 && ry < g.ch && rx >= 0 && rx < g.cw) {
                const bool dropped = training && iy >= g.dy1 && iy < g.dy2 && ix >= g.dx1 && ix < g.dx2;
                if (!dropped) {
                    const int oy0 = ylo[ry], oy1 = yhi[ry], ox0 = xlo[rx], ox1 = xhi[rx];
                    for (int oy = oy0; oy <= oy1; ++oy) {
                        float rowacc = 0.f;
                        for (int ox = ox0; ox <= ox1; ++ox) rowacc += wx[rx * 65 + ox] * smap[oy * W + ox];
                        acc += wy[ry * 65 + oy] * rowacc;
                    }
                    acc *= g.rate;
                }
            }
            dp[p] = acc;
        }
    }
}

int roi_crop_bwd_v2(const float* dy, const float* box, const float* drop, float* dx, int B, int C, int H, int W,
                    int training, hipStream_t st) {
    if (H > 64 || W > 64) return HK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roi_crop_bwd_tab2_kernel, dim3((C + ROI_BWD_CPB - 1) / ROI_BWD_CPB, B), dim3(256), 0, st, dy, box,
                       drop, dx, C, H, W, training);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
