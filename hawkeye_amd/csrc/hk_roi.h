// Geometry helpers of the AP-CNN ROI refinement kernels (shared by apcnn.hip and apcnn_roi2.hip).
#pragma once
#include "hk_common.h"

namespace hk {

struct CropGeom {
    int x1, y1, cw, ch;        // integer crop (python .long() truncation + slice clipping)
    int dx1, dy1, dx2, dy2;    // integer drop rectangle (empty if dx2 <= dx1)
    float rate;                // c*h*w / sum(mask) (1 in eval)
    float sh, sw;              // source scales in / out
};

__device__ __forceinline__ int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ CropGeom crop_geom(const float* box, const float* drop, int C, int H, int W, int training) {
    CropGeom g;
    const float fx1 = box[0], fy1 = box[1], fx2 = box[2], fy2 = box[3];
    const int x1 = clipi((int)fx1, 0, W), x2 = clipi((int)fx2, 0, W);
    const int y1 = clipi((int)fy1, 0, H), y2 = clipi((int)fy2, 0, H);
    g.x1 = x1; g.y1 = y1;
    g.cw = x2 > x1 ? x2 - x1 : 0;
    g.ch = y2 > y1 ? y2 - y1 : 0;
    g.dx1 = g.dy1 = 0; g.dx2 = g.dy2 = 0;
    g.rate = 1.f;
    if (training) {
        if (drop[2] > drop[0] || drop[3] > drop[1]) {
            g.dx1 = clipi((int)drop[0], 0, W); g.dx2 = clipi((int)drop[2], 0, W);
            g.dy1 = clipi((int)drop[1], 0, H); g.dy2 = clipi((int)drop[3], 0, H);
        }
        int ox = min(g.dx2, x2) - max(g.dx1, x1); if (ox < 0) ox = 0;
        int oy = min(g.dy2, y2) - max(g.dy1, y1); if (oy < 0) oy = 0;
        const float msum = (float)C * (float)(g.cw * g.ch - ox * oy);       // torch.sum(mask_un[crop])
        g.rate = ((float)C * (fy2 - fy1)) * (fx2 - fx1) / msum;             // :509-511 (float box, not the ints)
    }
    g.sh = (float)g.ch / (float)H;
    g.sw = (float)g.cw / (float)W;
    return g;
}

// torch upsample_bilinear2d source index (align_corners = False)
__device__ __forceinline__ void src_index(float scale, int dst, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// apcnn_roi2.hip: LDS-staged maps, 8-32 channel maps per workgroup, per-pixel geometry in registers; HK_ERR_UNSUPPORTED for
// maps above 64x64
int roi_crop_fwd_v2(const float* x, const float* box, const float* drop, float* y, int B, int C, int H, int W, int training,
                    hipStream_t st);
int roi_crop_bwd_v2(const float* dy, const float* box, const float* drop, float* dx, int B, int C, int H, int W, int training,
                    hipStream_t st);

}  // namespace hk
