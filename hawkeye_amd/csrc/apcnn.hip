// AP-CNN attention-pooling head (replaces pieces of model/methods/APCNN.py):
//   K8  attention pooling (gap / a_s-weighted gap in one pass)         :256-266,:377-405,:533-538
//   K9  attention -> ROI selection with on-device greedy NMS            :444-476 + nms.py:4-93
//   K10 ROI union/drop boxes + crop / drop / rescale / bilinear resize  :478-531
// All HBM-bound streaming or tiny latency-bound kernels: wave64 shuffles for the
// reductions, 16 B per-lane loads where the layout allows, fixed reduction
// orders (bit-reproducible), no host synchronisation anywhere.
#include <cstdlib>

#include "hk_common.h"
#include "hk_roi.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// ---------------------------------------------------------------------- K8
// one wave per (b,c) row of F; 4 rows per workgroup
template <bool VEC>
__global__ __launch_bounds__(256) void att_pool_fwd_kernel(const float* __restrict__ f, const float* __restrict__ a_s,
                                                           float* __restrict__ gap, float* __restrict__ sgap,
                                                           long long rows, int C, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(row / C);
    const float* fp = f + row * HW;
    const float* ap = a_s ? a_s + (long long)b * HW : nullptr;
    float s0 = 0.f, s1 = 0.f;
    if (VEC) {
        const int n4 = HW >> 2;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = reinterpret_cast<const float4*>(fp)[i];
            s0 += (v.x + v.y) + (v.z + v.w);
            if (ap) {
                const float4 a = reinterpret_cast<const float4*>(ap)[i];
                s1 += fmaf(v.x, a.x, v.y * a.y) + fmaf(v.z, a.z, v.w * a.w);   // (explicit: the same bits in every kernel that forms it)
            }
        }
    } else {
        for (int i = lane; i < HW; i += 64) {
            const float v = fp[i];
            s0 += v;
            if (ap) s1 = fmaf(v, ap[i], s1);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
        gap[row] = s0 / (float)HW;
        if (sgap) sgap[row] = s1 / (float)HW;
    }
}

// thread owns one hw column of image b and walks the channels:
//   df[b,c,hw] = (dsgap[b,c] a_s[b,hw] + dgap[b,c]) / HW ;  da_s[b,hw] = sum_c dsgap[b,c] F[b,c,hw] / HW
__global__ __launch_bounds__(256) void att_pool_bwd_kernel(const float* __restrict__ f, const float* __restrict__ a_s,
                                                           const float* __restrict__ dgap,
                                                           const float* __restrict__ dsgap, float* __restrict__ df,
                                                           float* __restrict__ da_s, int C, int HW) {
    HK_DYN_LDS(sm);  // dgap[C], dsgap[C] of this image
    const int b = blockIdx.y;
    float* sg = sm;
    float* ss = sm + C;
    for (int c = threadIdx.x; c < C; c += 256) {
        sg[c] = dgap ? dgap[(long long)b * C + c] : 0.f;
        ss[c] = dsgap ? dsgap[(long long)b * C + c] : 0.f;
    }
    __syncthreads();
    const int hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= HW) return;
    const float inv = 1.0f / (float)HW;
    const float a = a_s ? a_s[(long long)b * HW + hw] : 0.f;
    const float* fp = f + (long long)b * C * HW + hw;
    float* dp = df + (long long)b * C * HW + hw;
    if (a_s && da_s) {
        // sixteen channels per trip, their loads issued together (with four in flight the walk over 256 channels was
        // latency bound: 22 us for 25 MB at 28 x 28); four accumulation chains combined in a fixed order
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int c = 0;
        for (; c + 16 <= C; c += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = fp[(long long)(c + u) * HW];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                acc0 = fmaf(ss[c + u], v[u], acc0);
                acc1 = fmaf(ss[c + u + 1], v[u + 1], acc1);
                acc2 = fmaf(ss[c + u + 2], v[u + 2], acc2);
                acc3 = fmaf(ss[c + u + 3], v[u + 3], acc3);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) dp[(long long)(c + u) * HW] = fmaf(ss[c + u], a, sg[c + u]) * inv;
        }
        for (; c < C; ++c) {
            acc0 = fmaf(ss[c], fp[(long long)c * HW], acc0);
            dp[(long long)c * HW] = fmaf(ss[c], a, sg[c]) * inv;
        }
        da_s[(long long)b * HW + hw] = ((acc0 + acc1) + (acc2 + acc3)) * inv;
    } else {
#pragma unroll 8
        for (int c = 0; c < C; ++c) dp[(long long)c * HW] = fmaf(ss[c], a, sg[c]) * inv;
    }
}

// The plain-GAP backward (a_s == nullptr: ChannelGate, the FPN's global branch - up to 2048 channels on a 14 x 14 map):
// df[b,c,:] = dgap[b,c] / HW.  One wave per (b, c) row like the forward; the column-walking kernel above has
// ceil(HW / 256) x B workgroups - 16 of them for a 14 x 14 map - each looping over all C channels (55 us inside the
// AP-CNN step, profiles/r3_step_APCNN_kernel_stats.csv).
template <bool VEC>
__global__ __launch_bounds__(256) void att_pool_bwd_gap_kernel(const float* __restrict__ dgap, float* __restrict__ df,
                                                               long long rows, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float v = dgap[row] * (1.0f / (float)HW);           // (the column kernel: (0 * a + dgap) * inv - the same bits)
    float* dp = df + row * HW;
    if (VEC) {
        const float4 q = make_float4(v, v, v, v);
        for (int i = lane; i < (HW >> 2); i += 64) reinterpret_cast<float4*>(dp)[i] = q;
    } else {
        for (int i = lane; i < HW; i += 64) dp[i] = v;
    }
}

// The three pyramid levels of AP-CNN (56 x 56, 28 x 28, 14 x 14 at 448 x 448 inputs) in ONE launch per direction.  The
// per-level launches above are launch-bound at the two small levels (B = 16: 9.7 + 9.8 + 11.8 us forward for 68 MB = 8.5 us
// of traffic, 17 + 12.5 + 10 us backward); the levels are independent (APCNN.py:256-266: only the channel gates chain),
// so their rows / column blocks are dealt out of one grid, the largest level first.  Same per-row / per-column
// arithmetic as the kernels above: bit-identical results.
struct AttLevels {
    const float* f[3];
    const float* a[3];
    float* gap[3];          // forward: outputs; backward: dgap (read)
    float* sgap[3];         //                             dsgap (read)
    float* df[3];
    float* da[3];
    int hw[3];
    int blk0[4];            // backward: first blockIdx.x of each level (blk0[3] = grid.x)
};

__global__ __launch_bounds__(256) void att_pool3_fwd_kernel(AttLevels L, long long rows, int C) {
    const long long r3 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r3 >= 3 * rows) return;
    const int lvl = (int)(r3 / rows);                          // wave-uniform
    const long long row = r3 - lvl * rows;
    const int lane = threadIdx.x & 63, HW = L.hw[lvl];
    const int b = (int)(row / C);
    const float* fp = L.f[lvl] + row * HW;
    const float* ap = L.a[lvl] + (long long)b * HW;
    float s0 = 0.f, s1 = 0.f;
    const int n4 = HW >> 2;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = reinterpret_cast<const float4*>(fp)[i];
        const float4 a = reinterpret_cast<const float4*>(ap)[i];
        s0 += (v.x + v.y) + (v.z + v.w);
        s1 += fmaf(v.x, a.x, v.y * a.y) + fmaf(v.z, a.z, v.w * a.w);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
        L.gap[lvl][row] = s0 / (float)HW;
        L.sgap[lvl][row] = s1 / (float)HW;
    }
}

__global__ __launch_bounds__(256) void att_pool3_bwd_kernel(AttLevels L, int C) {
    HK_DYN_LDS(sm);  // dgap[C], dsgap[C] of this image and level
    const int lvl = (int)blockIdx.x >= L.blk0[2] ? 2 : ((int)blockIdx.x >= L.blk0[1] ? 1 : 0);   // block-uniform
    const int b = blockIdx.y, HW = L.hw[lvl];
    float* sg = sm;
    float* ss = sm + C;
    for (int c = threadIdx.x; c < C; c += 256) {
        sg[c] = L.gap[lvl][(long long)b * C + c];
        ss[c] = L.sgap[lvl][(long long)b * C + c];
    }
    __syncthreads();
    const int hw = ((int)blockIdx.x - L.blk0[lvl]) * 256 + threadIdx.x;
    if (hw >= HW) return;
    const float inv = 1.0f / (float)HW;
    const float a = L.a[lvl][(long long)b * HW + hw];
    const float* fp = L.f[lvl] + (long long)b * C * HW + hw;
    float* dp = L.df[lvl] + (long long)b * C * HW + hw;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int c = 0;
    for (; c + 16 <= C; c += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = fp[(long long)(c + u) * HW];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            acc0 = fmaf(ss[c + u], v[u], acc0);
            acc1 = fmaf(ss[c + u + 1], v[u + 1], acc1);
            acc2 = fmaf(ss[c + u + 2], v[u + 2], acc2);
            acc3 = fmaf(ss[c + u + 3], v[u + 3], acc3);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) dp[(long long)(c + u) * HW] = fmaf(ss[c + u], a, sg[c + u]) * inv;
    }
    for (; c < C; ++c) {
        acc0 = fmaf(ss[c], fp[(long long)c * HW], acc0);
        dp[(long long)c * HW] = fmaf(ss[c], a, sg[c]) * inv;
    }
    L.da[lvl][(long long)b * HW + hw] = ((acc0 + acc1) + (acc2 + acc3)) * inv;
}

// ---------------------------------------------------------------------- K9
struct Cand {
    float s;
    int i;
};
__device__ __forceinline__ Cand better(Cand a, Cand b) {  // higher score; ties: higher index
    return (b.s > a.s || (b.s == a.s && b.i > a.i)) ? b : a;
}

// one 256-thread workgroup per image; scores + alive flags in LDS (sm: 2 h w floats)
__device__ __forceinline__ void att_roi_select_body(const float* __restrict__ att, float* __restrict__ rois,
                                                    int* __restrict__ count, int h, int w, int stride, float anchor,
                                                    int img_h, int img_w, int r0, int r1, int c0, int c1, float thr,
                                                    int topk, float* sm) {
    __shared__ float red[4];
    __shared__ Cand wbest[4];
    __shared__ Cand winner;
    const int b = blockIdx.x, n = h * w, tid = threadIdx.x;
    float* score = sm;
    float* alive = sm + n;
    const float* ap = att + (long long)b * n;
    float part = 0.f;
    for (int p = tid; p < n; p += 256) {
        const int y = p / w, x = p % w;
        const float keep = (y >= r0 && y < r1 && x >= c0 && x < c1) ? 1.f : 0.f;
        const float s = ap[p] * keep;   // att_mask * att_corner_unmask   (APCNN.py:455)
        score[p] = s;
        part += s;
    }
    const float mean = block_sum<4>(part, red) / (float)n;      // scores.mean() over ALL cells (:460)
    for (int p = tid; p < n; p += 256) alive[p] = (score[p] > mean) ? 1.f : 0.f;   // :461
    __syncthreads();

    const float half = 0.5f * anchor;
    const float area = (2.f * half) * (2.f * half);              // (x2-x1)*(y2-y1), identical for all anchors
    float* out = rois + (long long)b * topk * 5;
    int found = 0;
    for (int t = 0; t < topk; ++t) {
        Cand best;
        best.s = -1.f; best.i = -1;
        for (int p = tid; p < n; p += 256)
            if (alive[p] > 0.f) {
                Cand c; c.s = score[p]; c.i = p;
                best = better(best, c);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Cand oth;
            oth.s = __shfl_xor(best.s, o, 64);
            oth.i = __shfl_xor(best.i, o, 64);
            best = better(best, oth);
        }
        if ((tid & 63) == 0) wbest[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            Cand r = wbest[0];
            for (int k = 1; k < 4; ++k) r = better(r, wbest[k]);
            winner = r;
        }
        __syncthreads();
        const Cand wv = winner;
        if (wv.i < 0) break;   // uniform: nothing left (reference loop ends, nms.py:38)
        const float wx = (float)((wv.i % w) * stride), wy = (float)((wv.i / w) * stride);
        const float wx1 = wx - half, wy1 = wy - half, wx2 = wx + half, wy2 = wy + half;
        if (tid == 0) {
            out[t * 5 + 0] = fmaxf(wx1, 0.f);                     // clamp after NMS (:469-472)
            out[t * 5 + 1] = fmaxf(wy1, 0.f);
            out[t * 5 + 2] = fminf(wx2, (float)(img_w - 1));
            out[t * 5 + 3] = fminf(wy2, (float)(img_h - 1));
            out[t * 5 + 4] = wv.s;
        }
        ++found;
        for (int p = tid; p < n; p += 256) {                      // suppress IoU >= thr  (nms.py:56-91)
            if (alive[p] <= 0.f) continue;
            if (p == wv.i) { alive[p] = 0.f; continue; }
            const float px = (float)((p % w) * stride), py = (float)((p / w) * stride);
            const float xx1 = fmaxf(px - half, wx1), yy1 = fmaxf(py - half, wy1);
            const float xx2 = fminf(px + half, wx2), yy2 = fminf(py + half, wy2);
            const float iw = fmaxf(xx2 - xx1, 0.f), ih = fmaxf(yy2 - yy1, 0.f);
            const float inter = iw * ih;
            const float uni = (area - inter) + area;
            if (!(inter / uni < thr)) alive[p] = 0.f;
        }
        __syncthreads();
    }
    for (int e = found * 5 + tid; e < topk * 5; e += 256) out[e] = 0.f;
    if (tid == 0) count[b] = found;
}

__global__ __launch_bounds__(256) void att_roi_select_kernel(const float* __restrict__ att, float* __restrict__ rois,
                                                             int* __restrict__ count, int h, int w, int stride,
                                                             float anchor, int img_h, int img_w, int r0, int r1, int c0,
                                                             int c1, float thr, int topk) {
    HK_DYN_LDS(sm);  // score[h*w] ; alive flags packed as floats (>0 alive)
    att_roi_select_body(att, rois, count, h, w, stride, anchor, img_h, img_w, r0, r1, c0, c1, thr, topk, sm);
}

// The pyramid levels of one forward (APCNN.py:256-266 calls get_att_roi once per level, each a chain of k dependent
// arg-max rounds on one workgroup per image): blockIdx.y = level, so the three chains run side by side - 41 us for three
// launches -> the longest level.
struct RoiLevel {
    const float* att;
    float* rois;
    int* count;
    int h, w, stride, r0, r1, c0, c1, topk;
    float anchor;
};
struct RoiLevels {
    RoiLevel l[3];
};
__global__ __launch_bounds__(256) void att_roi_select3_kernel(const RoiLevels L, int img_h, int img_w, float thr) {
    HK_DYN_LDS(sm);
    const int lv = blockIdx.y;
    const RoiLevel& q = lv == 0 ? L.l[0] : (lv == 1 ? L.l[1] : L.l[2]);
    att_roi_select_body(q.att, q.rois, q.count, q.h, q.w, q.stride, q.anchor, img_h, img_w, q.r0, q.r1, q.c0, q.c1, thr,
                        q.topk, sm);
}

// ---------------------------------------------------------------------- K10 boxes
__global__ void roi_boxes_kernel(const float* __restrict__ r3, const int* __restrict__ n3, int k3,
                                 const float* __restrict__ r4, const int* __restrict__ n4, int k4,
                                 const float* __restrict__ r5, const int* __restrict__ n5, int k5,
                                 const float* __restrict__ u01, float scale, float* __restrict__ box,
                                 float* __restrict__ drop, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float x1 = 3.0e38f, y1 = 3.0e38f, x2 = -3.0e38f, y2 = -3.0e38f;
    const float* tabs[3] = {r3 + (long long)b * k3 * 5, r4 + (long long)b * k4 * 5, r5 + (long long)b * k5 * 5};
    const int cnts[3] = {n3[b], n4[b], n5[b]};
    for (int l = 0; l < 3; ++l)
        for (int i = 0; i < cnts[l]; ++i) {
            const float* r = tabs[l] + i * 5;
            x1 = fminf(x1, r[0] / scale); y1 = fminf(y1, r[1] / scale);    // :487-489
            x2 = fmaxf(x2, r[2] / scale); y2 = fmaxf(y2, r[3] / scale);
        }
    box[b * 4 + 0] = x1; box[b * 4 + 1] = y1; box[b * 4 + 2] = x2; box[b * 4 + 3] = y2;
    float d[4] = {0.f, 0.f, -1.f, -1.f};
    if (u01) {
        const float pr = u01[b * 2], ui = u01[b * 2 + 1];
        const int lvl = pr < 0.3f ? 0 : (pr < 0.6f ? 1 : -1);               // :494-504
        if (lvl >= 0 && cnts[lvl] > 0) {
            int idx = (int)(ui * (float)cnts[lvl]);
            if (idx > cnts[lvl] - 1) idx = cnts[lvl] - 1;
            const float* r = tabs[lvl] + idx * 5;
            d[0] = r[0] / scale; d[1] = r[1] / scale; d[2] = r[2] / scale; d[3] = r[3] / scale;
        }
    }
    drop[b * 4 + 0] = d[0]; drop[b * 4 + 1] = d[1]; drop[b * 4 + 2] = d[2]; drop[b * 4 + 3] = d[3];
}

// ---------------------------------------------------------------------- K10 crop/resize

__global__ __launch_bounds__(256) void roi_crop_fwd_kernel(const float* __restrict__ x, const float* __restrict__ box,
                                                           const float* __restrict__ drop, float* __restrict__ y,
                                                           int C, int H, int W, int training) {
    const int b = blockIdx.z, c = blockIdx.y;
    const CropGeom g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    const float* xp = x + ((long long)b * C + c) * H * W;
    float* yp = y + ((long long)b * C + c) * H * W;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < H * W; o += gridDim.x * 256) {
        if (g.cw <= 0 || g.ch <= 0) { yp[o] = 0.f; continue; }
        const int oy = o / W, ox = o % W;
        int a0, a1, b0, b1;
        float la0, la1, lb0, lb1;
        src_index(g.sh, oy, g.ch, a0, a1, la0, la1);
        src_index(g.sw, ox, g.cw, b0, b1, lb0, lb1);
        float v[2][2];
        const int ys[2] = {g.y1 + a0, g.y1 + a1}, xs[2] = {g.x1 + b0, g.x1 + b1};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t = xp[ys[i] * W + xs[j]];
                if (training) {
                    const bool dropped = ys[i] >= g.dy1 && ys[i] < g.dy2 && xs[j] >= g.dx1 && xs[j] < g.dx2;
                    t = (dropped ? 0.f : t) * g.rate;
                }
                v[i][j] = t;
            }
        const float r0 = fmaf(lb1, v[0][1], lb0 * v[0][0]), r1 = fmaf(lb1, v[1][1], lb0 * v[1][0]);
        yp[o] = fmaf(la1, r1, la0 * r0);
    }
}

// gather form of the transpose: every input pixel sums the output pixels that sampled it
__global__ __launch_bounds__(256) void roi_crop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ box,
                                                           const float* __restrict__ drop, float* __restrict__ dx,
                                                           int C, int H, int W, int training) {
    const int b = blockIdx.z, c = blockIdx.y;
    const CropGeom g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    const float* gp = dy + ((long long)b * C + c) * H * W;
    float* dp = dx + ((long long)b * C + c) * H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int iy = p / W, ix = p % W;
        const int ry = iy - g.y1, rx = ix - g.x1;
        float acc = 0.f;
        if (g.cw > 0 && g.ch > 0 && ry >= 0 && ry < g.ch && rx >= 0 && rx < g.cw) {
            const bool dropped = training && iy >= g.dy1 && iy < g.dy2 && ix >= g.dx1 && ix < g.dx2;
            if (!dropped) {
                // output rows/cols whose source interval touches ry / rx (conservative bounds, exact weights)
                int oy0 = (int)floorf(((float)ry - 0.5f) / g.sh - 0.5f) - 1, oy1 = (int)ceilf(((float)ry + 1.5f) / g.sh - 0.5f) + 1;
                int ox0 = (int)floorf(((float)rx - 0.5f) / g.sw - 0.5f) - 1, ox1 = (int)ceilf(((float)rx + 1.5f) / g.sw - 0.5f) + 1;
                oy0 = clipi(oy0, 0, H - 1); oy1 = clipi(oy1, 0, H - 1);
                ox0 = clipi(ox0, 0, W - 1); ox1 = clipi(ox1, 0, W - 1);
                for (int oy = oy0; oy <= oy1; ++oy) {
                    int a0, a1; float la0, la1;
                    src_index(g.sh, oy, g.ch, a0, a1, la0, la1);
                    const float wy = (a0 == ry ? la0 : 0.f) + (a1 == ry ? la1 : 0.f);
                    if (wy == 0.f) continue;
                    float rowacc = 0.f;
                    for (int ox = ox0; ox <= ox1; ++ox) {
                        int b0, b1; float lb0, lb1;
                        src_index(g.sw, ox, g.cw, b0, b1, lb0, lb1);
                        const float wx = (b0 == rx ? lb0 : 0.f) + (b1 == rx ? lb1 : 0.f);
                        rowacc = fmaf(wx, gp[oy * W + ox], rowacc);
                    }
                    acc = fmaf(wy, rowacc, acc);
                }
                acc *= g.rate;
            }
        }
        dp[p] = acc;
    }
}


// ---------------------------------------------------------------------- K10 table-driven variants (H, W <= 64)
// One workgroup per (image, channel) map.  The crop geometry and the per-row / per-column bilinear tables depend only
// on the image, so they are built once per workgroup in LDS (the kernels above recompute them per thread: ~100 VALU
// instructions per output element, measured 118 us forward / 338 us backward at B=16, C=512, 56x56).
constexpr int ROI_CPB = 1;   // channel maps per workgroup; 8 was measured slower (fwd 110 vs 73 us: fewer workgroups), bwd unchanged

struct AxisTab {         // source taps of one output coordinate
    int i0, i1;
    float l0, l1;
};

__global__ __launch_bounds__(256) void roi_crop_fwd_tab_kernel(const float* __restrict__ x, const float* __restrict__ box,
                                                               const float* __restrict__ drop, float* __restrict__ y,
                                                               int C, int H, int W, int training) {
    __shared__ CropGeom g;
    __shared__ AxisTab ty[64], tx[64];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) g = crop_geom(box + b * 4, drop + b * 4, C, H, W, training);
    __syncthreads();
    if (tid < H && g.ch > 0) src_index(g.sh, tid, g.ch, ty[tid].i0, ty[tid].i1, ty[tid].l0, ty[tid].l1);
    if (tid >= 64 && tid - 64 < W && g.cw > 0)
        src_index(g.sw, tid - 64, g.cw, tx[tid - 64].i0, tx[tid - 64].i1, tx[tid - 64].l0, tx[tid - 64].l1);
    __syncthreads();
    const bool empty = g.cw <= 0 || g.ch <= 0;
    for (int cc = 0; cc < ROI_CPB && c * ROI_CPB + cc < C; ++cc) {     // the tables serve ROI_CPB channel maps
    const float* xp = x + ((long long)b * C + c * ROI_CPB + cc) * H * W;
    float* yp = y + ((long long)b * C + c * ROI_CPB + cc) * H * W;
    for (int o = tid; o < H * W; o += 256) {
        if (empty) { yp[o] = 0.f; continue; }
        const int oy = o / W, ox = o % W;
        const AxisTab a = ty[oy], q = tx[ox];
        const int ys[2] = {g.y1 + a.i0, g.y1 + a.i1}, xs[2] = {g.x1 + q.i0, g.x1 + q.i1};
        float v[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t = xp[ys[i] * W + xs[j]];
                if (training) {
                    const bool dropped = ys[i] >= g.dy1 && ys[i] < g.dy2 && xs[j] >= g.dx1 && xs[j] < g.dx2;
                    t = (dropped ? 0.f : t) * g.rate;
                }
                v[i][j] = t;
            }
        // (explicit fmaf: the same bits as roi_crop_fwd_tab2_kernel, whatever the compiler would contract)
        const float r0 = fmaf(q.l1, v[0][1], q.l0 * v[0][0]), r1 = fmaf(q.l1, v[1][1], q.l0 * v[1][0]);
        yp[o] = fmaf(a.l1, r1, a.l0 * r0);
    }
    }
}

// dX = rate * mask * (Wy^T dY Wx) restricted to the crop.  Wy [ch][H] / Wx [cw][W] are built dense in LDS (each
// output coordinate owns its column: two adds, no race) together with the non-zero range of every row, so the gather
// of an input pixel is a short doubly-bounded loop over table entries.  Deterministic, no atomics.
__global__ __launch_bounds__(256) void roi_crop_bwd_tab_kernel(const float* __restrict__ dy, const float* __restrict__ box,
                                                               const float* __restrict__ drop, float* __restrict__ dx,
                                                               int C, int H, int W, int training) {
    __shared__ CropGeom g;
    __shared__ float wy[64 * 65], wx[64 * 65];
    __shared__ int ylo[64], yhi[64], xlo[64], xhi[64];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
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
    __syncthreads();
    for (int cc = 0; cc < ROI_CPB && c * ROI_CPB + cc < C; ++cc) {     // the tables serve ROI_CPB channel maps
    const float* gp = dy + ((long long)b * C + c * ROI_CPB + cc) * H * W;
    float* dp = dx + ((long long)b * C + c * ROI_CPB + cc) * H * W;
    for (int p = tid; p < H * W; p += 256) {
        const int iy = p / W, ix = p % W;
        const int ry = iy - g.y1, rx = ix - g.x1;
        float acc = 0.f;
        if (g.cw > 0 && g.ch > 0 && ry >= 0 && ry < g.ch && rx >= 0 && rx < g.cw) {
            const bool dropped = training && iy >= g.dy1 && iy < g.dy2 && ix >= g.dx1 && ix < g.dx2;
            if (!dropped) {
                const int oy0 = ylo[ry], oy1 = yhi[ry], ox0 = xlo[rx], ox1 = xhi[rx];
                for (int oy = oy0; oy <= oy1; ++oy) {
                    float rowacc = 0.f;
                    // explicit fmaf: the backward kernels must agree bit for bit, whatever the compiler would contract
                    for (int ox = ox0; ox <= ox1; ++ox) rowacc = fmaf(wx[rx * 65 + ox], gp[oy * W + ox], rowacc);
                    acc = fmaf(wy[ry * 65 + oy], rowacc, acc);
                }
                acc *= g.rate;
            }
        }
        dp[p] = acc;
    }
    }
}

}  // namespace hk

using namespace hk;

extern "C" int hk_att_pool_fwd(const float* f, const float* a_s, float* gap, float* sgap, int B, int C, int HW,
                               hk_stream_t stream) {
    if (!f || !gap || B <= 0 || C <= 0 || HW <= 0 || (a_s && !sgap)) return HK_ERR_BAD_ARG;
    const long long rows = (long long)B * C;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bool vec = (HW % 4 == 0) && aligned16(f) && (!a_s || aligned16(a_s));
    if (vec)
        hipLaunchKernelGGL(att_pool_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, f, a_s, gap, sgap, rows, C, HW);
    else
        hipLaunchKernelGGL(att_pool_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, f, a_s, gap, sgap, rows, C, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_att_pool3_fwd(const float* f0, const float* f1, const float* f2, const float* a0, const float* a1,
                                const float* a2, float* gap, float* sgap, int B, int C, int HW0, int HW1, int HW2,
                                hk_stream_t stream) {
    if (!f0 || !f1 || !f2 || !a0 || !a1 || !a2 || !gap || !sgap || B <= 0 || C <= 0 || HW0 <= 0 || HW1 <= 0 || HW2 <= 0)
        return HK_ERR_BAD_ARG;
    const long long rows = (long long)B * C;
    const float* fs[3] = {f0, f1, f2};
    const float* as[3] = {a0, a1, a2};
    const int hws[3] = {HW0, HW1, HW2};
    bool vec = true;
    for (int l = 0; l < 3; ++l) vec = vec && hws[l] % 4 == 0 && aligned16(fs[l]) && aligned16(as[l]);
    if (!vec) {                                   // ragged maps: the per-level kernels
        for (int l = 0; l < 3; ++l) {
            const int rc = hk_att_pool_fwd(fs[l], as[l], gap + l * rows, sgap + l * rows, B, C, hws[l], stream);
            if (rc != HK_OK) return rc;
        }
        return HK_OK;
    }
    AttLevels L = {};
    for (int l = 0; l < 3; ++l) {
        L.f[l] = fs[l]; L.a[l] = as[l]; L.hw[l] = hws[l];
        L.gap[l] = gap + l * rows; L.sgap[l] = sgap + l * rows;
    }
    hipLaunchKernelGGL(att_pool3_fwd_kernel, dim3((unsigned)((3 * rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, L, rows, C);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_att_pool3_bwd(const float* f0, const float* f1, const float* f2, const float* a0, const float* a1,
                                const float* a2, const float* dgap, const float* dsgap, float* df0, float* df1, float* df2,
                                float* da0, float* da1, float* da2, int B, int C, int HW0, int HW1, int HW2,
                                hk_stream_t stream) {
    if (!f0 || !f1 || !f2 || !a0 || !a1 || !a2 || !dgap || !dsgap || !df0 || !df1 || !df2 || !da0 || !da1 || !da2 || B <= 0 ||
        C <= 0 || HW0 <= 0 || HW1 <= 0 || HW2 <= 0)
        return HK_ERR_BAD_ARG;
    if ((size_t)2 * C * sizeof(float) > 64 * 1024) return HK_ERR_UNSUPPORTED;
    const long long rows = (long long)B * C;
    AttLevels L = {};
    const float* fs[3] = {f0, f1, f2};
    const float* as[3] = {a0, a1, a2};
    float* dfs[3] = {df0, df1, df2};
    float* das[3] = {da0, da1, da2};
    const int hws[3] = {HW0, HW1, HW2};
    int nb = 0;
    for (int l = 0; l < 3; ++l) {
        L.f[l] = fs[l]; L.a[l] = as[l]; L.df[l] = dfs[l]; L.da[l] = das[l]; L.hw[l] = hws[l];
        L.gap[l] = const_cast<float*>(dgap) + l * rows;
        L.sgap[l] = const_cast<float*>(dsgap) + l * rows;
        L.blk0[l] = nb;
        nb += (hws[l] + 255) / 256;
    }
    L.blk0[3] = nb;
    hipLaunchKernelGGL(att_pool3_bwd_kernel, dim3(nb, B), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, L, C);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_att_pool_bwd(const float* f, const float* a_s, const float* dgap, const float* dsgap, float* df,
                               float* da_s, int B, int C, int HW, hk_stream_t stream) {
    if (!df || B <= 0 || C <= 0 || HW <= 0 || (!dgap && !dsgap)) return HK_ERR_BAD_ARG;
    // without a spatial gate there is no sgap: only dgap can arrive, and F itself is not needed (f may be NULL)
    if (!a_s && (!dgap || dsgap)) return HK_ERR_BAD_ARG;
    if (a_s && !f) return HK_ERR_BAD_ARG;
    if (!a_s) {                                  // plain GAP: row-parallel broadcast
        const long long rows = (long long)B * C;
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (HW % 4 == 0 && aligned16(df))
            hipLaunchKernelGGL(att_pool_bwd_gap_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dgap, df, rows, HW);
        else
            hipLaunchKernelGGL(att_pool_bwd_gap_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dgap, df, rows, HW);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    if ((size_t)2 * C * sizeof(float) > 64 * 1024) return HK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(att_pool_bwd_kernel, dim3((HW + 255) / 256, B), dim3(256), 2 * C * sizeof(float),
                       (hipStream_t)stream, f, a_s, dgap, dsgap, df, da_s, C, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_att_roi_select(const float* att, float* rois, int32_t* count, int B, int h, int w, int feature_stride,
                                 float anchor_size, int img_h, int img_w, int keep_r0, int keep_r1, int keep_c0,
                                 int keep_c1, float iou_thr, int topk, hk_stream_t stream) {
    if (!att || !rois || !count || B <= 0 || h <= 0 || w <= 0 || topk <= 0) return HK_ERR_BAD_ARG;
    const size_t sm = (size_t)2 * h * w * sizeof(float);
    if (sm > 150 * 1024) return HK_ERR_UNSUPPORTED;
    HK_ALLOW_BIG_LDS(&att_roi_select_kernel, sm);                 // maps above ~90 x 90: more than the default 64 KB
    hipLaunchKernelGGL(att_roi_select_kernel, dim3(B), dim3(256), sm, (hipStream_t)stream, att, rois, (int*)count, h, w,
                       feature_stride, anchor_size, img_h, img_w, keep_r0, keep_r1, keep_c0, keep_c1, iou_thr, topk);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_att_roi_select3(const float* const* att, float* const* rois, int32_t* const* count, int B, const int* h,
                                  const int* w, const int* feature_stride, const float* anchor_size, int img_h, int img_w,
                                  const int* keep, float iou_thr, const int* topk, hk_stream_t stream) {
    if (!att || !rois || !count || !h || !w || !feature_stride || !anchor_size || !keep || !topk || B <= 0)
        return HK_ERR_BAD_ARG;
    RoiLevels L;
    size_t sm = 0;
    for (int i = 0; i < 3; ++i) {
        if (!att[i] || !rois[i] || !count[i] || h[i] <= 0 || w[i] <= 0 || topk[i] <= 0) return HK_ERR_BAD_ARG;
        L.l[i].att = att[i]; L.l[i].rois = rois[i]; L.l[i].count = (int*)count[i];
        L.l[i].h = h[i]; L.l[i].w = w[i]; L.l[i].stride = feature_stride[i]; L.l[i].anchor = anchor_size[i];
        L.l[i].r0 = keep[4 * i]; L.l[i].r1 = keep[4 * i + 1]; L.l[i].c0 = keep[4 * i + 2]; L.l[i].c1 = keep[4 * i + 3];
        L.l[i].topk = topk[i];
        const size_t need = (size_t)2 * h[i] * w[i] * sizeof(float);
        sm = need > sm ? need : sm;
    }
    if (sm > 150 * 1024) return HK_ERR_UNSUPPORTED;
    HK_ALLOW_BIG_LDS(&att_roi_select3_kernel, sm);
    hipLaunchKernelGGL(att_roi_select3_kernel, dim3(B, 3), dim3(256), sm, (hipStream_t)stream, L, img_h, img_w, iou_thr);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_roi_boxes(const float* rois3, const int32_t* cnt3, int k3, const float* rois4, const int32_t* cnt4,
                            int k4, const float* rois5, const int32_t* cnt5, int k5, const float* u01, float scale,
                            float* box, float* drop, int B, hk_stream_t stream) {
    if (!rois3 || !cnt3 || !rois4 || !cnt4 || !rois5 || !cnt5 || !box || !drop || B <= 0 || scale <= 0.f)
        return HK_ERR_BAD_ARG;
    hipLaunchKernelGGL(roi_boxes_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, rois3, (const int*)cnt3,
                       k3, rois4, (const int*)cnt4, k4, rois5, (const int*)cnt5, k5, u01, scale, box, drop, B);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_roi_crop_resize_fwd(const float* x, const float* box, const float* drop, float* y, int B, int C, int H,
                                      int W, int training, hk_stream_t stream) {
    if (!x || !box || !drop || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HK_ERR_BAD_ARG;
    if (tuning().roi_bwd != 1) {                            // default: apcnn_roi2.hip; roi_bwd = 1 keeps the round-1 kernels (both directions)
        const int rc = roi_crop_fwd_v2(x, box, drop, y, B, C, H, W, training, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    if (H <= 64 && W <= 64) {
        hipLaunchKernelGGL(roi_crop_fwd_tab_kernel, dim3((C + ROI_CPB - 1) / ROI_CPB, B), dim3(256), 0, (hipStream_t)stream, x, box, drop, y, C, H, W,
                           training);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    int gx = (H * W + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(roi_crop_fwd_kernel, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream, x, box, drop, y, C, H, W,
                       training);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_roi_crop_resize_bwd(const float* dy, const float* box, const float* drop, float* dx, int B, int C,
                                      int H, int W, int training, hk_stream_t stream) {
    if (!dy || !box || !drop || !dx || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HK_ERR_BAD_ARG;
    if (tuning().roi_bwd != 1) {                            // default: uniform-window kernel (apcnn_roi2.hip); 1 = the round-1 kernel below
        const int rc = roi_crop_bwd_v2(dy, box, drop, dx, B, C, H, W, training, (hipStream_t)stream);
        if (rc != HK_ERR_UNSUPPORTED) return rc;
    }
    if (H <= 64 && W <= 64) {
        hipLaunchKernelGGL(roi_crop_bwd_tab_kernel, dim3((C + ROI_CPB - 1) / ROI_CPB, B), dim3(256), 0, (hipStream_t)stream, dy, box, drop, dx, C, H, W,
                           training);
        HK_LAUNCH_CHECK();
        return HK_OK;
    }
    int gx = (H * W + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(roi_crop_bwd_kernel, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream, dy, box, drop, dx, C, H, W,
                       training);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
