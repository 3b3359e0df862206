// Epilogues of the VGG trunk's convolutions (SURVEY 8a row A1': the BCNN / CBCNN module is `features` = 13 x [Conv2d 3x3 + bias,
// ReLU] with five MaxPool2d(2, 2), model/backbone/vgg.py:24-57, and at 448 x 448 / batch 64 its activations are up to 3.3 GB
// each).  The convolutions themselves stay MIOpen's; what PyTorch-ROCm runs AROUND them is one full-tensor pass per
// elementwise op - bias add, ReLU, max-pool forward; pool backward, ReLU backward, bias-gradient reduction - 26.9 ms of the
// 213.9 ms BCNN training step (profiles/r5_step_BCNN_kernel_stats.csv: 12.5 %, ninety times the whole pooling head).
// These kernels make it ONE pass per convolution and direction, on the channels_last tensors the trunk runs in:
//
//   bias_relu_fwd        y = max(x + b, 0) in place on the convolution's output                  (read 1, write 1; was 2 + 2)
//   bias_relu_bwd        dx = dy where y > 0, db = sum dx                                        (read 2 - or 1 1/16 with the forward's
//                        sign mask -, write 1; was 3 + 1 + 1)
//   bias_relu_pool_fwd   p = maxpool2x2(max(x + b, 0)) + a 2-bit argmax per element              (read 1, write 1/4 + 1/64;
//                        the full-resolution activation is never written: nobody needs it - the next convolution reads p,
//                        this convolution's weight gradient reads its INPUT - was 2 + 2 + 1.75)
//   bias_relu_pool_bwd   dx = dp at the argmax where p > 0, else 0; db = sum dx                  (read 1/2 + 1/64, write 1; was 5.75)
//
// All HBM-bound streams (16-byte accesses, NHWC: a pixel's channels are contiguous).  Same arithmetic as the ops they
// replace: x + b then max(., 0) (torch.relu's clamp_min, NaN kept); the pooling window scanned row-major with "strictly
// greater or NaN replaces" (ATen's max_pool2d: the FIRST maximum wins); the backward routes dp to that element and
// applies ReLU's mask (y > 0  <=>  p > 0 at the argmax).  db is summed in a fixed order (per thread over its rows, threads of
// a column quad in order, workgroups in order): deterministic, no atomics.
#include "hk_common.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

constexpr int TRUNK_PART_BLOCKS = 1024;        // workgroups (= partial db rows) of the backward kernels, at most

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int t = 0; t < 4; ++t) r[t] = v[t] < 0.f ? 0.f : v[t];            // (NaN < 0 is false: NaN stays, as clamp_min)
    return r;
}

// x [M][C] (M = N H W pixels), C % 4 == 0: in place.  n4 = M C / 4 float4, c4 = C / 4.
// MASK: also write one byte per float4 - bit t set where channel t of the quad came out positive - so that the backward reads
// 1/16 of a map instead of the map (the activation itself stays: it is the next convolution's input)
template <bool MASK>
__global__ __launch_bounds__(256) void bias_relu_fwd_kernel(float* __restrict__ x, const float* __restrict__ b,
                                                            uint8_t* __restrict__ mask, long long n4, int c4) {
    f32x4* x4 = reinterpret_cast<f32x4*>(x);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n4) v[u] = x4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = i + u * stride;
            if (j < n4) {
                const f32x4 r = relu4(v[u] + b4[(int)(j % c4)]);
                x4[j] = r;
                if (MASK) mask[j] = (uint8_t)((r[0] > 0.f ? 1 : 0) | (r[1] > 0.f ? 2 : 0) | (r[2] > 0.f ? 4 : 0) | (r[3] > 0.f ? 8 : 0));
            }
        }
    }
}

// Shared tail of the two backward kernels: this thread's partial db of its column quad -> the workgroup's partial row.
// Thread t owns column quad t % c4 (256 % c4 == 0: 256 / c4 threads per quad); their sums are added in thread order.
__device__ __forceinline__ void trunk_db_partial(f32x4 acc, float* __restrict__ part, int c4) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x;
    red[tid] = acc;
    __syncthreads();
    if (tid < c4) {
        f32x4 s = red[tid];
        for (int g = 1; g < 256 / c4; ++g) s += red[tid + g * c4];
        reinterpret_cast<f32x4*>(part + (long long)blockIdx.x * 4 * c4)[tid] = s;
    }
}

// dx = dy * (y > 0) ; part[block][C] = this workgroup's column sums of dx.  rows = M, workgroup b owns rows
// [b rpb, (b + 1) rpb).  dx may alias dy.
template <bool MASK>
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const float* dy, const float* __restrict__ y,
                                                            const uint8_t* __restrict__ mask, float* dx,
                                                            float* __restrict__ part, long long rows, long long rpb, int c4) {
    const int tid = threadIdx.x, q = tid % c4, r0 = tid / c4, rstep = 256 / c4;
    const long long rbeg = (long long)blockIdx.x * rpb, rend = rbeg + rpb < rows ? rbeg + rpb : rows;
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long long r = rbeg + r0; r < rend; r += 4 * rstep) {
        f32x4 g[4], v[4];
        unsigned mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + u * rstep;
            if (rr < rend) {
                g[u] = dy4[rr * c4 + q];
                if (MASK) mk[u] = mask[rr * c4 + q];
                else v[u] = y4[rr * c4 + q];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + u * rstep;
            if (rr < rend) {
                f32x4 d;
#pragma unroll
                for (int t = 0; t < 4; ++t)                                            // threshold_backward: y <= 0 -> 0
                    d[t] = (MASK ? ((mk[u] >> t) & 1u) != 0u : v[u][t] > 0.f) ? g[u][t] : 0.f;
                dx4[rr * c4 + q] = d;
                acc += d;
            }
        }
    }
    trunk_db_partial(acc, part, c4);
}

// db[c] = sum over the nblk partial rows, in order: one workgroup per 16 channels, 16 groups of partial rows, then the groups
__global__ __launch_bounds__(256) void trunk_db_final_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ db) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C) {
        int k = g;
        for (; k + 7 * 16 < nblk; k += 8 * 16) {        // eight independent loads in flight, added in order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + 16 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nblk; k += 16) s += part[(long long)k * C + c];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = red[0][cl];
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        db[c] = t;
    }
}

// x [N][H][W][C] -> p [N][H/2][W/2][C], am [N][H/2][W/2][C/4] (one byte per channel quad: 2 bits per channel, window position
// 2 dh + dw of the first maximum).  One thread per pooled float4.
__global__ __launch_bounds__(256) void bias_relu_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                                 float* __restrict__ p, uint8_t* __restrict__ am, long long n4out,
                                                                 int c4, int Wo, int Ho) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4out) return;
    const int q = (int)(i % c4);
    const long long pix = i / c4;                       // (n Ho + ho) Wo + wo
    const int wo = (int)(pix % Wo);
    const long long nh = pix / Wo;                      // n Ho + ho
    const int ho = (int)(nh % Ho);
    const long long n = nh / Ho;
    const long long W = 2ll * Wo;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const long long base = ((n * 2 * Ho + 2 * ho) * W + 2 * wo) * c4 + q;       // float4 index of window element (0, 0)
    f32x4 v[4];
    v[0] = x4[base];
    v[1] = x4[base + c4];
    v[2] = x4[base + W * c4];
    v[3] = x4[base + W * c4 + c4];
    const f32x4 bb = reinterpret_cast<const f32x4*>(b)[q];
    f32x4 m = relu4(v[0] + bb);
    unsigned code = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const f32x4 a = relu4(v[k] + bb);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool take = (a[t] > m[t]) || (a[t] != a[t]);          // ATen max_pool2d: (val > max) || isnan(val)
            m[t] = take ? a[t] : m[t];
            code = take ? ((code & ~(3u << (2 * t))) | ((unsigned)k << (2 * t))) : code;
        }
    }
    reinterpret_cast<f32x4*>(p)[i] = m;
    am[i] = (uint8_t)code;
}

// dp, p [N][Ho][Wo][C], am -> dx [N][2 Ho][2 Wo][C] (every element written), part[block][C] = column sums of dx.
// Workgroup b owns pooled pixels [b ppb, (b + 1) ppb).
__global__ __launch_bounds__(256) void bias_relu_pool_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p,
                                                                 const uint8_t* __restrict__ am, float* __restrict__ dx,
                                                                 float* __restrict__ part, long long npix, long long ppb, int c4,
                                                                 int Wo, int Ho) {
    const int tid = threadIdx.x, q = tid % c4, r0 = tid / c4, rstep = 256 / c4;
    const long long pbeg = (long long)blockIdx.x * ppb, pend = pbeg + ppb < npix ? pbeg + ppb : npix;
    const f32x4* dp4 = reinterpret_cast<const f32x4*>(dp);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
    f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
    const long long W = 2ll * Wo;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long long pix = pbeg + r0; pix < pend; pix += 2 * rstep) {
        f32x4 g[2], v[2];
        unsigned code[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long pp = pix + u * rstep;
            if (pp < pend) { g[u] = dp4[pp * c4 + q]; v[u] = p4[pp * c4 + q]; code[u] = am[pp * c4 + q]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long pp = pix + u * rstep;
            if (pp >= pend) continue;
            const int wo = (int)(pp % Wo);
            const long long nh = pp / Wo;
            const int ho = (int)(nh % Ho);
            const long long n = nh / Ho;
            const long long base = ((n * 2 * Ho + 2 * ho) * W + 2 * wo) * c4 + q;
            f32x4 d;
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = v[u][t] > 0.f ? g[u][t] : 0.f;
            acc += d;
            f32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) o[k][t] = ((code[u] >> (2 * t)) & 3u) == (unsigned)k ? d[t] : 0.f;
            dx4[base] = o[0];
            dx4[base + c4] = o[1];
            dx4[base + W * c4] = o[2];
            dx4[base + W * c4 + c4] = o[3];
        }
    }
    trunk_db_partial(acc, part, c4);
}

// The end of a ResNet bottleneck (model/backbone/resnet.py:89-136: `out += identity; out = relu(out)`): y = max(a + b, 0) in place
// on a - one pass (two reads, one write) where the framework runs add_ and relu_ (three reads, two writes).  Any dense layout:
// the two operands only have to share it.
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(float* __restrict__ a, const float* __restrict__ b, long long n4) {
    f32x4* a4 = reinterpret_cast<f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        f32x4 u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n4) { u[k] = a4[i + k * stride]; v[k] = b4[i + k * stride]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n4) a4[i + k * stride] = relu4(u[k] + v[k]);
    }
}

// g = dy where y > 0 else 0 (the gradient of BOTH operands of add_relu)
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ g, long long n4) {
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        f32x4 u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n4) { u[k] = d4[i + k * stride]; v[k] = y4[i + k * stride]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n4) {
                f32x4 r;
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = v[k][t] > 0.f ? u[k][t] : 0.f;
                g4[i + k * stride] = r;
            }
    }
}

static inline bool trunk_c_ok(int C) {              // 256 % (C / 4) == 0: a column quad per thread, whole rows per workgroup pass
    return C >= 4 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0;
}
static inline int trunk_blocks(long long rows) {
    long long b = (rows + 63) / 64;                  // at least 64 rows per workgroup
    return (int)(b < 1 ? 1 : (b > TRUNK_PART_BLOCKS ? TRUNK_PART_BLOCKS : b));
}

}  // namespace hk

using namespace hk;

extern "C" size_t hk_trunk_ws_bytes(int C) { return C > 0 ? (size_t)TRUNK_PART_BLOCKS * C * sizeof(float) : 0; }

extern "C" int hk_bias_relu_fwd(float* x, const float* bias, uint8_t* mask, long long rows, int C, hk_stream_t stream) {
    if (!x || !bias || rows <= 0 || C <= 0) return HK_ERR_BAD_ARG;
    if (C % 4 != 0 || !aligned16(x) || !aligned16(bias)) return HK_ERR_UNSUPPORTED;
    const long long n4 = rows * (C / 4);
    long long blocks = (n4 + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    if (mask) hipLaunchKernelGGL(bias_relu_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, mask, n4, C / 4);
    else hipLaunchKernelGGL(bias_relu_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, mask, n4, C / 4);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bias_relu_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, float* dbias, long long rows, int C,
                                void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!dy || (!y && !mask) || !dx || !dbias || rows <= 0 || C <= 0) return HK_ERR_BAD_ARG;
    if (!trunk_c_ok(C) || !aligned16(dy) || (!mask && !aligned16(y)) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hk_trunk_ws_bytes(C)) return HK_ERR_WORKSPACE;
    const int nblk = trunk_blocks(rows);
    const long long rpb = (rows + nblk - 1) / nblk;
    if (mask) hipLaunchKernelGGL(bias_relu_bwd_kernel<true>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, y, mask, dx, (float*)ws, rows, rpb, C / 4);
    else hipLaunchKernelGGL(bias_relu_bwd_kernel<false>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, y, mask, dx, (float*)ws, rows, rpb, C / 4);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(trunk_db_final_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, C, dbias);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bias_relu_pool_fwd(const float* x, const float* bias, float* p, uint8_t* argmax, int N, int H, int W, int C,
                                     hk_stream_t stream) {
    if (!x || !bias || !p || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0) return HK_ERR_BAD_ARG;
    if (C % 4 != 0 || H % 2 != 0 || W % 2 != 0 || !aligned16(x) || !aligned16(bias) || !aligned16(p)) return HK_ERR_UNSUPPORTED;
    const long long n4out = (long long)N * (H / 2) * (W / 2) * (C / 4);
    if ((n4out + 255) / 256 > 0x7fffffffll) return HK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bias_relu_pool_fwd_kernel, dim3((unsigned)((n4out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, bias, p,
                       argmax, n4out, C / 4, W / 2, H / 2);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_bias_relu_pool_bwd(const float* dp, const float* p, const uint8_t* argmax, float* dx, float* dbias, int N, int H,
                                     int W, int C, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!dp || !p || !argmax || !dx || !dbias || N <= 0 || H <= 0 || W <= 0 || C <= 0) return HK_ERR_BAD_ARG;
    if (!trunk_c_ok(C) || H % 2 != 0 || W % 2 != 0 || !aligned16(dp) || !aligned16(p) || !aligned16(dx)) return HK_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hk_trunk_ws_bytes(C)) return HK_ERR_WORKSPACE;
    const long long npix = (long long)N * (H / 2) * (W / 2);
    const int nblk = trunk_blocks(npix);
    const long long ppb = (npix + nblk - 1) / nblk;
    hipLaunchKernelGGL(bias_relu_pool_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dp, p, argmax, dx, (float*)ws, npix,
                       ppb, C / 4, W / 2, H / 2);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(trunk_db_final_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, C, dbias);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_add_relu_fwd(float* a, const float* b, long long n, hk_stream_t stream) {
    if (!a || !b || n <= 0) return HK_ERR_BAD_ARG;
    if (n % 4 != 0 || !aligned16(a) || !aligned16(b)) return HK_ERR_UNSUPPORTED;
    long long blocks = (n / 4 + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_relu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, n / 4);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_relu_mask_bwd(const float* dy, const float* y, float* g, long long n, hk_stream_t stream) {
    if (!dy || !y || !g || n <= 0) return HK_ERR_BAD_ARG;
    if (n % 4 != 0 || !aligned16(dy) || !aligned16(y) || !aligned16(g)) return HK_ERR_UNSUPPORTED;
    long long blocks = (n / 4 + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, g, n / 4);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

namespace hk {

// ---------------------------------------------------------------------------------------------------------------------------
// The FIRST convolution of the trunk (model/backbone/vgg.py:24-57: Conv2d(3, 64, 3, padding=1) + ReLU; 64 x 3 x 448 x 448 in,
// a 3.29 GB map out) as one kernel per direction.  27 multiply-adds per output are nothing for a matrix pipe to chew on - the
// layer is the WRITE of its output (forward) and the READ of the output's gradient (backward) - yet the framework spends five
// launches on it: the library's convolution (1.36 ms) + the bias / ReLU pass over the map (1.33 ms), and backwards the ReLU /
// bias-gradient pass (1.17 ms, writes a second 3.29 GB map) + the library's weight gradient that reads it back (1.31 ms):
// 5.2 ms of the 193 ms BCNN step (profiles/r6_step_BCNN_kernel_stats.csv).  Here:
//   conv1_fwd   y = max(conv(x, w) + b, 0) and the sign mask, written once          (reads 0.15 GB, writes 3.29 + 0.21 GB)
//   conv1_bwd   dW = sum_pix (dy o mask) (x) patch(x), db = sum_pix (dy o mask)      (reads 3.29 + 0.21 + 0.15 GB; the masked
//               gradient map is never written - the images need no gradient)
// Forward: a VALU kernel, a thread owns a pixel (64 accumulators, the weights wave-uniform in scalar registers, packed FMAs);
// backward: a 64 x 28 x 12.8 M GEMM on the matrix pipe (see conv1_bwd_kernel).  No LDS on the operand side of either.
// Weights are handed over TRANSPOSED, wt [27][64] with tap = (kh * 3 + kw) * Cin + c (a 7 KB permute of the layer's weight
// per call, made by the caller); dW comes back in the same layout.  Cin <= 3 (27 taps + the ones column fit one 32-wide MFMA
// tile), Cout = 64, stride 1, pad 1.
// Fixed summation orders (taps in order; pixels in row order per wave, waves and workgroups in order): deterministic.

constexpr int C1_OUT = 64;

// forward: thread = pixel (linear index over N H W), 256 pixels per workgroup
template <int CIN>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        uint8_t* __restrict__ mask, long long npix, int H, int W) {
    constexpr int NT = 9 * CIN;
    __shared__ __attribute__((aligned(16))) float tile[4][64][68];      // per wave: 64 pixels x 64 channels, pitch 68 (output turn-table)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p0 = (long long)blockIdx.x * 256;
    const long long p = p0 + tid;
    const bool live = p < npix;
    const long long pc = live ? p : npix - 1;
    const int w_ = (int)(pc % W);
    const long long nh = pc / W;
    const int h_ = (int)(nh % H);
    // this pixel's patch, zero outside the image
    float xv[NT];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = h_ + kh - 1;
        const bool rok = hh >= 0 && hh < H;
        const float* row = x + ((nh - h_ + (rok ? hh : h_)) * W) * CIN;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ww = w_ + kw - 1;
            const bool ok = rok && ww >= 0 && ww < W;
            const float* px = row + (long long)(ok ? ww : w_) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float v = px[c];
                xv[(kh * 3 + kw) * CIN + c] = ok ? v : 0.f;
            }
        }
    }
    float acc[C1_OUT];
#pragma unroll
    for (int o = 0; o < C1_OUT; ++o) acc[o] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int o = 0; o < C1_OUT; ++o) acc[o] = fmaf(xv[t], wt[t * C1_OUT + o], acc[o]);     // wt[..]: wave-uniform (scalar loads)
    }
    // bias, ReLU, the wave's 64 x 64 block turned through LDS so that it leaves as whole 256-byte pixel rows
    float (*T)[68] = tile[wave];
#pragma unroll
    for (int o4 = 0; o4 < C1_OUT / 4; ++o4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = acc[4 * o4 + j] + bias[4 * o4 + j];
            v[j] = s < 0.f ? 0.f : s;
        }
        *reinterpret_cast<f32x4*>(&T[lane][4 * o4]) = v;
    }
    HK_WAVE_SYNC();
    const long long pw = p0 + 64 * wave;                              // first pixel of this wave
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int pl = 4 * i + (lane >> 4), q = lane & 15;           // pixel of the wave, channel quad
        const f32x4 v = *reinterpret_cast<const f32x4*>(&T[pl][4 * q]);
        if (pw + pl < npix) {
            reinterpret_cast<f32x4*>(y)[(pw + pl) * 16 + q] = v;
            if (mask) mask[(pw + pl) * 16 + q] = (uint8_t)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
        }
    }
}

// backward: dW^T [tap][o] = sum over pixels of patch[pix][tap] g[pix][o] is a GEMM with a 12.8 M-deep reduction and a 64 x 28
// result - matrix-pipe work after all: v_mfma_f32_32x32x2f32 with A = g (rows = output channels of a 32-channel half, k = two
// pixels), B = the two pixels' patches (columns = taps, column NT = 1.0: that column of the result is dbias).  A wave walks whole
// image rows (row = n H + h), sixteen pixels per step with every load of the step in flight before its first MFMA: g as it
// lies (a lane reads one float: 128-byte runs), the mask byte, and the patch value as a per-lane gather from the (cached,
// 0.15 GB) input.  (The VALU form - lane = channel, 27 accumulators, patch values in scalar registers - stalled on its
// scalar loads: 2.4 ms where this takes the time of the read.)  part [workgroup][NT + 1][64]
template <int CIN>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ x, float* __restrict__ part, long long nrows,
                                                        int H, int W) {
    constexpr int NT = 9 * CIN;
    static_assert(NT + 1 <= 32, "taps + the ones column fit one 32-wide MFMA tile");
    constexpr int UN = 8;                                                // MFMA steps (pixel pairs) per loop iteration
    __shared__ float red[4][NT + 1][64];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this lane's column of B: tap l31 = (kh * 3 + kw) * CIN + c, or the ones column, or nothing
    const bool is_tap = l31 < NT, is_one = l31 == NT;
    const int kh = is_tap ? l31 / (3 * CIN) : 1, kw = is_tap ? (l31 / CIN) % 3 : 1, c = is_tap ? l31 % CIN : 0;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < nrows; r += nwaves) {        // (uniform per wave)
        const int h_ = (int)(r % H);
        const bool rowok = is_tap && (kh == 0 ? h_ > 0 : (kh == 2 ? h_ + 1 < H : true));
        const float* xb = x + ((r + (rowok ? kh - 1 : 0)) * (long long)W) * CIN + c;     // + (w + kw - 1) CIN
        const float* dyr = dy + r * (long long)W * C1_OUT + l31;
        const uint8_t* mr = mask + r * (long long)W * (C1_OUT / 4) + (l31 >> 2);
        for (int w0 = 0; w0 < W; w0 += 2 * UN) {
            float a0[UN], a1[UN], b[UN];
            unsigned m0[UN], m1[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int w_ = w0 + 2 * u + hf;
                const int wc = w_ < W ? w_ : W - 1;
                a0[u] = dyr[(long long)wc * C1_OUT];
                a1[u] = dyr[(long long)wc * C1_OUT + 32];
                m0[u] = mr[(long long)wc * (C1_OUT / 4)];
                m1[u] = mr[(long long)wc * (C1_OUT / 4) + 8];
                const int ww = w_ + kw - 1;
                const bool okb = rowok && w_ < W && ww >= 0 && ww < W;
                const float v = xb[(long long)(okb ? ww : wc) * CIN];
                b[u] = okb ? v : ((is_one && w_ < W) ? 1.0f : 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool in = w0 + 2 * u + hf < W;
                const float g0 = (in && ((m0[u] >> (l31 & 3)) & 1u)) ? a0[u] : 0.f;
                const float g1 = (in && ((m1[u] >> (l31 & 3)) & 1u)) ? a1[u] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, b[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, b[u], acc1, 0, 0, 0);
            }
        }
    }
    // C layout of the 32 x 32 MFMA: column = lane & 31 (tap), row = 8 (i / 4) + 4 (lane >> 5) + i % 4 (channel of the half)
    if (l31 <= NT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 8 * (i >> 2) + 4 * hf + (i & 3);
            red[wave][l31][o] = acc0[i];
            red[wave][l31][32 + o] = acc1[i];
        }
    }
    __syncthreads();
    float* pb = part + (long long)blockIdx.x * (NT + 1) * 64;
    for (int e = threadIdx.x; e < (NT + 1) * 64; e += 256) {
        const int t = e >> 6, o = e & 63;
        pb[e] = (red[0][t][o] + red[1][t][o]) + (red[2][t][o] + red[3][t][o]);
    }
}

// dwt [NT][64] and db [64] = the workgroup partials added in a fixed order: 64 elements per workgroup, sixteen interleaved chains
// per element (eight loads in flight each), combined in order
__global__ __launch_bounds__(1024) void conv1_bwd_final_kernel(const float* __restrict__ part, int nblk, int nel, int nt64,
                                                               float* __restrict__ dwt, float* __restrict__ db) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;
    float s = 0.f;
    if (e < nel) {
        int k = q;
        for (; k + 7 * 16 < nblk; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + 16 * u) * nel + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nblk; k += 16) s += part[(long long)k * nel + e];
    }
    red[q][l] = s;
    __syncthreads();
    if (q == 0 && e < nel) {
        float t = red[0][l];
        for (int k = 1; k < 16; ++k) t += red[k][l];
        if (e < nt64) dwt[e] = t;
        else db[e - nt64] = t;
    }
}

constexpr int C1_BWD_BLOCKS = 1024;          // four workgroups per CU: sixteen waves to hide a step's load latency behind the others' FMAs

}  // namespace hk

using namespace hk;

extern "C" size_t hk_conv1_ws_bytes(int Cin) {           // (for any Cin > 0: the workspace check comes before the shape check)
    return Cin > 0 ? (size_t)C1_BWD_BLOCKS * (9 * (Cin > 3 ? 3 : Cin) + 1) * 64 * sizeof(float) : 0;
}

extern "C" int hk_conv1_bias_relu_fwd(const float* x, const float* wt, const float* bias, float* y, uint8_t* mask, int N, int H, int W,
                                      int Cin, int Cout, hk_stream_t stream) {
    if (!x || !wt || !bias || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return HK_ERR_BAD_ARG;
    if (Cin > 3 || Cout != C1_OUT || !aligned16(y)) return HK_ERR_UNSUPPORTED;
    const long long npix = (long long)N * H * W;
    const long long blocks = (npix + 255) / 256;
    if (blocks > 0x7fffffffll) return HK_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks);
    switch (Cin) {
        case 1: hipLaunchKernelGGL(conv1_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, mask, npix, H, W); break;
        case 2: hipLaunchKernelGGL(conv1_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, mask, npix, H, W); break;
        default: hipLaunchKernelGGL(conv1_fwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, mask, npix, H, W); break;
    }
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_conv1_bias_relu_bwd(const float* dy, const uint8_t* mask, const float* x, float* dwt, float* dbias, int N, int H, int W,
                                      int Cin, int Cout, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!dy || !mask || !x || !dwt || !dbias || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_conv1_ws_bytes(Cin)) return HK_ERR_WORKSPACE;
    if (Cin > 3 || Cout != C1_OUT) return HK_ERR_UNSUPPORTED;
    const long long nrows = (long long)N * H;
    long long nblk = (nrows + 3) / 4;
    if (nblk > C1_BWD_BLOCKS) nblk = C1_BWD_BLOCKS;
    float* part = (float*)ws;
    switch (Cin) {
        case 1: hipLaunchKernelGGL(conv1_bwd_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, mask, x, part, nrows, H, W); break;
        case 2: hipLaunchKernelGGL(conv1_bwd_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, mask, x, part, nrows, H, W); break;
        default: hipLaunchKernelGGL(conv1_bwd_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dy, mask, x, part, nrows, H, W); break;
    }
    HK_LAUNCH_CHECK();
    const int nel = (9 * Cin + 1) * 64;
    hipLaunchKernelGGL(conv1_bwd_final_kernel, dim3((nel + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)part, (int)nblk, nel,
                       9 * Cin * 64, dwt, dbias);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
