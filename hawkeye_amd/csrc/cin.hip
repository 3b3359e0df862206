// CIN channel interaction: the Gram primitive at C = 2048, HW = 49 followed by a row softmax and a
// second product, plus the contrastive variant that mixes the interaction matrices of two batch halves.
// replaces ChannelInteractionModule.forward, model/methods/CIN.py:24-60 (the bmm / softmax / abs / bmm parts; the 3x3
// convolution, the residual and the 1-output fc stay on PyTorch-ROCm).
//
//   SCI:  W = softmax_rows(-X X^T / HW) ; Y = W X                          hk_cin_sci_fwd / hk_cin_sci_bwd
//   CCI:  Wc[b] = | W[b] - w_b W[partner(b)] | ; Yc = Wc X                 hk_cin_cci_fwd / hk_cin_cci_bwd
//         partner(b) = (b + B/2) mod B  (CIN.py:45-51: the two batch halves are contrast pairs)
// Both products and all four backward products run on the f32-MFMA GEMM (hk_bgemm.h); |W - w W'| is never stored:
// it is formed in the operand loader (LdAbsDiff) for Yc and for dX.  Row softmax and its backward are one workgroup
// per row, fixed reduction order.
#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

// in place: row <- softmax(row), n columns
__global__ __launch_bounds__(256) void cin_softmax_rows_kernel(float* __restrict__ w, int n) {
    __shared__ float red[4];
    float* p = w + (long long)blockIdx.x * n;
    float m = -3.402823466e38f;
    for (int c = threadIdx.x; c < n; c += 256) m = fmaxf(m, p[c]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) {
        const float e = expf(p[c] - m);
        p[c] = e;
        s += e;
    }
    s = block_sum<4>(s, red);
    for (int c = threadIdx.x; c < n; c += 256) p[c] = p[c] / s;
}

// in place on dW: dG = -dS = -W (.) (dW - <dW, W>_row)        (S = -G, softmax backward)
__global__ __launch_bounds__(256) void cin_softmax_bwd_rows_kernel(const float* __restrict__ w, float* __restrict__ dw,
                                                                  int n) {
    __shared__ float red[4];
    const float* p = w + (long long)blockIdx.x * n;
    float* q = dw + (long long)blockIdx.x * n;
    float t = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) t += p[c] * q[c];
    t = block_sum<4>(t, red);
    for (int c = threadIdx.x; c < n; c += 256) q[c] = -p[c] * (q[c] - t);
}

// |W[b] - w_b W[partner]| formed on the fly; (r, c) index the C x C matrix in memory order
struct LdAbsDiff {
    const float* p;      // W [B][C][C]
    const float* wt;     // w [B]
    int C, B;
    __device__ __forceinline__ void begin(int, int, int) {}
    __device__ __forceinline__ void finish(int, int, int, int, float*) {}
    __device__ __forceinline__ float4 ld4(int b, int r, int c) const {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < C) {
            const int pb = (b + B / 2) % B;
            const float wb = wt[b];
            const float* a = p + ((long long)b * C + r) * C;
            const float* o = p + ((long long)pb * C + r) * C;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (c + t < C) v[t] = fabsf(a[c + t] - wb * o[c + t]);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// dW[b] (+)= sign(D_b) (.) dWc[b] - w_pb sign(D_pb) (.) dWc[pb],  D_b = W[b] - w_b W[pb]   (sign(0) = 0: torch's abs)
// dwpart[b][blk] = - sum over this block's elements of sign(D_b) dWc[b] W[pb]
__global__ __launch_bounds__(256) void cin_cci_dw_kernel(const float* __restrict__ W, const float* __restrict__ wt,
                                                        const float* __restrict__ dWc, float* __restrict__ dW,
                                                        float* __restrict__ dwpart, int C, int B, int nblk) {
    __shared__ float red[4];
    const int b = blockIdx.y, pb = (b + B / 2) % B;
    const long long n = (long long)C * C;
    const float wb = wt[b], wp = wt[pb];
    const float* Wb = W + b * n;
    const float* Wp = W + pb * n;
    const float* Gb = dWc + b * n;
    const float* Gp = dWc + pb * n;
    float acc = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)nblk * 256) {
        const float a = Wb[e], o = Wp[e];
        const float db = a - wb * o, dp = o - wp * a;
        const float sb = (db > 0.f) ? 1.f : ((db < 0.f) ? -1.f : 0.f);
        const float sp = (dp > 0.f) ? 1.f : ((dp < 0.f) ? -1.f : 0.f);
        dW[b * n + e] = sb * Gb[e] - wp * sp * Gp[e];
        acc += sb * Gb[e] * o;
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) dwpart[(long long)b * nblk + blockIdx.x] = -acc;
}

__global__ __launch_bounds__(64) void cin_cci_dw_reduce_kernel(const float* __restrict__ dwpart, float* __restrict__ dwt,
                                                              int nblk) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nblk; ++i) s += dwpart[(long long)b * nblk + i];
        dwt[b] = s;
    }
}

constexpr int CIN_DW_BLOCKS = 64;

}  // namespace hk

using namespace hk;

#define HK_TRY(x)                       \
    do {                                \
        int rc__ = (x);                 \
        if (rc__ != HK_OK) return rc__; \
    } while (0)

extern "C" int hk_cin_sci_fwd(const float* x, float* w, float* y, int B, int C, int HW, hk_stream_t stream) {
    if (!x || !w || !y || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const LdPlain lx = make_plain(x, (long long)C * HW, HW, C, HW);
    HK_TRY((bgemm_launch<true, true>(lx, lx, make_affine(w, (long long)C * C, C, -1.0f / (float)HW, nullptr, 0.f, 0.f), C, C,
                                     HW, B, st)));                                              // -X X^T / HW   :31-32
    hipLaunchKernelGGL(cin_softmax_rows_kernel, dim3((unsigned)B * C), dim3(256), 0, st, w, C);
    HK_LAUNCH_CHECK();
    const LdPlain lw = make_plain(w, (long long)C * C, C, C, C);
    return bgemm_launch<true, false>(lw, lx, make_affine(y, (long long)C * HW, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B,
                                     st);                                                        // Y = W X       :34
}

// dx = W^T dY + (dG + dG^T) X / HW with dG = softmax-backward of (dY X^T + dw_extra).  `dwbuf` [B,C,C] is scratch; on
// entry it holds dw_extra (the gradient that reaches W from the CCI branch) when has_extra != 0.
extern "C" int hk_cin_sci_bwd(const float* x, const float* w, const float* dy, float* dwbuf, int has_extra, float* dx,
                              int B, int C, int HW, hk_stream_t stream) {
    if (!x || !w || !dy || !dwbuf || !dx || B <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long sx = (long long)C * HW, sw = (long long)C * C;
    const LdPlain lx = make_plain(x, sx, HW, C, HW);
    const LdPlain ldy = make_plain(dy, sx, HW, C, HW);
    const LdPlain lw = make_plain(w, sw, C, C, C);
    HK_TRY((bgemm_launch<true, true>(ldy, lx, make_affine(dwbuf, sw, C, 1.f, nullptr, has_extra ? 1.f : 0.f, 0.f), C, C, HW,
                                     B, st)));                                                   // dW = dY X^T (+ extra)
    hipLaunchKernelGGL(cin_softmax_bwd_rows_kernel, dim3((unsigned)B * C), dim3(256), 0, st, w, dwbuf, C);
    HK_LAUNCH_CHECK();
    HK_TRY((bgemm_launch<false, false>(lw, ldy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));  // W^T dY
    LdSym ls;
    ls.p = dwbuf; ls.bs = sw; ls.d = C;
    return bgemm_launch<true, false>(ls, lx, make_affine(dx, sx, HW, 1.0f / (float)HW, nullptr, 1.f, 0.f), C, HW, C, B, st);
}

extern "C" int hk_cin_cci_fwd(const float* x, const float* w, const float* wt, float* y, int B, int C, int HW,
                              hk_stream_t stream) {
    if (!x || !w || !wt || !y || B <= 0 || (B & 1) || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    LdAbsDiff la;
    la.p = w; la.wt = wt; la.C = C; la.B = B;
    const LdPlain lx = make_plain(x, (long long)C * HW, HW, C, HW);
    return bgemm_launch<true, false>(la, lx, make_affine(y, (long long)C * HW, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B,
                                     (hipStream_t)stream);                                       // :52-54
}

extern "C" size_t hk_cin_cci_ws_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    return ((size_t)B * C * C + (size_t)B * CIN_DW_BLOCKS) * sizeof(float) + 256;
}

// dy [B,C,HW] -> dx [B,C,HW] (through Wc X), dw [B,C,C] (gradient reaching W), dwt [B] (gradient of the weights w_b)
extern "C" int hk_cin_cci_bwd(const float* x, const float* w, const float* wt, const float* dy, float* dx, float* dw,
                              float* dwt, int B, int C, int HW, void* ws, size_t ws_bytes, hk_stream_t stream) {
    if (!x || !w || !wt || !dy || !dx || !dw || !dwt || B <= 0 || (B & 1) || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    if (!ws || ws_bytes < hk_cin_cci_ws_bytes(B, C)) return HK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long sx = (long long)C * HW, sw = (long long)C * C;
    float* dwc = (float*)ws;                       // dL/dWc = dY X^T
    float* dwpart = dwc + (size_t)B * sw;
    const LdPlain lx = make_plain(x, sx, HW, C, HW);
    const LdPlain ldy = make_plain(dy, sx, HW, C, HW);
    HK_TRY((bgemm_launch<true, true>(ldy, lx, make_affine(dwc, sw, C, 1.f, nullptr, 0.f, 0.f), C, C, HW, B, st)));
    LdAbsDiff la;
    la.p = w; la.wt = wt; la.C = C; la.B = B;
    HK_TRY((bgemm_launch<false, false>(la, ldy, make_affine(dx, sx, HW, 1.f, nullptr, 0.f, 0.f), C, HW, C, B, st)));  // Wc^T dY
    hipLaunchKernelGGL(cin_cci_dw_kernel, dim3(CIN_DW_BLOCKS, B), dim3(256), 0, st, w, wt, (const float*)dwc, dw, dwpart, C, B,
                       CIN_DW_BLOCKS);
    HK_LAUNCH_CHECK();
    hipLaunchKernelGGL(cin_cci_dw_reduce_kernel, dim3(B), dim3(64), 0, st, (const float*)dwpart, dwt, CIN_DW_BLOCKS);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
