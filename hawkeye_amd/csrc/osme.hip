// OSME head (replaces model/methods/OSME.py:19-24): the squeeze (GAP) and the
// per-attention channel re-scaling around the tiny excitation MLP.  Pure
// streaming kernels; one wave64 per (n,c) row of the 7x7 map.
#include "hk_common.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

__global__ __launch_bounds__(256) void osme_gap_kernel(const float* __restrict__ x, float* __restrict__ z, long long rows,
                                                       int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += p[i];
    s = wave_sum(s);
    if (lane == 0) z[row] = s / (float)HW;
}

// s[p,n,c,:] = m[p,n,c] * x[n,c,:]
__global__ __launch_bounds__(256) void osme_scale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                             float* __restrict__ s, int P, long long rows, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + row * HW;
    for (int p = 0; p < P; ++p) {
        const float g = m[(long long)p * rows + row];
        float* sp = s + ((long long)p * rows + row) * HW;
        for (int i = lane; i < HW; i += 64) sp[i] = g * xp[i];
    }
}

// dx[n,c,:] = sum_p m_p ds_p (+ dz/HW) ; dm[p,n,c] = sum_hw ds_p x
__global__ __launch_bounds__(256) void osme_scale_bwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                             const float* __restrict__ ds, const float* __restrict__ dz,
                                                             float* __restrict__ dx, float* __restrict__ dm, int P,
                                                             long long rows, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + row * HW;
    const float zc = dz ? dz[row] / (float)HW : 0.f;
    for (int i0 = 0; i0 < HW; i0 += 64) {
        const int i = i0 + lane;
        float acc = zc;
        if (i < HW)
            for (int p = 0; p < P; ++p) acc += m[(long long)p * rows + row] * ds[((long long)p * rows + row) * HW + i];
        if (i < HW) dx[row * HW + i] = acc;
    }
    for (int p = 0; p < P; ++p) {
        const float* dp = ds + ((long long)p * rows + row) * HW;
        float s = 0.f;
        for (int i = lane; i < HW; i += 64) s += dp[i] * xp[i];
        s = wave_sum(s);
        if (lane == 0) dm[(long long)p * rows + row] = s;
    }
}

}  // namespace hk

using namespace hk;

extern "C" int hk_osme_gap(const float* x, float* z, int N, int C, int HW, hk_stream_t stream) {
    if (!x || !z || N <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const long long rows = (long long)N * C;
    hipLaunchKernelGGL(osme_gap_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, z, rows, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_osme_scale_fwd(const float* x, const float* m, float* s, int P, int N, int C, int HW,
                                 hk_stream_t stream) {
    if (!x || !m || !s || P <= 0 || N <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const long long rows = (long long)N * C;
    hipLaunchKernelGGL(osme_scale_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, m, s,
                       P, rows, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}

extern "C" int hk_osme_scale_bwd(const float* x, const float* m, const float* ds, const float* dz, float* dx, float* dm,
                                 int P, int N, int C, int HW, hk_stream_t stream) {
    if (!x || !m || !ds || !dx || !dm || P <= 0 || N <= 0 || C <= 0 || HW <= 0) return HK_ERR_BAD_ARG;
    const long long rows = (long long)N * C;
    hipLaunchKernelGGL(osme_scale_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, m,
                       ds, dz, dx, dm, P, rows, HW);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
