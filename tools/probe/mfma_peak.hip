// Measurement probe (not part of libhawkeye_hip.so): what the fp32 matrix pipe of an MI355X sustains when NOTHING but
// MFMAs is issued - the ceiling against which the `frac` of every MFMA-bound kernel in this repo has to be read.
//   make -C tools/probe && python tools/mfma_peak.py
// Each wave keeps `NACC` independent accumulators and issues v_mfma_f32_16x16x4_f32 (or 32x32x2) back to back on random
// (finite, mixed-sign) operands held in registers; no LDS, no memory traffic inside the loop.
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma16_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    f32x4 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + n) & 3], acc[n], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int n = 1; n < NACC; ++n) s += acc[n];
    out[t] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(512, 2) void mfma32_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + n) & 3], acc[n], 0, 0, 0);
    }
    f32x16 s = acc[0] + acc[1] + acc[2] + acc[3];
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s[i];
    out[t] = r;
}

// 32x32x2 with operands that CHANGE from one MFMA to the next (16 + 16 random registers walked with different strides):
// the kernels above feed every MFMA of an accumulator the same four values, a real GEMM never does - the operand buses
// toggle, and the power manager answers (tools/clock_probe.py).
__global__ __launch_bounds__(512, 2) void mfma32v_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = in[(t * 32 + i) & 65535]; b[i] = in[(t * 32 + 16 + i) & 65535]; }
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 5 * n) & 15], b[(3 * k + n) & 15], acc[n], 0, 0, 0);
    }
    f32x16 s = acc[0] + acc[1] + acc[2] + acc[3];
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s[i];
    out[t] = r;
}

// kind 0: 16x16x4, 8 accumulators per wave; 1: 16x16x4, 16 accumulators; 2: 32x32x2, 4 accumulators;
// 3: 32x32x2, 4 accumulators, operands changing every instruction (iters a multiple of 4).
// Returns the MFMA count per wave (so the caller computes FLOPs: 2048 per 16x16x4, 4096 per 32x32x2), < 0 on error.
extern "C" long long hk_probe_mfma(const float* in, float* out, int kind, int blocks, int threads, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(mfma16_kernel<8>, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(mfma16_kernel<16>, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else if (kind == 3) hipLaunchKernelGGL(mfma32v_kernel, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    if (hipGetLastError() != hipSuccess) return -1;
    return (long long)iters * 4 * (kind == 0 ? 8 : kind == 1 ? 16 : 4);
}
