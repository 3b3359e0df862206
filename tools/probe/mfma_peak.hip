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

// 32x32x2 fed from LDS the way the Gram kernels feed it: per four MFMAs two ds_read_b128 (one A, one B fragment of a
// [64 rows][196 floats] panel, read one step ahead), two accumulator chains, nothing else - no barriers, no global
// traffic.  What an LDS-fed fp32 MFMA loop can reach, and at which clock.
// mode bit 0: a workgroup barrier after every 24 steps (a "tile"); bit 1: the tile's accumulators are summed, sent through
// sqrt / fma (the Gram epilogue's VALU work) and restarted from zero; bit 2: ... and stored (16 B per lane x 4, a
// 32-bit offset that walks 64 MB); bit 3: the next panel's staging - 13 global loads of 16 B per thread (L2 hits) at the
// top of a tile, written to a third LDS panel from inside its MFMA steps: each adds one ingredient of the real kernel
// to the bare loop.
__global__ __launch_bounds__(512, 1) void mfma32lds_kernel(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                           int mode, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 64 * 196];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // operands: one value of `in` per thread, varied arithmetically (no memory-bound fill in front of a short launch)
        const float base = in[(blockIdx.x * 512 + tid) & 65535];
        for (int i = tid; i < 2 * 64 * 196; i += blockDim.x) lds[i] = base + 0.001f * (float)(i & 1023);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5, wm = (wave >> 1) & 1, wn = wave & 1;
    const float* Ap = lds + (wm * 32 + l31) * 196 + 4 * lh;
    const float* Bp = lds + 64 * 196 + (wn * 32 + l31) * 196 + 4 * lh;
    f32x16 acc0, acc1, keep;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; keep[r] = 0.f; }
    unsigned so = (unsigned)t * 16u;
    for (int it = 0; it < iters; it += 24) {
        f32x4 a = *reinterpret_cast<const f32x4*>(Ap), q = *reinterpret_cast<const f32x4*>(Bp);
        f32x4 stg[13];
        if (mode & 8) {                       // the next 50 KB panel: 13 x 16 B per thread from an L2-resident buffer (in[], 256 KB)
            const f32x4* src = reinterpret_cast<const f32x4*>(in) + ((it * 64 + blockIdx.x * 3136) & 8191);
#pragma unroll
            for (int u = 0; u < 13; ++u) stg[u] = src[(tid + 256 * u) & 4095];
        }
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            f32x4 an = a, qn = q;
            if (s + 1 < 24) {
                an = *reinterpret_cast<const f32x4*>(Ap + 8 * (s + 1));
                qn = *reinterpret_cast<const f32x4*>(Bp + 8 * (s + 1));
            }
            if ((mode & 8) && s >= 10 && s < 23) {
                const int f = tid + 256 * (s - 10);
                if (f < 3136) reinterpret_cast<f32x4*>(lds + 2 * 64 * 196)[f] = stg[s - 10];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], q[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], q[3], acc1, 0, 0, 0);
            a = an;
            q = qn;
        }
        asm volatile("" ::: "memory");
        if (mode & 2) {
            f32x16 p = acc0 + acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_sqrtf(fmaf(fabsf(p[r]), 0.0051f, 1e-5f)) * 0.37f; acc0[r] = 0.f; acc1[r] = 0.f; }
            if (mode & 4) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(sink) + ((so + (unsigned)g * 8388608u) & 67108863u)) =
                        (f32x4){p[4 * g], p[4 * g + 1], p[4 * g + 2], p[4 * g + 3]};
                so += 16u * 131072u;
            }
            keep += p;
        }
        if (mode & 1) __syncthreads();
    }
    const f32x16 sres = acc0 + acc1 + keep;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += sres[i];
    out[t] = r;
}

// The same tile loop with EIGHT waves: the two waves of a SIMD split K (12 steps each), exchange half of their partial
// sums through 8 KB of LDS (each then owns 8 of the sub-tile's 16 registers: epilogue arithmetic and stores divide
// evenly), and share the staging of the next panel (7 loads per thread).  One more barrier per tile.  The question:
// does a second wave per SIMD cover the stalls of stores / staging / epilogue that cost the four-wave loop 0.95 -> 0.74?
__global__ __launch_bounds__(512, 1) void mfma32lds8_kernel(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                            float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 64 * 196 + 2048];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const float base = in[(blockIdx.x * 512 + tid) & 65535];
        for (int i = tid; i < 2 * 64 * 196; i += blockDim.x) lds[i] = base + 0.001f * (float)(i & 1023);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5, wm = (wave >> 1) & 1, wn = wave & 1, ks = wave >> 2;
    const float* Ap = lds + (wm * 32 + l31) * 196 + 4 * lh + 96 * ks;
    const float* Bp = lds + 64 * 196 + (wn * 32 + l31) * 196 + 4 * lh + 96 * ks;
    f32x4* xch = reinterpret_cast<f32x4*>(lds + 3 * 64 * 196) + (wave & 3) * 128 + lane;      // 2 KB per wave pair
    f32x16 acc0, acc1;
    f32x4 keep0 = (f32x4){0.f, 0.f, 0.f, 0.f}, keep1 = keep0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    unsigned so = (unsigned)t * 16u;
    for (int it = 0; it < iters; it += 24) {
        f32x4 a = *reinterpret_cast<const f32x4*>(Ap), q = *reinterpret_cast<const f32x4*>(Bp);
        f32x4 stg[7];
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + ((it * 64 + blockIdx.x * 3136) & 8191);
#pragma unroll
        for (int u = 0; u < 7; ++u) stg[u] = src[(tid + 512 * u) & 4095];
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            f32x4 an = a, qn = q;
            if (s + 1 < 12) {
                an = *reinterpret_cast<const f32x4*>(Ap + 8 * (s + 1));
                qn = *reinterpret_cast<const f32x4*>(Bp + 8 * (s + 1));
            }
            if (s >= 4 && s < 11) {
                const int f = tid + 512 * (s - 4);
                if (f < 3136) reinterpret_cast<f32x4*>(lds + 2 * 64 * 196)[f] = stg[s - 4];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], q[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], q[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], q[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], q[3], acc1, 0, 0, 0);
            a = an;
            q = qn;
        }
        asm volatile("" ::: "memory");
        f32x16 p = acc0 + acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // the half this wave does NOT keep goes to its partner (registers 8 .. 15 from the ks = 0 wave, 0 .. 7 from ks = 1),
        // 16 bytes per lane at a time: 8 KB is all the LDS three 50 KB panels leave
        f32x4 m0 = ks ? (f32x4){p[8], p[9], p[10], p[11]} : (f32x4){p[0], p[1], p[2], p[3]};
        f32x4 m1 = ks ? (f32x4){p[12], p[13], p[14], p[15]} : (f32x4){p[4], p[5], p[6], p[7]};
        xch[64 * ks] = ks ? (f32x4){p[0], p[1], p[2], p[3]} : (f32x4){p[8], p[9], p[10], p[11]};
        __syncthreads();
        m0 += xch[64 * (1 - ks)];
        __syncthreads();
        xch[64 * ks] = ks ? (f32x4){p[4], p[5], p[6], p[7]} : (f32x4){p[12], p[13], p[14], p[15]};
        __syncthreads();
        m1 += xch[64 * (1 - ks)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m0[r] = __builtin_amdgcn_sqrtf(fmaf(fabsf(m0[r]), 0.0051f, 1e-5f)) * 0.37f;
            m1[r] = __builtin_amdgcn_sqrtf(fmaf(fabsf(m1[r]), 0.0051f, 1e-5f)) * 0.37f;
        }
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(sink) + (so & 67108863u)) = m0;
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(sink) + ((so + 8388608u) & 67108863u)) = m1;
        so += 16u * 131072u;
        keep0 += m0;
        keep1 += m1;
        __syncthreads();
    }
    const f32x16 sres = acc0 + acc1;
    float r = keep0[0] + keep0[1] + keep0[2] + keep0[3] + keep1[0] + keep1[1] + keep1[2] + keep1[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += sres[i];
    out[t] = r;
}

// kind 0: 16x16x4, 8 accumulators per wave; 1: 16x16x4, 16 accumulators; 2: 32x32x2, 4 accumulators;
// 3: 32x32x2, 4 accumulators, operands changing every instruction (iters a multiple of 4);
// 4 + mode: 32x32x2 fed from LDS (iters a multiple of 24; returns iters * 4 MFMAs per wave); mode bits: 1 barrier per
// tile, 2 epilogue arithmetic, 4 stores.
// Returns the MFMA count per wave (so the caller computes FLOPs: 2048 per 16x16x4, 4096 per 32x32x2), < 0 on error.
static float* g_sink = nullptr;        // 64 MB target of the LDS-fed loop's store mode
extern "C" long long hk_probe_mfma(const float* in, float* out, int kind, int blocks, int threads, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(mfma16_kernel<8>, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(mfma16_kernel<16>, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else if (kind == 30) {                       // eight waves, K split between the waves of a SIMD; MFMAs per wave: iters * 2
        if (!g_sink && hipMalloc(&g_sink, 64u << 20) != hipSuccess) return -1;
        hipLaunchKernelGGL(mfma32lds8_kernel, dim3(blocks), dim3(512), 0, st, in, out, iters, g_sink);
        return hipGetLastError() != hipSuccess ? -1 : (long long)iters * 2;
    } else if (kind >= 4 && kind < 20) {          // 4 + mode
        if (!g_sink && hipMalloc(&g_sink, 64u << 20) != hipSuccess) return -1;
        hipLaunchKernelGGL(mfma32lds_kernel, dim3(blocks), dim3(threads), 0, st, in, out, iters, kind - 4, g_sink);
        return hipGetLastError() != hipSuccess ? -1 : (long long)iters * 4;
    } else if (kind == 3) hipLaunchKernelGGL(mfma32v_kernel, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    else hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(threads), 0, st, in, out, iters);
    if (hipGetLastError() != hipSuccess) return -1;
    return (long long)iters * 4 * (kind == 0 ? 8 : kind == 1 ? 16 : 4);
}
