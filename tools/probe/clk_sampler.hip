// Measurement probe (not part of libhawkeye_hip.so): the shader clock the chip actually runs while some OTHER kernel is
// executing.  One wave on one CU reads the shader-cycle counter (s_memtime) and the constant 100 MHz reference counter
// (s_memrealtime) every `period` reference ticks and writes the triples out; cycles / time between two samples is the
// clock.  The datasheet's 157.3 TF/s of fp32 MFMA is 256 CUs x 256 FLOP / cycle x 2.4 GHz: a kernel that makes the
// power manager lower the clock cannot reach it, whatever its structure.      python tools/clock_probe.py
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void clk_sampler_kernel(unsigned long long* __restrict__ out, int n, int period) {
    if (threadIdx.x != 0) return;
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        unsigned long long rt;
        do {
            __builtin_amdgcn_s_sleep(8);
            rt = __builtin_amdgcn_s_memrealtime();
        } while (rt < next);
        const unsigned long long cy = __builtin_readcyclecounter();
        const unsigned long long rt2 = __builtin_amdgcn_s_memrealtime();      // (the cycle read is bracketed: the host drops
        out[3 * i] = cy;                                                      //  samples whose two reference reads are far apart)
        out[3 * i + 1] = rt;
        out[3 * i + 2] = rt2;
        next = rt + (unsigned long long)period;
    }
}

extern "C" int hk_probe_clk_sampler(unsigned long long* out, int n, int period, void* stream) {
    hipLaunchKernelGGL(clk_sampler_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n, period);
    return (int)hipGetLastError();
}
