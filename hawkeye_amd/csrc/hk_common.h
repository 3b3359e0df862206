// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HK_OK 0
#define HK_ERR_BAD_ARG (-1)
#define HK_ERR_WORKSPACE (-2)
#define HK_ERR_UNSUPPORTED (-3)

// The two spellings below have no meaning off the GPU; the CPU emulation used by the test tier (tests/emu) supplies
// its own before this header is read.
#ifndef HK_DYN_LDS     // dynamic LDS of the launch as `float name[]` (HK_DYN_LDS16: declared 16-byte aligned)
#define HK_DYN_LDS(name) extern __shared__ float name[]
#define HK_DYN_LDS16(name) extern __shared__ __attribute__((aligned(16))) float name[]
#endif
#ifndef HK_LDS_BARRIER  // workgroup barrier that waits for LDS traffic only (global loads stay in flight across it)
#define HK_LDS_BARRIER()                                   \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)
#endif

#define HK_LAUNCH_CHECK()                                 \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return (int)e__;           \
    } while (0)

namespace hk {

// A/B levers for tests, benchmarks and profiling (hk_tuning_set / hk_tuning_get in the C ABI).  The values are seeded
// from the environment ONCE, when the library is first used; the launch paths read plain ints - no getenv per call.
// The product never sets them: every default is the measured winner.
struct Tuning {
    int bcnn_generic = 0;   // HK_BCNN_GENERIC  1: generic GEMM path instead of the panel-resident Gram / backward kernels
    int cbp_bin = -1;       // HK_CBP_BIN      -1: by batch size, 0: row-sketch, 1: CSR gather, 2: row-scatter
    int roi_bwd = 0;        // HK_ROI_BWD       0: uniform-window ROI-refinement backward (apcnn_roi2.hip), 1: the round-1 table kernel
    int linear_slabs = 0;   // HK_LINEAR_SLABS  0: automatic split-K slab count of hk_linear_fwd
    int ns_tn = 0;          // HK_NS_TN         0: automatic, 64 / 128: forced tile width of the Newton-Schulz products
    int bwd_v = 0;          // HK_BWD_V         Gram backward: 0 / 1 the 64-row kernel (bcnn_fast.hip), 5 the 128-row kernel (hk_bwd128.h)
    int ns_streams = 1;     // HK_NS_STREAMS    1: the two halves of the batch run the Newton-Schulz chain on two HIP queues (default), 0: one queue
};
Tuning& tuning();           // api.hip

// One helper HIP queue per device (created on first use, kept for the life of the process) for entry points that run
// two independent halves of a batch side by side: fork = everything enqueued on `st` so far happens before the helper
// queue's work; join = the helper queue's work happens before whatever is enqueued on `st` next.  Event record / wait
// pairs only - no host synchronisation, capturable in a hipGraph like any fork / join.
struct AuxQueue {
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
inline AuxQueue* aux_queue() {
    static AuxQueue tab[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    AuxQueue& a = tab[dev];
    if (!a.aux) {
        if (hipStreamCreateWithFlags(&a.aux, hipStreamNonBlocking) != hipSuccess) { a.aux = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &a;
}
// the helper queue, ordered after `st`; nullptr when it cannot be had (the caller then stays on `st`)
inline hipStream_t aux_fork(hipStream_t st) {
    AuxQueue* a = aux_queue();
    if (!a) return nullptr;
    if (hipEventRecord(a->fork, st) != hipSuccess || hipStreamWaitEvent(a->aux, a->fork, 0) != hipSuccess) return nullptr;
    return a->aux;
}
inline int aux_join(hipStream_t aux, hipStream_t st) {
    if (!aux) return HK_OK;
    AuxQueue* a = aux_queue();
    hipError_t e = hipEventRecord(a->join, aux);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, a->join, 0);
    return e == hipSuccess ? HK_OK : (int)e;
}

constexpr int WAVE = 64;
constexpr int NXCD = 8;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x = NW*64 threads; `red` is NW floats of LDS.
// Fixed reduction order -> bit-reproducible.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// Work-item -> (sample, tile) map that keeps every tile of one batch sample on
// one XCD (blocks are dispatched round-robin, block b -> XCD b % 8; each XCD has
// a private 4 MiB L2, so a sample's operand panels are fetched into ONE L2).
// Speed only: correctness never depends on the placement.
// grid size = hk_xcd_grid(nb, tiles).
__host__ __device__ __forceinline__ int xcd_grid(int nb, int tiles) {
    return (nb >= NXCD) ? NXCD * ((nb + NXCD - 1) / NXCD) * tiles : nb * tiles;
}
__device__ __forceinline__ bool xcd_map(int bid, int nb, int tiles, int& sample, int& tile) {
    if (nb >= NXCD) {
        const int xcd = bid % NXCD, slot = bid / NXCD;
        sample = xcd + NXCD * (slot / tiles);
        tile = slot % tiles;
    } else {
        sample = bid / tiles;
        tile = bid % tiles;
    }
    return sample < nb;
}

__host__ __forceinline__ bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute belongs to the
// (function, device) pair.  HK_ALLOW_BIG_LDS(fn, bytes) raises it to `bytes` (the launch's dynamic size; static LDS of
// the kernel counts against the same 160 KB, so the attribute is never set higher than needed) once per device and
// size for the call site's kernel, and returns the HIP error from the enclosing function if that fails.
#define HK_ALLOW_BIG_LDS(fn, bytes)                                                                            \
    do {                                                                                                       \
        static size_t have_[32] = {0};                                                                         \
        int dev_ = 0;                                                                                          \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ > 31) dev_ = 0;                             \
        if ((size_t)(bytes) > 64 * 1024 && (size_t)(bytes) > have_[dev_]) {                                    \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),                      \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) return (int)e_;                                                              \
            have_[dev_] = (size_t)(bytes);                                                                     \
        }                                                                                                      \
    } while (0)

}  // namespace hk
