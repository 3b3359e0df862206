// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HK_OK 0
#define HK_ERR_BAD_ARG (-1)
#define HK_ERR_WORKSPACE (-2)
#define HK_ERR_UNSUPPORTED (-3)

#include <hk_isa.h>   // the gfx950-only spellings (dynamic LDS, LDS barriers, buffer stores, pinned instructions, untracked loads)

// s_waitcnt vmcnt(n) + s_barrier: the workgroup barrier that ends a pipeline step of an LDS-DMA stream - waits for all but
// the n most recent vector-memory operations of this wave (the pieces it has just issued), then publishes the stage
#define HK_VMCNT_IMM(n) (((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))     /* gfx9 s_waitcnt: vmcnt only */
#define HK_VM_BARRIER(n)                                                                                       \
    do {                                                                                                       \
        asm volatile("" ::: "memory");                 /* no LDS access moves across */                        \
        __builtin_amdgcn_s_waitcnt(HK_VMCNT_IMM(n));                                                           \
        __builtin_amdgcn_s_barrier();                                                                          \
        asm volatile("" ::: "memory");                                                                         \
    } while (0)

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
    int bwd_v = 0;          // HK_BWD_V         Gram backward: 0 automatic, 1 the four-wave 64-row panel kernel (bcnn_fast.hip)
    int ns_streams = 1;     // HK_NS_STREAMS    n: the batch runs the Newton-Schulz chain in n + 1 parts on n + 1 HIP queues (default 1: two halves), 0: one queue
    int ns_sym = 1;         // HK_NS_SYM        1: hk_ns_sqrtm_fwd_sym skips the tiles below the diagonal blocks, 0: it computes every tile
    int lin_walk = -1;      // HK_LIN_WALK      classifier backward: 1: workgroup s walks chunks s, s + S, ..; 0: a contiguous slab per workgroup;
                            //                  -1: the measured winner per kernel (linear_bwd64_kernel 1, linear_bwd16_kernel 0)
    int fwd_fold = 0;       // HK_FWD_FOLD      hk_bcnn_pool_fwd: 0: one launch (the Gram kernel adds up the sample's columns itself), -1: the
                            //                  column-sum kernel + the Gram kernel
    int bwd_fold = 0;       // HK_BWD_FOLD      hk_bcnn_pool_bwd_tdot: 0: the rank-1 term in the GEMM kernel's epilogue where that kernel runs
                            //                  (one launch), -1: always GEMM + dot product + bcnn_rank1_fix_kernel
    int sched_b = 0;        // HK_SCHED_B       > 0: work-split heuristics that depend on the batch size behave as if it were this (tests: the
                            //                  large-batch schedules on small inputs); results do not depend on it
};
Tuning& tuning();           // api.hip

// One helper HIP queue per device (created on first use, kept for the life of the process) for entry points that run
// two independent halves of a batch side by side: fork = everything enqueued on `st` so far happens before the helper
// queue's work; join = the helper queue's work happens before whatever is enqueued on `st` next.  Event record / wait
// pairs only - no host synchronisation, capturable in a hipGraph like any fork / join.
constexpr int HK_MAX_AUX = 3;
struct AuxQueue {
    hipStream_t aux[HK_MAX_AUX] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[HK_MAX_AUX] = {nullptr, nullptr, nullptr};
    bool ok = false;
    std::mutex busy;            // held from fork to join: two host threads driving one device take turns on the helper queues
};
inline AuxQueue* aux_queue(hipStream_t st) {
    // one slot per device, each created exactly once whatever thread gets there first (std::call_once); after that the
    // slot is read-only.  A slot whose creation failed stays empty and callers fall back to the caller's stream.
    // The device is the one that OWNS the caller's stream (the helper queues fork from / join into that stream); the
    // entry points require it to be the current device as well - the helper queues are created on it - and fall back to
    // one queue otherwise.
    static AuxQueue tab[16];
    static std::once_flag once[16];
    int dev = 0, sdev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (st != nullptr && (hipStreamGetDevice(st, &sdev) != hipSuccess || sdev != dev)) return nullptr;
    AuxQueue& a = tab[dev];
    std::call_once(once[dev], [&a]() {
        bool good = hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < HK_MAX_AUX && good; ++i)
            good = hipStreamCreateWithFlags(&a.aux[i], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&a.join[i], hipEventDisableTiming) == hipSuccess;
        if (!good) {                                   // leave nothing half-made behind
            for (int i = 0; i < HK_MAX_AUX; ++i) {
                if (a.join[i]) (void)hipEventDestroy(a.join[i]);
                if (a.aux[i]) (void)hipStreamDestroy(a.aux[i]);
                a.join[i] = nullptr; a.aux[i] = nullptr;
            }
            if (a.fork) (void)hipEventDestroy(a.fork);
            a.fork = nullptr;
        }
        a.ok = good;
    });
    return a.ok ? &a : nullptr;
}
// Scope of one fork / join onto `n` (<= HK_MAX_AUX) helper queues.  Construction (when n > 0): everything enqueued on
// `st` so far happens before the helper queues' work; join() - or the destructor, on an early error return - makes their
// work happen before whatever is enqueued on `st` next, so a caller that frees or reuses its buffers on `st` after a
// failed call is still ordered behind the parts of the batch that are in flight on the helper queues.
class AuxScope {
  public:
    AuxScope(hipStream_t st, int n) : st_(st) {
        if (n <= 0) return;
        if (n > HK_MAX_AUX) n = HK_MAX_AUX;
        q_ = aux_queue(st);
        if (!q_) return;
        q_->busy.lock();
        bool good = hipEventRecord(q_->fork, st_) == hipSuccess;
        for (int i = 0; i < n && good; ++i) good = hipStreamWaitEvent(q_->aux[i], q_->fork, 0) == hipSuccess;
        if (good) {
            n_ = n;
        } else {
            q_->busy.unlock();
            q_ = nullptr;
        }
    }
    AuxScope(const AuxScope&) = delete;
    AuxScope& operator=(const AuxScope&) = delete;
    ~AuxScope() { (void)join(); }
    int count() const { return n_; }                          // helper queues in use (0: stay on the caller's stream)
    hipStream_t aux(int i) const { return q_->aux[i]; }
    int join() {
        if (!n_) return HK_OK;
        hipError_t e = hipSuccess;
        for (int i = 0; i < n_; ++i) {
            hipError_t r = hipEventRecord(q_->join[i], q_->aux[i]);
            if (r == hipSuccess) r = hipStreamWaitEvent(st_, q_->join[i], 0);
            if (e == hipSuccess) e = r;
        }
        n_ = 0;
        q_->busy.unlock();
        q_ = nullptr;
        return e == hipSuccess ? HK_OK : (int)e;
    }

  private:
    hipStream_t st_;
    int n_ = 0;
    AuxQueue* q_ = nullptr;
};

constexpr int WAVE = 64;
constexpr int NXCD = 8;

// LDS-DMA: one global_load_lds_dwordx4 - lane l of the wave copies the 16 bytes at ITS global address g to the
// wave-uniform LDS address l + 16 l (1 KB per wave-instruction, no registers, counted in vmcnt)
template <int AUX = 0>
__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

// ... and its 4-byte form (global_load_lds_dword): lane l copies 4 bytes to the wave-uniform LDS address l + 4 l - 256 bytes
// per instruction, for the tail of a run that is not a whole number of 1 KB pieces
__device__ __forceinline__ void glds4(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

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
