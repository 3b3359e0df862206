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

#ifndef HK_LDS_VOLATILE  // a volatile view of an LDS array: accesses stay in program order AND stay ds_read / ds_write.  A plain
                         // `volatile float*` is a generic pointer: flat_load / flat_store, which also count in vmcnt and
                         // wait behind every global load in flight (hk_cbp_fused.h: 3 us per tile)
#define HK_LDS_VOLATILE(p) ((volatile __attribute__((address_space(3))) float*)(p))
#endif

#ifndef HK_LDS_CONST  // a read-only view of an LDS location through an explicit LDS pointer (constant offsets fold into the
                      // instruction's offset field; through a generic pointer the compiler adds the - zero - LDS base
                      // with a VALU op per access)
#define HK_LDS_CONST(p) ((const __attribute__((address_space(3))) float*)(p))
#endif

// Plain (cached) stores through a buffer descriptor: the hardware drops the lanes whose byte offset lies beyond the
// descriptor's size - a ragged edge needs no predicate, so the instruction ALWAYS issues (a store under `if (row < n)` is
// skipped altogether when no lane passes: its place in a counted s_waitcnt vmcnt(n) would then be taken by an older load)
#ifndef HK_BUF_RSRC
namespace hk {
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t buf_rsrc(const float* base, long long floats) {      // base, floats: wave-uniform
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(floats * 4), 0x00020000);
}
// (the builtin's own vector type is kept behind decltype: converting to a user vector typedef makes the compiler splat ONE dword)
// AUX: cache-policy bits of the instruction (0 default; 2 = nt: streamed once, do not keep in L2)
template <int AUX = 0>
__device__ __forceinline__ void buf_store16(buf_rsrc_t rs, unsigned byte_off, f32x4 f) {
    decltype(__builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 0)) v;
    __builtin_memcpy(&v, &f, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, AUX);
}
__device__ __forceinline__ void buf_store4(buf_rsrc_t rs, unsigned byte_off, float f) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), rs, (int)byte_off, 0, 0);
}
}  // namespace hk
#define HK_BUF_RSRC 1
#endif

#ifndef HK_WAVE_SYNC
// Orders a wave's LDS writes before its own later LDS reads of other lanes' data (LDS operations of one wave execute in
// order: no instruction is needed, only the compiler must not move the accesses)
#define HK_WAVE_SYNC()                        \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_wave_barrier();      \
        asm volatile("" ::: "memory");        \
    } while (0)
#endif

#ifndef HK_FMAC_PINNED  // acc = fma(a, b, acc) as ONE v_fmac_f32 that stays where it is written: left to the compiler, a chain of
                        // side-product FMAs next to an MFMA stream is packed (v_pk_fma_f32) and sunk to the end of the
                        // loop body, which keeps every operand alive until there (hk_bwd3.h: +75 live registers, spills)
#define HK_FMAC_PINNED(acc, a, b) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
#endif

#ifndef HK_PIN_LOADED   // "this value is used here, unconditionally": placed behind a batch of loads whose results only feed selects,
                        // it keeps the loads ahead of the selects (the compiler otherwise sinks each load into its
                        // select's branch and waits for it there, one memory round trip per element)
#define HK_PIN_LOADED(v) asm volatile("" : "+v"(v))
#endif

#ifndef HK_LOAD16_ASYNC  // a 16-byte global load the compiler does NOT keep books on: `dst` counts as written at once, and it is the
                         // CALLER who guarantees - with a counted HK_VM_BARRIER between the request and the first use - that the
                         // data has arrived.  For register prefetches several pipeline steps ahead next to LDS-DMA and stores:
                         // with a tracked load the compiler's own s_waitcnt at the first use is vmcnt(0) as soon as stores are
                         // pending too (it assumes loads and stores may return out of order), which drains the whole pipeline
#define HK_LOAD16_ASYNC(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define HK_LOAD4_ASYNC(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#endif

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
