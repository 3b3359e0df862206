// CPU emulation of the slice of the HIP device model that hawkeye_amd/csrc uses.
//
// TEST INFRASTRUCTURE ONLY.  tests/emu/build_emu.py compiles the *unchanged* kernel sources (hawkeye_amd/csrc/*.hip)
// for x86 against this header into tests/emu/_build/libhawkeye_emu.so, so that the CPU test tier can run the real
// kernel code (index arithmetic, LDS staging, barriers, MFMA operand layouts, reduction orders) against the oracle
// without a GPU.  Nothing under hawkeye_amd/ loads it: the product library is libhawkeye_hip.so (gfx950) and the
// product path raises without a GPU.  It says nothing about speed.
//
// Model: one workgroup at a time; its work-items are fibers (own stacks, hipemu_switch.S) run round-robin on one OS thread, switching
// only at __syncthreads()/s_barrier and at wave-wide collectives (__shfl_xor, MFMA), which is where real wave64
// hardware synchronises too.  A barrier that not every live work-item reaches aborts with a message instead of
// hanging.  Dynamic and static LDS are plain host memory (dynamic LDS is poisoned with NaNs at every block start).
//
// MFMA operand/result layouts (CDNA3/4 ISA guide, "Matrix Arithmetic Instructions"):
//   v_mfma_f32_32x32x2_f32 : lane l supplies A[i=l%32][k=l/32], B[k=l/32][j=l%32];
//                            result reg v (0..15) of lane l is C[i = 8*(v/4) + 4*(l/32) + v%4][j = l%32]
//   v_mfma_f32_16x16x4_f32 : lane l supplies A[i=l%16][k=l/16], B[k=l/16][j=l%16];
//                            result reg v (0..3) of lane l is C[i = 4*(l/16) + v][j = l%16]
// The GPU-validated kernels of round 1 are the emulator's own test: they only reproduce the oracle if these hold.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int min(int a, int b) { return a < b ? a : b; }       // HIP's device-side integer overloads
inline int max(int a, int b) { return a > b ? a : b; }
typedef int hipError_t;
enum { hipSuccess = 0 };
struct hipemu_stream_tag;
typedef hipemu_stream_tag* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// helper queues / events (the Newton-Schulz two-queue dispatch): everything runs in launch order on the CPU
struct hipemu_event_tag;
typedef hipemu_event_tag* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamGetDevice(hipStream_t, int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(sizeof(void*)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(sizeof(void*)); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
// instruction-scheduling fences: no meaning off the GPU
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_readfirstlane(v) (v)      /* only ever applied to wave-uniform values */

extern "C" void hipemu_switch(void** save_sp, void* load_sp);   // tests/emu/shim/hipemu_switch.S
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

namespace hipemu {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
constexpr size_t STACK_BYTES = 512 * 1024;

struct Wave {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    uint64_t slot[2][64][4];   // exchange slots (32 bytes per lane) of the two most recent collectives
};
struct Fiber {
    void* sp = nullptr;          // saved stack pointer while switched out
    int state = READY;
    unsigned wait_gen = 0;
    dim3 tid;
    int lin = 0, wave = 0, lane = 0;
    unsigned parity = 0;
    char* stack = nullptr;
};
struct Block {
    dim3 grid, block, bid;
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    std::vector<Wave> waves;
    char* dyn_lds = nullptr;
};

inline Block* B = nullptr;
inline Fiber* cur = nullptr;
inline void* sched_sp = nullptr;
inline const std::function<void()>* body = nullptr;
inline std::vector<Fiber> pool;

inline void release_block() {
    B->arrived = 0;
    B->gen++;
}
inline void release_wave(Wave& w) {
    w.arrived = 0;
    w.gen++;
}
inline void block_barrier() {
    const unsigned g = B->gen;
    if (++B->arrived == B->alive) {
        release_block();
        return;
    }
    cur->state = WAIT_BLOCK;
    cur->wait_gen = g;
    hipemu_switch(&cur->sp, sched_sp);
}
inline void wave_barrier() {
    Wave& w = B->waves[cur->wave];
    const unsigned g = w.gen;
    if (++w.arrived == w.alive) {
        release_wave(w);
        return;
    }
    cur->state = WAIT_WAVE;
    cur->wait_gen = g;
    hipemu_switch(&cur->sp, sched_sp);
}
inline void trampoline() {
    (*body)();
    Fiber* f = cur;
    f->state = DONE;
    Wave& w = B->waves[f->wave];
    if (--B->alive > 0 && B->arrived == B->alive) release_block();     // exited work-items do not take part in barriers
    if (--w.alive > 0 && w.arrived == w.alive) release_wave(w);
    hipemu_switch(&f->sp, sched_sp);
    abort();                                   // a finished fiber is never resumed
}

inline void run_block(const std::function<void()>& fn, dim3 grid, dim3 block, dim3 bid, size_t lds_bytes) {
    const int n = (int)(block.x * block.y * block.z);
    Block blk;
    blk.grid = grid;
    blk.block = block;
    blk.bid = bid;
    blk.alive = n;
    blk.waves.resize((n + 63) / 64);
    std::vector<uint32_t> lds((lds_bytes + 3) / 4 + 16, 0x7fc00000u);   // NaN poison
    blk.dyn_lds = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
#ifdef HIPEMU_ASAN
    {   // everything of the buffer past the requested size is a red zone
        char* end = blk.dyn_lds + ((lds_bytes + 7) & ~size_t(7));
        char* cap = reinterpret_cast<char*>(lds.data() + lds.size());
        if (cap > end) __asan_poison_memory_region(end, (size_t)(cap - end));
    }
#endif
    if ((int)pool.size() < n) pool.resize(n);
    B = &blk;
    body = &fn;
    for (int t = 0; t < n; ++t) {
        Fiber& f = pool[t];
        if (!f.stack) f.stack = static_cast<char*>(malloc(STACK_BYTES));
        f.state = READY;
        f.parity = 0;
        f.lin = t;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.wave = t / 64;
        f.lane = t % 64;
        blk.waves[f.wave].alive++;
#ifdef HIPEMU_ASAN
        __asan_unpoison_memory_region(f.stack, STACK_BYTES);        // stale redzones of the fiber that ran here before
#endif
        // initial frame: six callee-saved registers, then the address hipemu_switch "returns" to; the slot above it
        // stands for trampoline's own return address, so that rsp = 16 n + 8 at its entry as the ABI requires
        void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~uintptr_t(15));
        *--top = nullptr;
        *--top = reinterpret_cast<void*>(&trampoline);
        for (int r = 0; r < 6; ++r) *--top = nullptr;
        f.sp = top;
    }
    // Order in which runnable work-items are resumed.  Results must not depend on it: a kernel that gives different
    // answers under HK_EMU_ORDER=rev / rand:<seed> is missing a barrier (the emulator only switches work-items at
    // barriers and wave collectives, so a race is invisible in any single fixed order).
    std::vector<int> order(n);
    for (int t = 0; t < n; ++t) order[t] = t;
    if (const char* o = getenv("HK_EMU_ORDER")) {
        if (!strncmp(o, "rev", 3)) {
            for (int t = 0; t < n; ++t) order[t] = n - 1 - t;
        } else if (!strncmp(o, "rand", 4)) {
            uint64_t st = 0x9e3779b97f4a7c15ull ^ (o[4] == ':' ? strtoull(o + 5, nullptr, 10) : 0) ^
                          (uint64_t(bid.x) * 1315423911u + bid.y * 2654435761u + bid.z);
            for (int t = n - 1; t > 0; --t) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const int j = (int)((st >> 33) % (uint64_t)(t + 1));
                const int tmp = order[t];
                order[t] = order[j];
                order[j] = tmp;
            }
        }
    }
    int remaining = n;
    while (remaining) {
        bool progressed = false;
        for (int q = 0; q < n; ++q) {
            Fiber& f = pool[order[q]];
            if (f.state == DONE) continue;
            if (f.state == WAIT_BLOCK && blk.gen == f.wait_gen) continue;
            if (f.state == WAIT_WAVE && blk.waves[f.wave].gen == f.wait_gen) continue;
            f.state = READY;
            cur = &f;
            hipemu_switch(&sched_sp, f.sp);
            progressed = true;
            if (f.state == DONE) --remaining;
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): a barrier or wave collective was not reached by every "
                            "live work-item\n", bid.x, bid.y, bid.z);
            abort();
        }
    }
#ifdef HIPEMU_ASAN
    __asan_unpoison_memory_region(lds.data(), lds.size() * sizeof(uint32_t));
#endif
    B = nullptr;
    cur = nullptr;
}

inline void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& fn) {
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) run_block(fn, grid, block, dim3(x, y, z), lds_bytes);
}

// value of `v` held by lane `src(lane)` of the calling wave (own value if that lane does not exist)
template <class T, class SrcFn>
inline T wave_exchange(T v, SrcFn src) {
    static_assert(sizeof(T) <= 8, "wave_exchange: 4- or 8-byte types");
    Wave& w = B->waves[cur->wave];
    const unsigned p = cur->parity;
    cur->parity ^= 1u;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.slot[p][cur->lane][0] = raw;
    wave_barrier();
    const int s = src(cur->lane);
    const int width = (int)(B->block.x * B->block.y * B->block.z) - cur->wave * 64;
    T out = v;
    if (s >= 0 && s < 64 && s < width) memcpy(&out, &w.slot[p][s][0], sizeof(T));
    return out;
}

inline v16f mfma_f32_32x32x2(float a, float b, v16f c, int, int, int) {
    Wave& w = B->waves[cur->wave];
    const unsigned p = cur->parity;
    cur->parity ^= 1u;
    const int l = cur->lane;
    float ab[2] = {a, b};
    memcpy(&w.slot[p][l][0], ab, 8);
    wave_barrier();
    const int j = l % 32, hb = l / 32;
    for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * hb + v % 4;
        float acc = c[v];
        for (int k = 0; k < 2; ++k) {
            float ai[2], bj[2];
            memcpy(ai, &w.slot[p][32 * k + i][0], 8);
            memcpy(bj, &w.slot[p][32 * k + j][0], 8);
            acc = fmaf(ai[0], bj[1], acc);
        }
        c[v] = acc;
    }
    return c;
}

inline v4f mfma_f32_16x16x4(float a, float b, v4f c, int, int, int) {
    Wave& w = B->waves[cur->wave];
    const unsigned p = cur->parity;
    cur->parity ^= 1u;
    const int l = cur->lane;
    float ab[2] = {a, b};
    memcpy(&w.slot[p][l][0], ab, 8);
    wave_barrier();
    const int j = l % 16, hb = l / 16;
    for (int v = 0; v < 4; ++v) {
        const int i = 4 * hb + v;
        float acc = c[v];
        for (int k = 0; k < 4; ++k) {
            float ai[2], bj[2];
            memcpy(ai, &w.slot[p][16 * k + i][0], 8);
            memcpy(bj, &w.slot[p][16 * k + j][0], 8);
            acc = fmaf(ai[0], bj[1], acc);
        }
        c[v] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_bf16 (gfx950): lane l supplies eight consecutive k of row / column l % 32, k = 8 (l / 32) + 0..7;
// C/D as the other 32x32 forms.  (The A/B layout is this model's assumption - see hk_bgemm.h - until confirmed on
// the device.)  Piece products are exact in fp32; they are accumulated as an fp32 fma chain in k order.
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
inline v16f mfma_f32_32x32x16_bf16(v8bf a, v8bf b, v16f c, int, int, int) {
    Wave& w = B->waves[cur->wave];
    const unsigned p = cur->parity;
    cur->parity ^= 1u;
    const int l = cur->lane;
    memcpy(&w.slot[p][l][0], &a, 16);
    memcpy(&w.slot[p][l][2], &b, 16);
    wave_barrier();
    const int j = l % 32, hb = l / 32;
    for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * hb + v % 4;
        float acc = c[v];
        for (int kb = 0; kb < 2; ++kb) {
            uint16_t ai[8], bj[8];
            memcpy(ai, &w.slot[p][32 * kb + i][0], 16);
            memcpy(bj, &w.slot[p][32 * kb + j][2], 16);
            for (int t = 0; t < 8; ++t) {
                uint32_t ua = (uint32_t)ai[t] << 16, ub = (uint32_t)bj[t] << 16;
                float fa, fb;
                memcpy(&fa, &ua, 4);
                memcpy(&fb, &ub, 4);
                acc = fmaf(fa, fb, acc);
            }
        }
        c[v] = acc;
    }
    return c;
}

}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::B->bid)
#define blockDim (hipemu::B->block)
#define gridDim (hipemu::B->grid)

#define __syncthreads() hipemu::block_barrier()
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)   /* memory operations of a fiber complete at once */
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu::mfma_f32_32x32x2
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu::mfma_f32_16x16x4
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu::mfma_f32_32x32x16_bf16

// LDS-DMA (global_load_lds_dwordx4 ...): lane l of the wave copies `size` bytes from ITS global address to the
// wave-uniform LDS base + size * l.  Executed at once by each fiber (the real copy is asynchronous and is waited for by
// the barrier; the kernels only ever target an LDS stage nobody reads before the next barrier).
namespace hipemu {
inline void global_load_lds(const void* g, void* l, unsigned size, int off) {
    memcpy((char*)l + off + (size_t)size * (cur->tid.x & 63), (const char*)g + off, size);
}
}  // namespace hipemu
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipemu::global_load_lds((const void*)(g), (void*)(l), size, off)

// LDS / global integer atomics: the fibers of a workgroup are switched only at barriers and wave-wide collectives, so a
// read-modify-write between two switch points is atomic by construction (and min / max do not depend on the order)
inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }

template <class T>
inline T __shfl_xor(T v, int mask, int = 64) {
    return hipemu::wave_exchange(v, [mask](int lane) { return lane ^ mask; });
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int = 64) {
    return hipemu::wave_exchange(v, [delta](int lane) { return lane + (int)delta; });
}
template <class T>
inline T __shfl(T v, int srclane, int = 64) {
    return hipemu::wave_exchange(v, [srclane](int) { return srclane; });
}

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                         \
    do {                                                                                                  \
        (void)(stream);                                                                                   \
        hipemu::launch(dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); });           \
    } while (0)
