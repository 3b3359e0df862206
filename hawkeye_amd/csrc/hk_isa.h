// gfx950 spellings the kernels use for what C++ has no word for: dynamic LDS, LDS-only barriers, explicit LDS address
// spaces, buffer-descriptor stores, pinned VALU instructions, loads the compiler keeps no books on.  Included by
// hk_common.h as <hk_isa.h>: the product build finds this file (-I. in the Makefile); the CPU emulation of the test tier
// (tests/emu) puts a directory with its own hk_isa.h earlier on the include path - nothing here is conditional.
#pragma once

// dynamic LDS of the launch as `float name[]` (HK_DYN_LDS16: declared 16-byte aligned)
#define HK_DYN_LDS(name) extern __shared__ float name[]
#define HK_DYN_LDS16(name) extern __shared__ __attribute__((aligned(16))) float name[]
// workgroup barrier that waits for LDS traffic only (global loads stay in flight across it)
#define HK_LDS_BARRIER()                                   \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)

// a volatile view of an LDS array: accesses stay in program order AND stay ds_read / ds_write.  A plain
// `volatile float*` is a generic pointer: flat_load / flat_store, which also count in vmcnt and
// wait behind every global load in flight (hk_cbp_fused.h: 3 us per tile)
#define HK_LDS_VOLATILE(p) ((volatile __attribute__((address_space(3))) float*)(p))

// a read-only view of an LDS location through an explicit LDS pointer (constant offsets fold into the
// instruction's offset field; through a generic pointer the compiler adds the - zero - LDS base
// with a VALU op per access)
#define HK_LDS_CONST(p) ((const __attribute__((address_space(3))) float*)(p))

// Plain (cached) stores through a buffer descriptor: the hardware drops the lanes whose byte offset lies beyond the
// descriptor's size - a ragged edge needs no predicate, so the instruction ALWAYS issues (a store under `if (row < n)` is
// skipped altogether when no lane passes: its place in a counted s_waitcnt vmcnt(n) would then be taken by an older load)
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

// Orders a wave's LDS writes before its own later LDS reads of other lanes' data (LDS operations of one wave execute in
// order: no instruction is needed, only the compiler must not move the accesses)
#define HK_WAVE_SYNC()                        \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_wave_barrier();      \
        asm volatile("" ::: "memory");        \
    } while (0)

// acc = fma(a, b, acc) as ONE v_fmac_f32 that stays where it is written: left to the compiler, a chain of
// side-product FMAs next to an MFMA stream is packed (v_pk_fma_f32) and sunk to the end of the
// loop body, which keeps every operand alive until there (hk_bwd3.h: +75 live registers, spills)
#define HK_FMAC_PINNED(acc, a, b) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))

// "this value is used here, unconditionally": placed behind a batch of loads whose results only feed selects,
// it keeps the loads ahead of the selects (the compiler otherwise sinks each load into its
// select's branch and waits for it there, one memory round trip per element)
#define HK_PIN_LOADED(v) asm volatile("" : "+v"(v))

// a 16-byte global load the compiler does NOT keep books on: `dst` counts as written at once, and it is the
// CALLER who guarantees - with a counted HK_VM_BARRIER between the request and the first use - that the
// data has arrived.  For register prefetches several pipeline steps ahead next to LDS-DMA and stores:
// with a tracked load the compiler's own s_waitcnt at the first use is vmcnt(0) as soon as stores are
// pending too (it assumes loads and stores may return out of order), which drains the whole pipeline
#define HK_LOAD16_ASYNC(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define HK_LOAD4_ASYNC(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
