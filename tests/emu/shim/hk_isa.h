// CPU-emulation counterpart of hawkeye_amd/csrc/hk_isa.h (TEST INFRASTRUCTURE: tests/emu/build_emu.py puts this directory
// ahead of the kernel sources on the include path, so hk_common.h's `#include <hk_isa.h>` finds this file; the product
// build never sees it).  Same names, host meanings: dynamic LDS from the emulated block, plain fma, synchronous loads,
// bounds-checked buffer stores, fiber barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#define HK_DYN_LDS(name) float* name = reinterpret_cast<float*>(hipemu::B->dyn_lds)
#define HK_DYN_LDS16(name) HK_DYN_LDS(name)
#define HK_FMAC_PINNED(acc, a, b) ((acc) = fmaf((a), (b), (acc)))
#define HK_PIN_LOADED(v) ((void)0)
#define HK_LOAD16_ASYNC(dst, ptr) ((dst) = *reinterpret_cast<const hipemu::v4f*>(ptr))
#define HK_LOAD4_ASYNC(dst, ptr) ((dst) = *(ptr))
// buffer-descriptor stores: bounds-checked like the hardware (lanes beyond the descriptor's size are dropped)
namespace hk {
struct buf_rsrc_t { char* p; long long bytes; };
inline buf_rsrc_t buf_rsrc(const float* base, long long floats) { return buf_rsrc_t{(char*)base, floats * 4}; }
template <int AUX = 0>
inline void buf_store16(buf_rsrc_t rs, unsigned off, hipemu::v4f f) { if ((long long)off + 16 <= rs.bytes) memcpy(rs.p + off, &f, 16); }
inline void buf_store4(buf_rsrc_t rs, unsigned off, float f) { if ((long long)off + 4 <= rs.bytes) memcpy(rs.p + off, &f, 4); }
}
#define HK_WAVE_SYNC() hipemu::wave_barrier()   /* the fibers of a wave are not in lockstep between collectives */
#define HK_LDS_VOLATILE(p) ((volatile float*)(p))
#define HK_LDS_CONST(p) ((const float*)(p))
#define HK_LDS_BARRIER() hipemu::block_barrier()
