// Timing-only instances of the classifier-backward kernel (hawkeye_amd/csrc/hk_linear_bwd.h, LABV != 0): which side of
// linear_bwd64_kernel - the MFMA stream, the LDS-DMA reads, the stores, the fragment reads - the time of the whole is
// made of.  Results of these instances are WRONG by construction; nothing in the product links this file.
//   make -C tools/probe && python tools/linear_lab.py
#include "../../hawkeye_amd/csrc/hk_linear_bwd.h"

using namespace hk;

template <int LABV, int MODE>
static int launch_m(const float* g, const float* w, const float* y, float* dy, float* dw, float* db, int B, int J, int K, int walk,
                  hipStream_t st) {
    const int nchunk = J / 64;
    const int CPS = (nchunk + 255) / 256, S = (nchunk + CPS - 1) / CPS;
    const size_t ldsb = (size_t)4 * (25 + 8) * 1024;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bwd64_kernel<50, MODE, LABV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ldsb) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL((linear_bwd64_kernel<50, MODE, LABV>), dim3(S), dim3(512), ldsb, st, g, w, y, dy, dw, db, B, J, K, CPS, S, walk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int LABV>
static int launch(const float* g, const float* w, const float* y, float* dy, float* dw, float* db, int B, int J, int K, int walk,
                  hipStream_t st) {
    if (dy && dw) return launch_m<LABV, 0>(g, w, y, dy, dw, db, B, J, K, walk, st);
    if (dy) return launch_m<LABV, 1>(g, w, y, dy, dw, db, B, J, K, walk, st);
    return launch_m<LABV, 2>(g, w, y, dy, dw, db, B, J, K, walk, st);
}

// K must be 197..200 (50 class steps), B <= 64, J % 64 == 0
extern "C" int hk_probe_linear_bwd64(int labv, const float* g, const float* w, const float* y, float* dy, float* dw, float* db, int B,
                                     int J, int K, int walk, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (labv) {
        case 0: return launch<0>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 1: return launch<1>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 2: return launch<2>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 4: return launch<4>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 6: return launch<6>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 7: return launch<7>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 9: return launch<9>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 14: return launch<14>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 8: return launch<8>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 16: return launch<16>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 32: return launch<32>(g, w, y, dy, dw, db, B, J, K, walk, st);
        case 48: return launch<48>(g, w, y, dy, dw, db, B, J, K, walk, st);
        default: return -3;
    }
}

// ---- forward: linear_skinny_kernel<13, 4> at 64 samples x 200 classes (256 slabs of KS features)
#include "../../hawkeye_amd/csrc/hk_linear_fwd.h"

template <int LABV>
static int launch_fwd(const float* y, const float* w, float* part, int B, int J, int K, int walk, hipStream_t st) {
    const int S = 256, KS = ((J / 32 + S - 1) / S) * 32;
    const size_t ldsb = (size_t)4 * (64 * 32 + 13 * 16 * 32) * sizeof(float);
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_skinny_kernel<13, 4, LABV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL((linear_skinny_kernel<13, 4, LABV>), dim3(xcd_grid(S, 1)), dim3(512), ldsb, st, y, w, part, B, J, K, KS, S, 1, walk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// part: 256 * B * K floats
extern "C" int hk_probe_linear_fwd(int labv, const float* y, const float* w, float* part, int B, int J, int K, int walk, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (labv) {
        case 0: return launch_fwd<0>(y, w, part, B, J, K, walk, st);
        case 1: return launch_fwd<1>(y, w, part, B, J, K, walk, st);
        case 2: return launch_fwd<2>(y, w, part, B, J, K, walk, st);
        case 3: return launch_fwd<3>(y, w, part, B, J, K, walk, st);
        case 8: return launch_fwd<8>(y, w, part, B, J, K, walk, st);
        case 9: return launch_fwd<9>(y, w, part, B, J, K, walk, st);
        case 10: return launch_fwd<10>(y, w, part, B, J, K, walk, st);
        case 32: return launch_fwd<32>(y, w, part, B, J, K, walk, st);
        default: return -3;
    }
}
