// Last stage of the input pipeline on the device: the data workers ship uint8 HWC crops (a quarter of
// the host->device bytes of float images, and no float math on the CPU); this kernel does what the reference's
// preset does after its PIL stages - PILToTensor + ConvertImageDtype(float) + Normalize + RandomErasing(value 0),
// dataset/transforms.py:38-46 - in one pass, straight into the layout the backbone runs in (NCHW or channels_last).
// Same fp32 operations in the same order (u / 255, - mean, / std: IEEE divisions), so it is bit-identical to the
// CPU path.  HBM-bound: 3 bytes read, 12 written per pixel.
#include "hk_common.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

struct Norm3 {
    float m[3], s[3];
};

__global__ __launch_bounds__(256) void image_finalize_kernel(const uint8_t* __restrict__ u8, Norm3 nm,
                                                            const int32_t* __restrict__ erase, float* __restrict__ out,
                                                            int H, int W, int channels_last) {
    const int b = blockIdx.y;
    const long long hw = (long long)H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / W), x = (int)(p % W);
    bool erased = false;
    if (erase) {
        const int top = erase[b * 4 + 0], left = erase[b * 4 + 1], eh = erase[b * 4 + 2], ew = erase[b * 4 + 3];
        erased = eh > 0 && ew > 0 && y >= top && y < top + eh && x >= left && x < left + ew;
    }
    const uint8_t* src = u8 + ((long long)b * hw + p) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = erased ? 0.f : ((float)src[c] / 255.0f - nm.m[c]) / nm.s[c];
    if (channels_last) {
        float* dst = out + ((long long)b * hw + p) * 3;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
    } else {
        float* dst = out + (long long)b * 3 * hw + p;
        dst[0] = v[0]; dst[hw] = v[1]; dst[2 * hw] = v[2];
    }
}

}  // namespace hk

using namespace hk;

extern "C" int hk_image_finalize(const uint8_t* u8, const float* mean3, const float* std3, const int32_t* erase, float* out,
                                 int B, int H, int W, int channels_last, hk_stream_t stream) {
    if (!u8 || !mean3 || !std3 || !out || B <= 0 || H <= 0 || W <= 0) return HK_ERR_BAD_ARG;
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        nm.m[c] = mean3[c];            // host pointers: three floats each
        nm.s[c] = std3[c];
        if (nm.s[c] == 0.f) return HK_ERR_BAD_ARG;
    }
    const long long hw = (long long)H * W;
    hipLaunchKernelGGL(image_finalize_kernel, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, u8, nm,
                       erase, out, H, W, channels_last);
    HK_LAUNCH_CHECK();
    return HK_OK;
}
