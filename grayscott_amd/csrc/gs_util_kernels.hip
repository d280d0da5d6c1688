// gs_util_kernels.hip -- plane utilities that are not arithmetic-flavour dependent.
//
// fill_rect backs Concentration::zeros / ones / fill_slice
// (/root/reference/data/src/concentration/mod.rs:205-243) directly in HBM: the seed
// rectangle of Species::new (:36-59) is written on the device, there is no host staging
// copy to upload (contrast ImageConcentration, data/src/concentration/gpu/image/mod.rs:86-131).
#include "gs_kernels.h"

namespace {
__global__ __launch_bounds__(256) void gs_fill_rect_k(float *row0, int pitch, int r0, int c0,
                                                      int c1, float value)
{
    const int r = r0 + blockIdx.y;
    const int c = c0 + blockIdx.x * 256 + threadIdx.x;
    if (c < c1) row0[(ptrdiff_t)r * pitch + c] = value;
}
} // namespace

hipError_t gs_launch_fill_rect(float *row0, int32_t pitch, int32_t r0, int32_t r1, int32_t c0,
                               int32_t c1, float value, hipStream_t s)
{
    if (r1 <= r0 || c1 <= c0) return hipSuccess;
    // gridDim.y is limited to 65535: walk tall rectangles in bands.
    for (int32_t b0 = r0; b0 < r1; b0 += 32768) {
        const int32_t b1 = (r1 - b0 > 32768) ? b0 + 32768 : r1;
        dim3 grid((unsigned)((c1 - c0 + 255) / 256), (unsigned)(b1 - b0));
        void *kargs[] = {&row0, &pitch, &b0, &c0, &c1, &value};
        hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(&gs_fill_rect_k), grid, dim3(256),
                                       kargs, 0, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
