// gs_util_kernels.hip -- plane utilities that are not arithmetic-flavour dependent.
//
// colormap is the per-pixel work of data-to-pics (V plane -> RGB8 through a palette).
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
// A plane's `cols` columns without its padding, row after row: the staging copy of gs_field_download_async.  (Not
// hipMemcpy2DAsync: a device-to-device copy may go to the same copy engines that carry the images to the host, and an
// image then leaves every 168 us where the link takes 152 -- tools/ubench/d2h_probe.hip, tools/call_pattern.py.)
template <typename T>
__global__ __launch_bounds__(256) void gs_pack_rows_k(const float *row0, int pitch, int rows, int cols_t, T *dst)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols_t) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        dst[(size_t)r * cols_t + c] = reinterpret_cast<const T *>(row0 + (ptrdiff_t)r * pitch)[c];
}
// Colour mapping of a result plane, data-to-pics/src/main.rs:139-144:
//     let color = ui::GRADIENT.eval_continuous((ui::AMPLITUDE_SCALE * value).into());
// i.e. an f32 multiply, widened to f64, handed to colorous 1.0.16 (Cargo.lock:389-392; the crate is not
// vendored).  Its sequential gradients are a port of d3-scale-chromatic's `ramp`: a table of n colours
// indexed with floor(t * n) clamped to [0, n - 1]; a NaN or negative t lands on entry 0 (saturating
// float -> usize cast).  The table itself is DATA handed in by the caller (for the reference: the 256
// entries of colorous::INFERNO), so only that indexing rule is restated here.
__global__ __launch_bounds__(256) void gs_colormap_k(const float *row0, int pitch, int rows, int cols, float scale,
                                                     const uint8_t *palette, int n, uint8_t *rgb)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= cols || r >= rows) return;
    const double t = (double)(scale * row0[(ptrdiff_t)r * pitch + c]);
    const double x = floor(t * (double)n);
    const int i = !(x >= 0.0) ? 0 : (x >= (double)n ? n - 1 : (int)x);
    uint8_t *px = rgb + ((size_t)r * cols + c) * 3;
    px[0] = palette[3 * i];
    px[1] = palette[3 * i + 1];
    px[2] = palette[3 * i + 2];
}
// The probe of gs_fields_place: one pass that reads two blocks and writes both back, word for word what it read (the
// XOR with a kernel argument that is zero at run time keeps the compiler from dropping the stores) -- the HBM traffic of
// one time step on a slot's two planes, and the blocks keep their contents.  Two 1 GiB blocks of ONE physical region
// ("group") take 0.86-0.96 ms, two of different groups 0.72-0.79 ms (tools/ubench/hbm_kinds.hip, profiles/r06_placement.md).
__global__ __launch_bounds__(256) void gs_pair_probe_k(uint4 *x, uint4 *y, size_t n, uint32_t zero)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 a = x[i], b = y[i];
        a.x ^= zero; a.y ^= zero; a.z ^= zero; a.w ^= zero;
        b.x ^= zero; b.y ^= zero; b.z ^= zero; b.w ^= zero;
        x[i] = a;
        y[i] = b;
    }
}
} // namespace

hipError_t gs_launch_pair_probe(void *x, void *y, size_t bytes, hipStream_t s)
{
    size_t n = bytes / sizeof(uint4);
    if (n == 0) return hipSuccess;
    uint4 *px = static_cast<uint4 *>(x), *py = static_cast<uint4 *>(y);
    uint32_t zero = 0;
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    void *kargs[] = {&px, &py, &n, &zero};
    return hipLaunchKernel(reinterpret_cast<const void *>(&gs_pair_probe_k), dim3(grid), dim3(256), kargs, 0, s);
}

hipError_t gs_launch_colormap(const float *row0, int32_t pitch, int32_t rows, int32_t cols, float scale,
                              const uint8_t *palette, int32_t n, uint8_t *rgb, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    for (int32_t b0 = 0; b0 < rows; b0 += 32768) { // gridDim.y is limited to 65535
        const int32_t nb = rows - b0 > 32768 ? 32768 : rows - b0;
        const float *src = row0 + (ptrdiff_t)b0 * pitch;
        uint8_t *dst = rgb + (size_t)b0 * cols * 3;
        void *kargs[] = {&src, &pitch, const_cast<int32_t *>(&nb), &cols, &scale, &palette, &n, &dst};
        hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(&gs_colormap_k), dim3((unsigned)((cols + 255) / 256), (unsigned)nb),
                                       dim3(256), kargs, 0, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t gs_launch_pack_rows(const float *row0, int32_t pitch, int32_t rows, int32_t cols, float *dst, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const unsigned gy = (unsigned)(rows < 32768 ? rows : 32768);
    // 16 bytes per lane where the rows allow it (the planes' pitch is a multiple of 64 floats, hipMalloc aligns to 256 B)
    if (cols % 4 == 0 && pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(row0) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
        const int ct = cols / 4;
        hipLaunchKernelGGL(gs_pack_rows_k<float4>, dim3((unsigned)((ct + 255) / 256), gy), dim3(256), 0, s, row0, pitch, rows, ct,
                           reinterpret_cast<float4 *>(dst));
    } else {
        hipLaunchKernelGGL(gs_pack_rows_k<float>, dim3((unsigned)((cols + 255) / 256), gy), dim3(256), 0, s, row0, pitch, rows, cols, dst);
    }
    return hipGetLastError();
}

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
