// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the step
// kernels use (MI355X_MICROARCH.md, "HBM": only 16 B/lane streaming is calibrated there).
// Each kernel streams a 1 GiB buffer into another, one element of the given width per lane and
// iteration; the byte counts are known, the counters come from
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib      (and a second run: WRITE_SIZE)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void copy_k(const T *__restrict__ in, T *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

int main()
{
    const size_t bytes = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        copy_k<float4><<<8192, 256>>>((const float4 *)a, (float4 *)b, bytes / 16);
        copy_k<float2><<<8192, 256>>>((const float2 *)a, (float2 *)b, bytes / 8);
        copy_k<float><<<8192, 256>>>((const float *)a, (float *)b, bytes / 4);
    }
    (void)hipDeviceSynchronize();
    std::printf("copied %zu bytes per launch, widths 16 / 8 / 4 B per lane\n", bytes);
    return 0;
}
