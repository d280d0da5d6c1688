// How fast can an 8.3 MB image (the V plane of the reference's default 1080 x 1920) leave the device?  The floor of the
// reference's 32-step calls with an image each is one image per call (tools/call_pattern.py: 168 us with two in flight =
// 49 GB/s = 394 k Mcells x steps / s at 32 steps).  Candidates: hipMemcpyAsync into page-locked memory (what
// gs_field_download_async does, after a device-side staging copy), the same with the copy engines switched off
// (HSA_ENABLE_SDMA=0: blit kernels), and a kernel that stores straight into mapped page-locked memory -- by how many
// workgroups, with plain or nontemporal stores -- one, two and three images in flight.
//
//   hipcc --offload-arch=gfx950 -O3 -o d2h_probe d2h_probe.hip ;  ./d2h_probe [rows=1080] [cols=1920]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            std::printf("FAILED %s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            std::fflush(stdout);                                                                      \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

template <int NT>
__global__ __launch_bounds__(256) void push_k(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = reinterpret_cast<const v4f *>(src)[i];
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst) + i);
        else reinterpret_cast<v4f *>(dst)[i] = v;
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const size_t rows = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1080, cols = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 1920;
    const size_t bytes = rows * cols * sizeof(float), n4 = bytes / 16;
    float *dev = nullptr;
    CK(hipMalloc(reinterpret_cast<void **>(&dev), bytes));
    std::vector<float> init(rows * cols);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (float)(i % 1000) * 0.001f;
    CK(hipMemcpy(dev, init.data(), bytes, hipMemcpyHostToDevice));
    constexpr int kBuf = 3;
    float *host[kBuf], *host_dev[kBuf];
    hipStream_t st[kBuf];
    for (int i = 0; i < kBuf; ++i) {
        CK(hipHostMalloc(reinterpret_cast<void **>(&host[i]), bytes, hipHostMallocMapped));
        CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&host_dev[i]), host[i], 0));
        std::memset(host[i], 0, bytes);
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    std::printf("%zu x %zu f32 = %.2f MB per image; HSA_ENABLE_SDMA=%s\n", rows, cols, bytes / 1e6, std::getenv("HSA_ENABLE_SDMA") ? std::getenv("HSA_ENABLE_SDMA") : "(unset)");
    const int reps = 200;
    auto run = [&](const char *what, int in_flight, auto &&enqueue) -> int {
        double best = 1e30;
        for (int round = 0; round < 4; ++round) {
            for (int i = 0; i < kBuf; ++i) CK(hipStreamSynchronize(st[i]));
            const double t0 = now_us();
            for (int i = 0; i < reps; ++i) {
                const int b = i % in_flight;
                CK(hipStreamSynchronize(st[b])); // the image before last on this buffer has arrived
                if (enqueue(b)) return 1;
            }
            for (int i = 0; i < kBuf; ++i) CK(hipStreamSynchronize(st[i]));
            best = std::min(best, (now_us() - t0) / reps);
        }
        std::printf("  %-58s %d in flight: %7.1f us per image = %5.1f GB/s -> %6.0f k Mcells x steps / s at 32 steps\n", what, in_flight, best,
                    bytes / best / 1e3, 32.0 * rows * cols / best / 1e3);
        std::fflush(stdout);
        return 0;
    };
    for (int f = 1; f <= kBuf; ++f)
        if (run("hipMemcpyAsync device -> page-locked", f, [&](int b) -> int { CK(hipMemcpyAsync(host[b], dev, bytes, hipMemcpyDeviceToHost, st[b])); return 0; })) return 1;
    for (int grid : {16, 32, 64, 128, 256, 512})
        for (int nt = 0; nt < 2; ++nt)
            for (int f = 1; f <= kBuf; f += 1) {
                char what[96];
                std::snprintf(what, sizeof what, "kernel stores to mapped memory, %3d workgroups, %s", grid, nt ? "nontemporal" : "plain");
                if (run(what, f, [&](int b) -> int {
                        if (nt) hipLaunchKernelGGL(push_k<1>, dim3(grid), dim3(256), 0, st[b], reinterpret_cast<const float4 *>(dev), reinterpret_cast<float4 *>(host_dev[b]), n4);
                        else hipLaunchKernelGGL(push_k<0>, dim3(grid), dim3(256), 0, st[b], reinterpret_cast<const float4 *>(dev), reinterpret_cast<float4 *>(host_dev[b]), n4);
                        CK(hipGetLastError());
                        return 0;
                    }))
                    return 1;
            }
    // did the bytes arrive?
    bool ok = true;
    for (int i = 0; i < kBuf; ++i) ok = ok && std::memcmp(host[i], init.data(), bytes) == 0;
    std::printf("contents %s\n", ok ? "identical in all buffers" : "DIFFER");
    return ok ? 0 : 2;
}
