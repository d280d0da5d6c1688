// Can a process reach another physical region ("group", profiles/r06_placement.md) of HBM without HOLDING tens of GiB?
// On a fresh box allocations walk linearly through physical memory and a group is tens of GiB long: twelve sequential
// 1 GiB draws stay inside it (bench line r06z: 4 + 12 blocks of one group, single_step 0.654).  A "spacer" -- one large
// allocation that is never touched and is freed a moment later -- moves the NEXT allocation that far ahead.
// This program: 4 reference blocks, then for spacers of 0, 4, 8, ... GiB: hipMalloc(spacer), two 1 GiB candidates,
// hipFree(spacer), in-place pair probes of the candidates against reference block 0 (same group: > 0.83 ms).
//   hipcc --offload-arch=gfx950 -O3 -o spacer_probe spacer_probe.hip && ./spacer_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

constexpr size_t GiB = 1ull << 30;
__global__ __launch_bounds__(256) void pair_k(uint4 *x, uint4 *y, size_t n, uint32_t zero)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 a = x[i], b = y[i];
        a.x ^= zero; b.x ^= zero;
        x[i] = a; y[i] = b;
    }
}
static hipEvent_t e0, e1;
static float pair_ms(void *x, void *y)
{
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        if (r) hipEventRecord(e0, 0);
        pair_k<<<4096, 256>>>((uint4 *)x, (uint4 *)y, GiB / 16, 0u);
        if (r) {
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            best = std::min(best, t);
        }
    }
    return best;
}
int main()
{
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<void *> ref(4);
    for (auto &p : ref) { if (hipMalloc(&p, GiB) != hipSuccess) return 1; hipMemset(p, 0, GiB); }
    std::printf("reference blocks: pairs with #0:");
    for (int i = 1; i < 4; ++i) std::printf(" %.3f", pair_ms(ref[0], ref[i]));
    std::printf(" ms (same group: > 0.83)\n");
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t s : {0, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 160, 200}) {
            void *spacer = nullptr, *c[2] = {nullptr, nullptr};
            const double t0 = now();
            if (s && hipMalloc(&spacer, s * GiB) != hipSuccess) { (void)hipGetLastError(); std::printf("spacer %3zu GiB: allocation failed\n", s); continue; }
            const double t1 = now();
            const bool ok = hipMalloc(&c[0], GiB) == hipSuccess && hipMalloc(&c[1], GiB) == hipSuccess;
            if (spacer) hipFree(spacer);
            const double t2 = now();
            if (!ok) { (void)hipGetLastError(); std::printf("spacer %3zu GiB: no candidates\n", s); if (c[0]) hipFree(c[0]); continue; }
            const float a = pair_ms(ref[0], c[0]), b = pair_ms(ref[0], c[1]), ab = pair_ms(c[0], c[1]);
            std::printf("spacer %3zu GiB (held %.1f ms, its hipMalloc %.1f ms): candidates against #0: %.3f %.3f ms -> %s %s; against each other %.3f\n", s,
                        (t2 - t0) * 1e3, (t1 - t0) * 1e3, a, b, a > 0.83f ? "same" : "OTHER", b > 0.83f ? "same" : "OTHER", ab);
            std::fflush(stdout);
            hipFree(c[0]);
            hipFree(c[1]);
        }
        std::printf("--- again (the allocator has seen the frees)\n");
    }
    return 0;
}
