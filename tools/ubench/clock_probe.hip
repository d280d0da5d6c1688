// Micro-benchmark: the shader clock a kernel actually runs at, by launch length and launch pattern.
// Every wave reads s_memtime (shader-clock counter) and s_memrealtime (constant 100 MHz) around a VALU loop
// of the step kernel's tap mix; the host prints shader cycles per microsecond (= MHz) and VALU instructions
// per cycle and SIMD for (a) one long launch, (b) trains of short dependent launches of 256 / 160 / 48
// workgroups of 16 waves -- the shape of the LDS-window kernel on mid-size grids.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *t, int iters, float seed)
{
    float a[8], x[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.f; x[i] = seed + i + threadIdx.x; }
    const float b = seed * 0.5f, c = seed * 0.25f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) // 5 VALU per tap pair: sub, mul, add, sub div:2, add
            asm volatile("v_sub_f32 %1, %2, %3\n\tv_mul_f32 %1, %4, %1\n\tv_add_f32 %0, %0, %1\n\t"
                         "v_sub_f32_e64 %1, %3, %2 div:2\n\tv_add_f32 %0, %0, %1"
                         : "+v"(a[i]), "+v"(x[i]) : "v"(b), "v"(c), "v"(c));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
        t[2 * w] = c1 - c0;
        t[2 * w + 1] = r1 - r0;
    }
}

int main()
{
    float *out;
    unsigned long long *t;
    const int maxw = 1024 * 16;
    hipMalloc(&out, 4096);
    hipMalloc(&t, maxw * 16);
    std::vector<unsigned long long> h(2 * maxw);
    struct Case { const char *name; int blocks, iters, launches; } cases[] = {
        {"one long launch, 256 WGs", 256, 20000, 1},
        {"one long launch, 512 WGs", 512, 20000, 1},
        {"train of short launches, 256 WGs", 256, 12, 400},
        {"train of short launches, 160 WGs", 160, 12, 400},
        {"train of short launches, 48 WGs", 48, 12, 400},
        {"train of short launches, 512 WGs", 512, 12, 400},
        {"train of medium launches, 256 WGs", 256, 200, 200},
    };
    for (auto &cs : cases) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<cs.blocks, 1024>>>(out, t, cs.iters, 1.0f); // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < cs.launches; ++l) probe<<<cs.blocks, 1024>>>(out, t, cs.iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), t, cs.blocks * 16 * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz, cyc, wall;
        for (int w = 0; w < cs.blocks * 16; ++w) {
            if (h[2 * w + 1] == 0) continue;
            mhz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] / 100.0));
            cyc.push_back((double)h[2 * w]);
            wall.push_back((double)h[2 * w + 1] * 10.0); // ns
        }
        std::sort(mhz.begin(), mhz.end());
        std::sort(cyc.begin(), cyc.end());
        std::sort(wall.begin(), wall.end());
        const double insts = 40.0 * cs.iters; // per wave
        const double waves_per_simd = 4.0 * ((cs.blocks + 255) / 256 > 1 ? 2 : 1);
        std::printf("%-40s %8.2f us per launch | shader clock median %6.0f MHz (min %6.0f, max %6.0f) | wave loop %8.0f cycles"
                    " = %.2f counter ticks per VALU instruction and SIMD at %g waves per SIMD | wave loop %8.0f ns = %.3f ns per instruction and SIMD\n",
                    cs.name, ms * 1000.0 / cs.launches, mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc[cyc.size() / 2],
                    cyc[cyc.size() / 2] / (insts * waves_per_simd), waves_per_simd, wall[wall.size() / 2],
                    wall[wall.size() / 2] / (insts * waves_per_simd));
    }
    return 0;
}
