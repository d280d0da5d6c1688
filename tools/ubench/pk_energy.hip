// Packed f32 math against plain f32 math IN JOULES ("next round" item 4a).  Round 1 measured that
// v_pk_add_f32 / v_pk_mul_f32 have the lane throughput of the plain instructions on gfx950 (two passes per wave64) and
// dismissed them; round 5 found that the marching kernel sits on the board's power cap, where the quantity that matters is
// energy per cell-step, not issue slots -- and never asked what a packed operation costs in energy.  The accumulations
// of the update come in (U, V) pairs (acc_u += t_u; acc_v += t_v), so half of the kernel's adds could be packed.
//
// Every mode runs the same number of lane-operations on non-trivial operands at the marching kernel's occupancy (4 waves
// per SIMD) for about `seconds`; the card's accumulated-energy counter (rocm-smi) is read before and after inside this
// process, so that process start-up is not in the window.  Reported: lane-ops/s, watts, pJ per lane-op.
//   modes: 0 idle loop (s_nop: the floor)   1 v_add_f32 + v_mul_f32   2 v_pk_add_f32 + v_pk_mul_f32
//          3 the tap mix of the update, plain      4 the same with the accumulating adds packed over (U, V)
//   hipcc --offload-arch=gfx950 -O3 -o pk_energy pk_energy.hip && ./pk_energy [seconds=4]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[16];
    v2f p[8];
    const float lanev = (float)(threadIdx.x * 2654435761u >> 8) * (1.0f / 16777216.0f); // "random" mantissas per lane
    for (int i = 0; i < 16; ++i) a[i] = seed + lanev * (float)(i + 1);
    for (int i = 0; i < 8; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    float b = 1.0f + lanev * 1e-6f, c = lanev * 1e-3f;           // x <- x * (1 + eps) + small: values stay finite, bits keep toggling
    v2f b2 = {b, b}, c2 = {c, -c};
    float t0 = 0.0f, t1 = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
        } else if (MODE == 1) { // 16 x (add, mul) = 32 lane-ops per lane
#define P1(i) asm volatile("v_add_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %1, %1, %3" : "+v"(a[2 * i]), "+v"(a[2 * i + 1]) : "v"(c), "v"(b));
            REP8(P1)
        } else if (MODE == 2) { // 8 x (pk_add, pk_mul) = 32 lane-ops per lane
#define P2(i) asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %2" : "+v"(p[i]) : "v"(c2), "v"(b2));
            REP8(P2)
        } else if (MODE == 3) {
            // per pair of species: corner tap (sub, mul, add) and side tap (sub div:2, add) for U and for V = 10 instructions,
            // 10 lane-ops per lane
#define P3(i) asm volatile("v_sub_f32 %2, %0, %4\n\tv_sub_f32 %3, %1, %4\n\tv_mul_f32 %2, %5, %2\n\tv_mul_f32 %3, %5, %3\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t" \
                           "v_sub_f32_e64 %2, %0, %4 div:2\n\tv_sub_f32_e64 %3, %1, %4 div:2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" \
                           : "+v"(a[2 * i]), "+v"(a[2 * i + 1]), "+v"(t0), "+v"(t1) : "v"(c), "v"(b));
            REP8(P3)
        }
    }
    float s = t0 + t1;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// MODE 4 written with the compiler's register-pair handling instead of %L / %H: components through .x / .y operands
__global__ __launch_bounds__(256) void k_pkmix(float *out, int iters, float seed)
{
    v2f p[8];
    const float lanev = (float)(threadIdx.x * 2654435761u >> 8) * (1.0f / 16777216.0f);
    for (int i = 0; i < 8; ++i) p[i] = v2f{seed + lanev * (float)(2 * i + 1), seed + lanev * (float)(2 * i + 2)};
    const float b = 1.0f + lanev * 1e-6f, c = lanev * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v2f t;
            // corner tap: two subs, two muls, ONE packed add
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t.x) : "v"(p[i].x), "v"(c));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t.y) : "v"(p[i].y), "v"(c));
            asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t.x) : "v"(b));
            asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t.y) : "v"(b));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));
            // side tap: two subs with div:2, ONE packed add
            asm volatile("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(t.x) : "v"(p[i].x), "v"(c));
            asm volatile("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(t.y) : "v"(p[i].y), "v"(c));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static double energy_uj()
{
    FILE *f = popen("rocm-smi -d 0 --showenergycounter 2>/dev/null", "r");
    if (!f) return -1.0;
    char line[512];
    double v = -1.0;
    while (fgets(line, sizeof line, f)) {
        const char *p = std::strstr(line, "Accumulated Energy (uJ):");
        if (p) v = std::atof(p + 24);
    }
    pclose(f);
    return v;
}

template <typename K>
static void run(const char *name, K kernel, double lane_ops_per_lane_iter, double seconds, float *d)
{
    const dim3 grid(256 * 4 * 8), block(256); // 4 blocks of 4 waves per CU at a time (one wave per SIMD each), 8 rounds
    // calibrate iterations for ~40 ms per launch
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, grid, block, 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, grid, block, 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 40.0 / ms);
    // rate: a few launches between events
    hipEventRecord(e0);
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double rate = (double)grid.x * 256.0 * (double)iters * lane_ops_per_lane_iter * 8.0 / (ms * 1e-3);
    // power: the slope of the card's energy counter between two readings taken WHILE a queue of launches that outlasts
    // both keeps the chip under the same load (each reading is stamped with the middle of its rocm-smi call)
    const int launches = (int)((seconds + 2.5) * 1000.0 / 40.0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, d, iters, 1.0f);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto reading = [&](double *t, double *uj) { const double a = now(); *uj = energy_uj(); *t = 0.5 * (a + now()); };
    std::this_thread::sleep_for(std::chrono::milliseconds(600));
    double ta, ea, tb, eb;
    reading(&ta, &ea);
    std::this_thread::sleep_for(std::chrono::milliseconds((int)(seconds * 1000)));
    reading(&tb, &eb);
    const bool busy = hipStreamQuery(0) == hipErrorNotReady; // the queue is still running
    hipDeviceSynchronize();
    const double watts = (eb - ea) * 1e-6 / (tb - ta);
    std::printf("%-62s %8.2f T lane-ops/s  %7.1f W over %.2f s%s  %6.2f pJ per lane-op\n", name, rate / 1e12, watts, tb - ta, busy ? "" : " (QUEUE DRAINED EARLY)",
                watts / rate * 1e12);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 4.0;
    float *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    if (energy_uj() < 0) { std::printf("no energy counter (rocm-smi)\n"); return 2; }
    run("idle loop (s_nop)", k<0>, 1.0, seconds, d);
    run("v_add_f32 + v_mul_f32 (32 instructions)", k<1>, 32.0, seconds, d);
    run("v_pk_add_f32 + v_pk_mul_f32 (16 instructions, same lane-ops)", k<2>, 32.0, seconds, d);
    run("tap mix, plain: 10 instructions per (U, V) tap pair", k<3>, 80.0, seconds, d);
    run("tap mix, accumulating adds packed: 8 instructions", k_pkmix, 80.0, seconds, d);
    return 0;
}
