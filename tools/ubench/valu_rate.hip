// Micro-benchmark: issue rate of the VALU instructions the step kernels are made of, on gfx950.
// Each wave runs ITER iterations of 16 independent accumulator updates with one instruction
// kind; 256 CUs x 4 SIMDs x W waves.  Prints lane-ops/s (a packed op counts 2 lane-ops).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){a[i], a[i] + 1.f}; }
    float b = seed * 0.5f, c = seed * 0.25f;
    v2f pb = (v2f){b, c}, pc = (v2f){c, b};
    for (int it = 0; it < iters; ++it) {
#define OP0(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP1(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define OP2(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define OP3(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define OP4(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
#define OP5(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
#define OP6(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define OP7(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 5) & 15]));
#define OP8(i) asm volatile("v_sub_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
#define OP9(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP10(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define OP11(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[0]) : "v"(b));  /* one dependent chain */
#define OP12(i) asm volatile("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %3, %0\n\tv_add_f32 %4, %4, %0" : "=&v"(p[i].x), "+v"(a[i]) : "v"(b), "v"(c), "v"(a[i]));
        if (OP == 0) { REP16(OP0) }
        if (OP == 1) { REP16(OP1) }
        if (OP == 2) { REP16(OP2) }
        if (OP == 3) { REP16(OP3) }
        if (OP == 4) { REP16(OP4) }
        if (OP == 5) { REP16(OP5) }
        if (OP == 6) { REP16(OP6) }
        if (OP == 7) { REP16(OP7) }
        if (OP == 8) { REP16(OP8) }
        if (OP == 9) { REP16(OP9) }
        if (OP == 10) { REP16(OP10) }
        if (OP == 11) { REP16(OP11) }
        if (OP == 12) { REP16(OP12) }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
double run(int waves_per_simd, int iters, float *d)
{
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    return (double)grid.x * 256 * iters * 16 / (ms * 1e-3); // instructions-lanes per second
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    const int iters = 400000;
    const char *names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32",
                           "v_mov_b32", "v_mov_b32_dpp", "v_sub_f32_dpp", "v_fmac_f32", "v_cndmask_b32", "v_add dep-chain", "sub,mul,add x3"};
    printf("%-16s %8s %8s %8s %8s   (T lane-instr/s; packed ops do 2 flop-lanes each)\n", "op", "1w/SIMD", "2w", "3w", "4w");
    for (int op = 0; op < 13; ++op) {
        printf("%-16s", names[op]);
        for (int w = 1; w <= 4; ++w) {
            double r = 0;
            switch (op) {
            case 0: r = run<0>(w, iters, d); break;
            case 1: r = run<1>(w, iters, d); break;
            case 2: r = run<2>(w, iters, d); break;
            case 3: r = run<3>(w, iters, d); break;
            case 4: r = run<4>(w, iters, d); break;
            case 5: r = run<5>(w, iters, d); break;
            case 6: r = run<6>(w, iters, d); break;
            case 7: r = run<7>(w, iters, d); break;
            case 8: r = run<8>(w, iters, d); break;
            case 9: r = run<9>(w, iters, d); break;
            case 10: r = run<10>(w, iters, d); break;
            case 11: r = run<11>(w, iters, d); break;
            case 12: r = 3 * run<12>(w, iters, d); break;
            }
            printf(" %8.2f", r / 1e12);
        }
        printf("\n");
    }
    return 0;
}
