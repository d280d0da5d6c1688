// Micro-benchmark, second take: VALU issue rate by waves per SIMD with the OCCUPANCY PINNED.
// tools/ubench/valu_rate.hip launches 256 x W blocks and lets the dispatcher place them, which
// leaves some CUs with 4 blocks and others with 2 (its "3 waves" column is 3/4 of its "4 waves"
// column for that reason).  Here every block asks for 160 KB / W of LDS, so a CU holds exactly W
// blocks of 4 waves (one per SIMD) at a time, and the grid is 16 rounds of that.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip && ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    extern __shared__ float lds[];
    float a[16], t[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; t[i] = 0.f; }
    float b = seed * 0.5f, c = seed * 0.25f;
    const int addr = ((threadIdx.x + 1) & 63) << 2;
    for (int it = 0; it < iters; ++it) {
#define OP0(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define OP1(i) asm volatile("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %3, %0\n\tv_add_f32 %4, %4, %0" : "=&v"(t[i]), "+v"(a[i]) : "v"(b), "v"(c), "v"(a[i]));
#define OP2(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t[i]) : "v"(a[(i + 5) & 15]));
#define OP3(i) asm volatile("v_sub_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t[i]) : "v"(a[(i + 5) & 15]), "v"(b));
#define OP4(i) asm volatile("s_nop 1\n\tv_sub_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t[i]) : "v"(a[(i + 5) & 15]), "v"(b));
#define OP5(i) asm volatile("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(t[i]) : "v"(a[i]), "v"(b));
#define OP6(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
// the tap mix of the step kernel: corner (dpp-sub, mul, add), side (sub div:2, add), corner, side
#define OP7(i) asm volatile("v_sub_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0\n\tv_sub_f32 %0, %2, %3\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0" : "=&v"(t[i]), "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(b), "v"(c));
// the same without DPP
#define OP8(i) asm volatile("v_sub_f32 %0, %2, %3\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0\n\tv_sub_f32 %0, %2, %3\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0" : "=&v"(t[i]), "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(b), "v"(c));
// ds_bpermute_b32 alone, and beside the no-DPP tap mix (result consumed one iteration later)
#define OP9(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(t[i]) : "v"(addr), "v"(a[(i + 5) & 15]));
#define OP10(i) asm volatile("ds_bpermute_b32 %0, %5, %2\n\tv_sub_f32 %0, %2, %3\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0\n\tv_sub_f32 %0, %2, %3\n\tv_mul_f32 %0, %4, %0\n\tv_add_f32 %1, %1, %0\n\t" \
                            "v_sub_f32_e64 %0, %2, %3 div:2\n\tv_add_f32 %1, %1, %0" : "=&v"(t[i]), "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(b), "v"(c), "v"(addr));
        if (OP == 9) { REP16(OP9) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 10) { REP16(OP10) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 0) { REP16(OP0) }
        if (OP == 1) { REP16(OP1) }
        if (OP == 2) { REP16(OP2) }
        if (OP == 3) { REP16(OP3) }
        if (OP == 4) { REP16(OP4) }
        if (OP == 5) { REP16(OP5) }
        if (OP == 6) { REP16(OP6) }
        if (OP == 7) { REP16(OP7) }
        if (OP == 8) { REP16(OP8) }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + t[i];
    if (s == 12345.678f) { out[threadIdx.x] = s; lds[threadIdx.x] = s; }
}

template <int OP>
double run(int w, int iters, float *d, int per_op)
{
    const size_t lds = (160 * 1024 / w) & ~255u;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256 * w * 16), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, lds, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, grid, block, lds, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    return (double)grid.x * 256 * iters * 16 * per_op / (ms * 1e-3);
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    const int iters = 4000;
    const char *names[] = {"v_add_f32", "sub,mul,add", "v_mov_b32_dpp", "v_sub_f32_dpp", "s_nop1+v_sub_dpp", "v_sub div:2",
                           "v_fma_f32", "tap mix (dpp)", "tap mix (no dpp)", "ds_bpermute_b32", "tap mix + bperm"};
    const int per_op[] = {1, 3, 1, 1, 1, 1, 1, 10, 10, 1, 10};
    printf("%-18s", "op \\ waves/SIMD");
    for (int w = 1; w <= 8; ++w) printf(" %7d", w);
    printf("   (T lane-instr/s, VALU instructions only; peak 78.6 at 2.4 GHz)\n");
    for (int op = 0; op < 11; ++op) {
        printf("%-18s", names[op]);
        for (int w = 1; w <= 8; ++w) {
            double r = 0;
            switch (op) {
            case 0: r = run<0>(w, iters, d, per_op[op]); break;
            case 1: r = run<1>(w, iters, d, per_op[op]); break;
            case 2: r = run<2>(w, iters, d, per_op[op]); break;
            case 3: r = run<3>(w, iters, d, per_op[op]); break;
            case 4: r = run<4>(w, iters, d, per_op[op]); break;
            case 5: r = run<5>(w, iters, d, per_op[op]); break;
            case 6: r = run<6>(w, iters, d, per_op[op]); break;
            case 7: r = run<7>(w, iters, d, per_op[op]); break;
            case 8: r = run<8>(w, iters, d, per_op[op]); break;
            case 9: r = run<9>(w, iters, d, per_op[op]); break;
            case 10: r = run<10>(w, iters, d, per_op[op]); break;
            }
            printf(" %7.2f", r / 1e12);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
