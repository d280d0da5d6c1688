// handoff_probe.hip -- what would trading aprons between workgroups INSIDE a launch cost?
//
// Emulates the exchange of a persistent window kernel for a 1080 x 1920 grid (profiles/archive/r03_sweeps.md, section 3):
// 15 x 16 workgroups of 1024 threads, one per CU, each owning a 72 x 120 tile of two f32 planes.  Per iteration a
// workgroup "computes" for a fixed time (a spin on s_memtime), stores the 4-cell ring of its tile (sc1 stores, 8 B
// per lane), drains (s_waitcnt vmcnt(0) in every storing wave, workgroup barrier), publishes a flag (one lane, sc1
// store), polls the flags of its up to 8 neighbours (one lane each, one sc1 vector load per poll, bounded), and loads its 4-cell apron from
// the neighbours' rings (sc1 loads) -- the form MI355X_MICROARCH.md lists as measured-valid for one workgroup per
// CU.  Every word carries the iteration it was written in; a loaded apron word of another iteration is counted as
// STALE.  Planes alternate between iterations, as the real kernel's would.
//
//   hipcc --offload-arch=gfx950 -O2 -o handoff_probe handoff_probe.hip && ./handoff_probe [iters] [work_us]
//
// Prints microseconds per iteration without the exchange, with it, and with parts of it only (the stale count is
// meaningful for the full exchange alone), and whether a poll timed out.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int TR = 72, TC = 120, K = 4, GY = 15, GX = 16, PITCH = 2048;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

struct Args {
    unsigned *plane[2][2]; // [buffer][species]
    int *flags, *abort_flag;
    unsigned long long *stamps; // 2 per workgroup: first entry, last exit (s_memrealtime)
    int *stale;
    int iters, work_cycles, exchange;
};

__global__ __launch_bounds__(1024) void probe(Args a)
{
    const int wg = blockIdx.x, by = wg / GX, bx = wg - by * GX, t = threadIdx.x;
    const int r0 = by * TR, c0 = bx * TC; // my tile's origin in the planes
    unsigned long long start = __builtin_amdgcn_s_memrealtime();
    int stale = 0;
    __shared__ int go;
    for (int it = 1; it <= a.iters; ++it) {
        // "compute"
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - w0) < a.work_cycles) __builtin_amdgcn_s_sleep(1);
        if (!a.exchange) continue;
        const int b = it & 1;
        // the ring of my tile: rows [0, K) and [TR - K, TR) x all columns; columns [0, K) and [TC - K, TC) x the rows between
        int rr = -1, cc = 0; // this thread's float2 of the ring (row, first column) inside the tile
        if (t < 240) { rr = t / 60; cc = (t % 60) * 2; }
        else if (t < 480) { rr = TR - K + (t - 240) / 60; cc = ((t - 240) % 60) * 2; }
        else if (t < 608) { rr = K + (t - 480) / 2; cc = ((t - 480) % 2) * 2; }
        else if (t < 736) { rr = K + (t - 608) / 2; cc = TC - K + ((t - 608) % 2) * 2; }
        if (rr >= 0 && (a.exchange & 1)) {
            const unsigned tag = ((unsigned)it << 16) | (unsigned)wg;
            const int off = ((r0 + rr) * PITCH + c0 + cc) * 4;
            for (int sp = 0; sp < 2; ++sp) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                v2u val = {tag, tag};
                __builtin_amdgcn_raw_buffer_store_b64(val, rsrc_of(a.plane[b][sp]), off, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0 && !(a.exchange & 2)) go = 1;
        if (t < 64 && (a.exchange & 2)) {
            // wave 0: lane 0 publishes, lanes 0..8 each watch one neighbour (one vector load polls all of them)
            if (t == 0) __builtin_amdgcn_raw_buffer_store_b32(it, rsrc_of(a.flags), wg * 4, 0, 16);
            const int dy = t / 3 - 1, dx = t % 3 - 1, ny = by + dy, nx = bx + dx;
            const bool watch = t < 9 && t != 4 && ny >= 0 && ny < GY && nx >= 0 && nx < GX;
            int ok = 1, spins = 0;
            for (;;) {
                const int seen = watch ? __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(a.flags), (ny * GX + nx) * 4, 0, 16) : it;
                if (!__builtin_amdgcn_ballot_w64(seen < it)) break;
                if (++spins > (1 << 18) || __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(a.abort_flag), 0, 0, 16)) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (t == 0) {
                if (!ok) __builtin_amdgcn_raw_buffer_store_b32(1, rsrc_of(a.abort_flag), 0, 0, 16);
                go = ok;
            }
        }
        __syncthreads();
        if (!go) break;
        // my apron: rows [-K, 0) and [TR, TR + K) x columns [-K, TC + K); columns [-K, 0) and [TC, TC + K) x rows [0, TR)
        int ar = -1000, ac = 0;
        if (t < 256) { ar = -K + t / 64; ac = -K + (t % 64) * 2; }
        else if (t < 512) { ar = TR + (t - 256) / 64; ac = -K + ((t - 256) % 64) * 2; }
        else if (t < 656) { ar = (t - 512) / 2; ac = -K + ((t - 512) % 2) * 2; }
        else if (t < 800) { ar = (t - 656) / 2; ac = TC + ((t - 656) % 2) * 2; }
        if (ar > -1000 && (a.exchange & 4)) {
            const int gr = r0 + ar, gc = c0 + ac;
            if (gr >= 0 && gr < GY * TR && gc >= 0 && gc + 1 < GX * TC)
                for (int sp = 0; sp < 2; ++sp) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_of(a.plane[b][sp]), (gr * PITCH + gc) * 4, 0, 16);
                    if (a.exchange == 7 && ((v[0] >> 16) != (unsigned)it || (v[1] >> 16) != (unsigned)it)) ++stale;
                    else if (v[0] == 0xffffffffu) ++stale; // keeps the load alive in the partial modes
                }
        }
    }
    if (stale) atomicAdd(a.stale, stale);
    if (t == 0) {
        a.stamps[2 * wg] = start;
        a.stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    const double work_us = argc > 2 ? std::atof(argv[2]) : 10.0;
    Args a{};
    const size_t plane_bytes = (size_t)(GY * TR + 8) * PITCH * 4;
    for (auto &buf : a.plane)
        for (auto &p : buf) {
            CHECK(hipMalloc(reinterpret_cast<void **>(&p), plane_bytes));
            CHECK(hipMemset(p, 0, plane_bytes));
        }
    CHECK(hipMalloc(reinterpret_cast<void **>(&a.flags), GY * GX * 4));
    CHECK(hipMalloc(reinterpret_cast<void **>(&a.abort_flag), 4));
    CHECK(hipMalloc(reinterpret_cast<void **>(&a.stale), 4));
    CHECK(hipMalloc(reinterpret_cast<void **>(&a.stamps), GY * GX * 16));
    a.iters = iters;
    a.work_cycles = (int)(work_us * 2300.0); // shader cycles at ~2.3 GHz
    std::vector<unsigned long long> st(2 * GY * GX);
    for (int rep = 0; rep < 3; ++rep)
        for (int exchange : {0, 7, 1, 2, 4, 3, 6}) {
            a.exchange = exchange;
            CHECK(hipMemset(a.flags, 0, GY * GX * 4));
            CHECK(hipMemset(a.abort_flag, 0, 4));
            CHECK(hipMemset(a.stale, 0, 4));
            hipLaunchKernelGGL(probe, dim3(GY * GX), dim3(1024), 0, 0, a);
            CHECK(hipDeviceSynchronize());
            int aborted = 0, stale = 0;
            CHECK(hipMemcpy(&aborted, a.abort_flag, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&stale, a.stale, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < GY * GX; ++w) { lo = st[2 * w] < lo ? st[2 * w] : lo; hi = st[2 * w + 1] > hi ? st[2 * w + 1] : hi; }
            static const char *what[] = {"no exchange", "ring stores + drain", "flag + poll", "stores + flag + poll", "apron loads", "", "flag + poll + loads", "full exchange"};
            std::printf("rep %d: %-22s: %.2f us per iteration (%d iterations, %.1f us of work each), stale words %d, poll timed out: %s\n", rep,
                        what[exchange], (double)(hi - lo) * 0.01 / iters, iters, work_us, stale,
                        aborted ? "YES" : "no");
        }
    return 0;
}
