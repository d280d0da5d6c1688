// What are the "two kinds" of 1 GiB blocks that decide the HBM-bound kernels' rate (profiles/r05_cross_lane.md, section 4),
// and can planes be COMPOSED of both kinds through HIP's virtual-memory API instead of drawn in a lottery?
// ("next round" item 1.)  Streaming kernels only: the phenomenon is the step kernel's, but it shows in
// any kernel that reads two 1 GiB arrays and writes two at the same offsets.
//
//   E1  per hipMalloc block: solo read and solo write bandwidth
//   E2  per pair (X, Y): in-place update of both (reads X, Y; writes X, Y: "4 + 0" when X and Y are of one kind, a balanced
//       "2 + 2" when they are not), dual read, copy X -> Y; kinds from the in-place times
//   E3  quads (a, b -> c, d) over the kinds found: 4 + 0, 3 + 1, 2 + 2 balanced, 2 + 2 with both inputs of one kind
//   E4  per XCD: solo read of one block of each kind by the workgroups of ONE XCD at a time (is a kind "near" some XCDs?)
//   E5  in-place update of (X, Y) with Y's index skewed by up to 512 MiB (does an address offset change anything?)
//   E6  hipMemCreate / hipMemMap: granularity, sub-range mapping, chunk handles grouped into 1 GiB aggregates and classed
//       like E2, then four planes each ALTERNATING chunks of both kinds: quads on them against quads on whole aggregates
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_kinds hbm_kinds.hip ;  ./hbm_kinds [blocks=32] [chunk_MiB=8] [groups=24]
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

constexpr size_t GiB = 1ull << 30, MiB = 1ull << 20;
constexpr size_t N4 = GiB / 16; // float4 elements per block
constexpr int kGrid = 4096, kBlock = 256;

__global__ __launch_bounds__(256) void read_k(const float4 *__restrict__ a, size_t n, float *sink, int xcd)
{
    // xcd < 0: every workgroup; else only the workgroups the dispatcher deals to that XCD (round robin over 8)
    if (xcd >= 0 && (int)(blockIdx.x & 7) != xcd) return;
    const size_t stride = (size_t)(xcd >= 0 ? gridDim.x / 8 : gridDim.x) * blockDim.x;
    float s = 0.0f;
    for (size_t i = (size_t)(xcd >= 0 ? blockIdx.x / 8 : blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) *sink = s;
}
__global__ __launch_bounds__(256) void write_k(float4 *__restrict__ a, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void copy_k(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
__global__ __launch_bounds__(256) void dual_read_k(const float4 *__restrict__ a, const float4 *__restrict__ b, size_t n, float *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = a[i], w = b[i];
        s += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
    }
    if (s == 123.456f) *sink = s;
}
// reads a[i], b[j]; writes c[i], d[j], j = (i + skew) mod n: the traffic of one step (16 B per cell)
__global__ __launch_bounds__(256) void quad_k(const float4 *a, const float4 *b, float4 *c, float4 *d, size_t n, size_t skew)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        size_t j = i + skew;
        if (j >= n) j -= n;
        const float4 v = a[i], w = b[j];
        c[i] = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
        d[j] = make_float4(v.x - w.x, v.y - w.y, v.z - w.z, v.w - w.w);
    }
}

static hipEvent_t ev0, ev1;
static float *sink;

template <typename F>
static float timed(F &&launch, int reps = 3)
{
    launch();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(ev0, 0);
        launch();
        (void)hipEventRecord(ev1, 0);
        (void)hipEventSynchronize(ev1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        best = std::min(best, ms);
    }
    return best;
}
static float t_read(const void *a, int xcd = -1) { return timed([&] { read_k<<<kGrid, kBlock>>>((const float4 *)a, N4, sink, xcd); }); }
static float t_write(void *a) { return timed([&] { write_k<<<kGrid, kBlock>>>((float4 *)a, N4, 0.0f); }); }
static float t_copy(const void *a, void *b) { return timed([&] { copy_k<<<kGrid, kBlock>>>((const float4 *)a, (float4 *)b, N4); }); }
static float t_dual(const void *a, const void *b) { return timed([&] { dual_read_k<<<kGrid, kBlock>>>((const float4 *)a, (const float4 *)b, N4, sink); }); }
static float t_quad(void *a, void *b, void *c, void *d, size_t skew = 0)
{
    return timed([&] { quad_k<<<kGrid, kBlock>>>((const float4 *)a, (const float4 *)b, (float4 *)c, (float4 *)d, N4, skew); });
}
static float t_inplace(void *x, void *y, size_t skew = 0) { return t_quad(x, y, x, y, skew); }

// kinds from in-place pair times against block `ref`: 0 = as ref, 1 = the other; returns how many of kind 1
static int classify(const std::vector<void *> &b, std::vector<int> &kind, std::vector<float> &t, const char *what)
{
    const int n = (int)b.size();
    kind.assign((size_t)n, 0);
    t.assign((size_t)n, 0.0f);
    for (int j = 1; j < n; ++j) t[(size_t)j] = t_inplace(b[0], b[(size_t)j]);
    float lo = 1e30f, hi = 0.0f;
    for (int j = 1; j < n; ++j) { lo = std::min(lo, t[(size_t)j]); hi = std::max(hi, t[(size_t)j]); }
    int ones = 0;
    const bool two = hi - lo > 0.04f * hi;
    std::printf("%s: in-place pair with #0, ms:", what);
    for (int j = 1; j < n; ++j) {
        kind[(size_t)j] = two && t[(size_t)j] < 0.5f * (lo + hi) ? 1 : 0;
        ones += kind[(size_t)j];
        std::printf(" %.3f", t[(size_t)j]);
    }
    std::printf("\n%s: min %.3f max %.3f -> %s; kinds: ", what, lo, hi, two ? "two kinds" : "one kind");
    for (int j = 0; j < n; ++j) std::printf("%c", kind[(size_t)j] ? 'B' : 'a');
    std::printf("\n");
    std::fflush(stdout);
    return ones;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int nblocks = argc > 1 ? std::atoi(argv[1]) : 32;
    const size_t chunk = (size_t)(argc > 2 ? std::atoi(argv[2]) : 8) * MiB;
    const int groups = argc > 3 ? std::atoi(argv[3]) : 24;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    CK(hipMalloc((void **)&sink, 256));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    std::printf("device memory: %.1f GiB free of %.1f GiB\n", free_b / (double)GiB, total_b / (double)GiB);

    // ---------------- E1 ----------------
    std::vector<void *> blk;
    for (int i = 0; i < nblocks; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, GiB) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(p, 0, GiB));
        blk.push_back(p);
    }
    const int n = (int)blk.size();
    std::printf("E1: %d hipMalloc blocks of 1 GiB; address, solo read ms (GB/s), solo write ms (GB/s)\n", n);
    std::vector<float> rd((size_t)n), wr((size_t)n);
    for (int i = 0; i < n; ++i) {
        rd[(size_t)i] = t_read(blk[(size_t)i]);
        wr[(size_t)i] = t_write(blk[(size_t)i]);
        std::printf("  #%02d %p  read %.4f (%.0f)  write %.4f (%.0f)\n", i, blk[(size_t)i], rd[(size_t)i], GiB / rd[(size_t)i] * 1e-6,
                    wr[(size_t)i], GiB / wr[(size_t)i] * 1e-6);
    }
    std::fflush(stdout);

    // ---------------- E2 ----------------
    std::vector<int> kind;
    std::vector<float> t0;
    const int ones = classify(blk, kind, t0, "E2 hipMalloc blocks");
    const int m = std::min(n, 16);
    std::printf("E2: pair matrices over the first %d blocks (upper: in-place update ms; then dual read ms; then copy ms)\n", m);
    for (int pass = 0; pass < 3; ++pass) {
        std::printf("  %s\n", pass == 0 ? "in-place (reads X, Y; writes X, Y)" : pass == 1 ? "dual read" : "copy row -> column");
        for (int i = 0; i < m; ++i) {
            std::printf("  %c#%02d", kind[(size_t)i] ? 'B' : 'a', i);
            for (int j = 0; j < m; ++j) {
                if (j == i || (pass < 2 && j < i)) { std::printf("     . "); continue; }
                const float t = pass == 0 ? t_inplace(blk[(size_t)i], blk[(size_t)j]) : pass == 1 ? t_dual(blk[(size_t)i], blk[(size_t)j])
                                                                                                   : t_copy(blk[(size_t)i], blk[(size_t)j]);
                std::printf(" %.3f", t);
            }
            std::printf("\n");
        }
        std::fflush(stdout);
    }
    std::vector<int> A, B;
    for (int i = 0; i < n; ++i) (kind[(size_t)i] ? B : A).push_back(i);
    // ---------------- E3 ----------------
    auto quad_row = [&](const char *label, int a, int b, int c, int d) {
        const float t = t_quad(blk[(size_t)a], blk[(size_t)b], blk[(size_t)c], blk[(size_t)d]);
        std::printf("  %-44s blocks %2d %2d -> %2d %2d: %.4f ms = %.0f GB/s\n", label, a, b, c, d, t, 4.0 * GiB / t * 1e-6);
    };
    std::printf("E3: quads (reads two blocks, writes two others), %zu of kind a, %zu of kind B\n", A.size(), B.size());
    if (A.size() >= 4) { quad_row("4 + 0 (a a -> a a)", A[0], A[1], A[2], A[3]); quad_row("4 + 0 again", A[3], A[2], A[1], A[0]); }
    if (B.size() >= 4) quad_row("0 + 4 (B B -> B B)", B[0], B[1], B[2], B[3]);
    if (A.size() >= 3 && B.size() >= 1) { quad_row("3 + 1 (a B -> a a)", A[0], B[0], A[1], A[2]); quad_row("3 + 1 (a a -> a B)", A[0], A[1], A[2], B[0]); }
    if (A.size() >= 2 && B.size() >= 2) {
        quad_row("2 + 2 balanced (a B -> a B)", A[0], B[0], A[1], B[1]);
        quad_row("2 + 2 balanced (a B -> B a)", A[0], B[0], B[1], A[1]);
        quad_row("2 + 2 inputs of one kind (a a -> B B)", A[0], A[1], B[0], B[1]);
        quad_row("2 + 2 inputs of one kind (B B -> a a)", B[0], B[1], A[0], A[1]);
    }
    if (A.size() >= 1 && B.size() >= 3) quad_row("1 + 3 (a B -> B B)", A[0], B[0], B[1], B[2]);
    std::fflush(stdout);
    // ---------------- E4 ----------------
    std::printf("E4: solo read by the workgroups of one XCD at a time, ms (whole chip for comparison)\n");
    for (int which = 0; which < 2; ++which) {
        const std::vector<int> &K = which ? B : A;
        for (size_t r = 0; r < std::min<size_t>(2, K.size()); ++r) {
            std::printf("  kind %c #%02d: chip %.4f | xcd", which ? 'B' : 'a', K[r], t_read(blk[(size_t)K[r]]));
            for (int x = 0; x < 8; ++x) std::printf(" %.3f", t_read(blk[(size_t)K[r]], x));
            std::printf("\n");
        }
    }
    std::fflush(stdout);
    // ---------------- E5 ----------------
    {
        const size_t skews[] = {0, 256, 4096, 65536, MiB, 2 * MiB + 4096, 16 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB};
        std::printf("E5: in-place update of a pair with the second block's index skewed, ms\n");
        auto sweep = [&](const char *label, int x, int y) {
            std::printf("  %-18s #%02d #%02d:", label, x, y);
            for (size_t s : skews) std::printf(" %.3f", t_inplace(blk[(size_t)x], blk[(size_t)y], s / 16));
            std::printf("\n");
        };
        std::printf("  skew bytes:              ");
        for (size_t s : skews) std::printf(" %zu", s);
        std::printf("\n");
        if (A.size() >= 2) sweep("same kind (a, a)", A[0], A[1]);
        if (B.size() >= 2) sweep("same kind (B, B)", B[0], B[1]);
        if (A.size() >= 1 && B.size() >= 1) sweep("two kinds (a, B)", A[0], B[0]);
    }
    std::fflush(stdout);
    (void)ones;
    for (void *p : blk) (void)hipFree(p);
    blk.clear();

    // ---------------- E6 ----------------
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    std::printf("E6: allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    {
        // can a sub-range of a handle be mapped?
        hipMemGenericAllocationHandle_t h;
        void *va = nullptr;
        CK(hipMemCreate(&h, 64 * MiB, &prop, 0));
        CK(hipMemAddressReserve(&va, 64 * MiB, 0, nullptr, 0));
        hipError_t e = hipMemMap(va, 32 * MiB, 32 * MiB, h, 0);
        std::printf("E6: hipMemMap of the second half of a 64 MiB handle (offset 32 MiB): %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            e = hipMemMap((char *)va + 32 * MiB, 32 * MiB, 0, h, 0);
            std::printf("E6: ... and its first half behind it: %s\n", hipGetErrorString(e));
            if (e == hipSuccess) {
                e = hipMemSetAccess(va, 64 * MiB, &acc, 1);
                std::printf("E6: hipMemSetAccess over both: %s\n", hipGetErrorString(e));
                if (e == hipSuccess) e = hipMemset(va, 0, 64 * MiB);
                std::printf("E6: memset through the swapped mapping: %s\n", hipGetErrorString(e));
                (void)hipMemUnmap((char *)va + 32 * MiB, 32 * MiB);
            }
            (void)hipMemUnmap(va, 32 * MiB);
        }
        (void)hipGetLastError();
        (void)hipMemAddressFree(va, 64 * MiB);
        (void)hipMemRelease(h);
    }
    const size_t per_group = GiB / chunk;
    const size_t nchunks = (size_t)groups * per_group;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double t_create = now();
    for (size_t k = 0; k < nchunks; ++k) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        hs.push_back(h);
    }
    t_create = now() - t_create;
    const int G = (int)(hs.size() / per_group);
    std::printf("E6: %zu chunk handles of %zu MiB created in %.3f s (%d aggregates of 1 GiB)\n", hs.size(), chunk / MiB, t_create, G);
    if (G < 4) { std::printf("E6: too few aggregates\n"); return 0; }
    char *view = nullptr;
    CK(hipMemAddressReserve((void **)&view, (size_t)G * GiB, 0, nullptr, 0));
    double t_map = now();
    // (every chunk mapped, given access and unmapped on its own: ranges that span several mappings are not relied on)
    auto map_chunk = [&](char *va, hipMemGenericAllocationHandle_t h) -> hipError_t {
        hipError_t e = hipMemMap(va, chunk, 0, h, 0);
        return e == hipSuccess ? hipMemSetAccess(va, chunk, &acc, 1) : e;
    };
    for (size_t k = 0; k < (size_t)G * per_group; ++k) CK(map_chunk(view + k * chunk, hs[k]));
    t_map = now() - t_map;
    std::printf("E6: mapped in allocation order in %.3f s\n", t_map);
    CK(hipMemset(view, 0, (size_t)G * GiB));
    std::vector<void *> agg;
    for (int g = 0; g < G; ++g) agg.push_back(view + (size_t)g * GiB);
    std::vector<int> gk;
    std::vector<float> gt;
    classify(agg, gk, gt, "E6 aggregates of chunk handles");
    std::vector<int> GA, GB;
    for (int g = 0; g < G; ++g) (gk[(size_t)g] ? GB : GA).push_back(g);
    auto quad_agg = [&](const char *label, int a, int b, int c, int d) {
        const float t = t_quad(agg[(size_t)a], agg[(size_t)b], agg[(size_t)c], agg[(size_t)d]);
        std::printf("  %-44s aggregates %2d %2d -> %2d %2d: %.4f ms = %.0f GB/s\n", label, a, b, c, d, t, 4.0 * GiB / t * 1e-6);
        return t;
    };
    if (GA.size() >= 4) quad_agg("4 + 0 (a a -> a a)", GA[0], GA[1], GA[2], GA[3]);
    if (GB.size() >= 4) quad_agg("0 + 4 (B B -> B B)", GB[0], GB[1], GB[2], GB[3]);
    if (GA.size() >= 2 && GB.size() >= 2) {
        quad_agg("2 + 2 balanced (a B -> a B)", GA[0], GB[0], GA[1], GB[1]);
        quad_agg("2 + 2 inputs of one kind (a a -> B B)", GA[0], GA[1], GB[0], GB[1]);
        // four planes, each alternating chunks of two aggregates of different kinds: plane p = (GA[p / 2 .. ], GB[...])
        // planes 0, 1 share aggregates GA[0], GB[0] (even / odd chunk slots swapped), planes 2, 3 share GA[1], GB[1]
        char *mix = nullptr;
        CK(hipMemAddressReserve((void **)&mix, 4 * GiB, 0, nullptr, 0));
        for (int stride_chunks = 1; stride_chunks <= (int)per_group / 2; stride_chunks *= 4) {
            // unmap the four aggregates from the view, map them interleaved (runs of stride_chunks chunks)
            const int src[4] = {GA[0], GB[0], GA[1], GB[1]};
            for (int q = 0; q < 4; ++q)
                for (size_t s = 0; s < per_group; ++s) CK(hipMemUnmap(view + (size_t)src[q] * GiB + s * chunk, chunk));
            // plane p (0..3): slot s takes chunk s of aggregate (p ^ ((s / stride) & 1)) within its pair
            for (int p = 0; p < 4; ++p)
                for (size_t s = 0; s < per_group; ++s) {
                    const int pair = p / 2, flip = (int)((s / (size_t)stride_chunks) & 1);
                    const int from = src[pair * 2 + ((p & 1) ^ flip)];
                    CK(map_chunk(mix + (size_t)p * GiB + s * chunk, hs[(size_t)from * per_group + s]));
                }
            void *P[4] = {mix, mix + GiB, mix + 2 * GiB, mix + 3 * GiB};
            const float t = t_quad(P[0], P[1], P[2], P[3]);
            const float t2 = t_quad(P[0], P[2], P[1], P[3]);
            std::printf("  planes alternating kinds every %4zu MiB: quad %.4f ms = %.0f GB/s; other pairing %.4f ms\n",
                        (size_t)stride_chunks * chunk / MiB, t, 4.0 * GiB / t * 1e-6, t2);
            std::fflush(stdout);
            for (size_t k = 0; k < 4 * per_group; ++k) CK(hipMemUnmap(mix + k * chunk, chunk));
            for (int q = 0; q < 4; ++q)
                for (size_t s = 0; s < per_group; ++s)
                    CK(map_chunk(view + (size_t)src[q] * GiB + s * chunk, hs[(size_t)src[q] * per_group + s]));
        }
        (void)hipMemAddressFree(mix, 4 * GiB);
    } else {
        std::printf("E6: the aggregates are of one kind: no composition to time\n");
    }
    for (size_t k = 0; k < (size_t)G * per_group; ++k) CK(hipMemUnmap(view + k * chunk, chunk));
    for (auto h : hs) (void)hipMemRelease(h);
    (void)hipMemAddressFree(view, (size_t)G * GiB);
    std::printf("done\n");
    return 0;
}
