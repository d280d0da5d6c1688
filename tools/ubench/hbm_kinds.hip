// What are the "two kinds" of 1 GiB blocks that decide the HBM-bound kernels' rate (profiles/r05_cross_lane.md, section 4),
// and can planes be COMPOSED of both kinds through HIP's virtual-memory API instead of drawn in a lottery?
// ("next round" item 1.)  Streaming kernels only: the phenomenon is the step kernel's, but it shows in
// any kernel that reads two 1 GiB arrays and writes two at the same offsets.
//
//   (run 1 of this file, profiles/r06_logs/hbm_kinds_run1.log, also had E4: per-XCD solo reads -- all alike -- and E5: the
//   second block of an in-place pair skewed by 256 B ... 512 MiB -- no effect; this is the form of run 2)
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


// Groups by in-place pair tests against one representative per group found so far: two blocks are of one group when
// updating both in place takes more than `thr` ms (run 1: 0.86 ... 0.96 within a group, 0.72 ... 0.80 across).
static constexpr float kSameGroupMs = 0.83f;
static std::vector<int> group_blocks(const std::vector<void *> &b, std::vector<int> &reps, const char *what)
{
    std::vector<int> g(b.size(), -1);
    reps.clear();
    std::printf("%s: block -> group (in-place ms against each representative)\n", what);
    for (size_t i = 0; i < b.size(); ++i) {
        std::printf("  #%02zu:", i);
        for (size_t r = 0; r < reps.size() && g[i] < 0; ++r) {
            const float t = t_inplace(b[(size_t)reps[r]], b[i]);
            std::printf(" %c %.3f", (char)('A' + r), t);
            if (t > kSameGroupMs) g[i] = (int)r;
        }
        if (g[i] < 0) { g[i] = (int)reps.size(); reps.push_back((int)i); }
        std::printf(" -> %c\n", (char)('A' + g[i]));
    }
    std::printf("%s: groups in allocation order: ", what);
    for (size_t i = 0; i < b.size(); ++i) std::printf("%c", (char)('A' + g[i]));
    std::printf("  (%zu groups)\n", reps.size());
    std::fflush(stdout);
    return g;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void quads_over_groups(const std::vector<void *> &b, const std::vector<int> &g, int ngroups, const char *what)
{
    // members per group
    std::vector<std::vector<int>> m((size_t)ngroups);
    for (size_t i = 0; i < b.size(); ++i) m[(size_t)g[i]].push_back((int)i);
    // groups by size, largest first
    std::vector<int> order((size_t)ngroups);
    for (int i = 0; i < ngroups; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return m[(size_t)x].size() > m[(size_t)y].size(); });
    auto q = [&](const char *label, int a, int bb, int c, int d) {
        const float fwd = t_quad(b[(size_t)a], b[(size_t)bb], b[(size_t)c], b[(size_t)d]);
        const float back = t_quad(b[(size_t)c], b[(size_t)d], b[(size_t)a], b[(size_t)bb]);
        std::printf("  %-34s %c%c -> %c%c  (#%02d #%02d -> #%02d #%02d): %.4f / %.4f ms there / back = %.0f GB/s\n", label, 'A' + g[(size_t)a],
                    'A' + g[(size_t)bb], 'A' + g[(size_t)c], 'A' + g[(size_t)d], a, bb, c, d, fwd, back, 8.0 * GiB / (fwd + back) * 1e-6);
        std::fflush(stdout);
    };
    std::printf("%s: quads (reads two, writes two; there and back = two steps of a Species)\n", what);
    const std::vector<int> &G0 = m[(size_t)order[0]];
    if (G0.size() >= 4) q("all of one group", G0[0], G0[1], G0[2], G0[3]);
    if (ngroups >= 2) {
        const std::vector<int> &G1 = m[(size_t)order[1]];
        if (G0.size() >= 2 && G1.size() >= 2) {
            q("U one group, V another (x y -> x y)", G0[0], G1[0], G0[1], G1[1]);
            q("crosswise (x y -> y x)", G0[0], G1[0], G1[1], G0[1]);
            q("slot 0 one group, slot 1 another", G0[0], G0[1], G1[0], G1[1]);
        }
        if (G0.size() >= 3 && G1.size() >= 1) q("3 + 1", G0[0], G0[1], G0[2], G1[0]);
    }
    if (ngroups >= 3) {
        const std::vector<int> &G1 = m[(size_t)order[1]], &G2 = m[(size_t)order[2]];
        if (G0.size() >= 2) q("x y -> x z", G0[0], G1[0], G0[1], G2[0]);
        if (G0.size() >= 2) q("x x -> y z", G0[0], G0[1], G1[0], G2[0]);
    }
    if (ngroups >= 4) {
        q("four groups (x y -> z w)", m[(size_t)order[0]][0], m[(size_t)order[1]][0], m[(size_t)order[2]][0], m[(size_t)order[3]][0]);
        q("four groups (x z -> y w)", m[(size_t)order[0]][0], m[(size_t)order[2]][0], m[(size_t)order[1]][0], m[(size_t)order[3]][0]);
        q("four groups (x w -> y z)", m[(size_t)order[0]][0], m[(size_t)order[3]][0], m[(size_t)order[1]][0], m[(size_t)order[2]][0]);
    }
}

int main(int argc, char **argv)
{
    const int nblocks = argc > 1 ? std::atoi(argv[1]) : 48;
    const size_t chunk = (size_t)(argc > 2 ? std::atoi(argv[2]) : 8) * MiB;
    const int groups = argc > 3 ? std::atoi(argv[3]) : 16;
    const int arena_gib = argc > 4 ? std::atoi(argv[4]) : 16;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    CK(hipMalloc((void **)&sink, 256));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    std::printf("device memory: %.1f GiB free of %.1f GiB\n", free_b / (double)GiB, total_b / (double)GiB);

    // ---------------- E1: hipMalloc blocks ----------------
    std::vector<void *> blk;
    for (int i = 0; i < nblocks; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, GiB) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(p, 0, GiB));
        blk.push_back(p);
    }
    const int n = (int)blk.size();
    float rlo = 1e30f, rhi = 0, wlo = 1e30f, whi = 0;
    for (int i = 0; i < n; ++i) {
        const float r = t_read(blk[(size_t)i]), w = t_write(blk[(size_t)i]);
        rlo = std::min(rlo, r); rhi = std::max(rhi, r); wlo = std::min(wlo, w); whi = std::max(whi, w);
    }
    std::printf("E1: %d hipMalloc blocks of 1 GiB: solo read %.4f ... %.4f ms (%.0f ... %.0f GB/s), solo write %.4f ... %.4f ms (%.0f ... %.0f GB/s)\n", n,
                rlo, rhi, GiB / rhi * 1e-6, GiB / rlo * 1e-6, wlo, whi, GiB / whi * 1e-6, GiB / wlo * 1e-6);
    // ---------------- E2: groups ----------------
    std::vector<int> reps;
    const std::vector<int> g = group_blocks(blk, reps, "E2 hipMalloc blocks");
    std::printf("E2: in-place ms among the groups' representatives (and each group's second member, if any)\n");
    {
        std::vector<int> probe = reps;
        for (size_t r = 0; r < reps.size(); ++r)
            for (int i = 0; i < n; ++i)
                if (g[(size_t)i] == (int)r && i != reps[r]) { probe.push_back(i); break; }
        std::printf("        ");
        for (int j : probe) std::printf("  %c#%02d ", 'A' + g[(size_t)j], j);
        std::printf("\n");
        for (size_t i = 0; i < probe.size(); ++i) {
            std::printf("  %c#%02d  ", 'A' + g[(size_t)probe[i]], probe[i]);
            for (size_t j = 0; j < probe.size(); ++j) {
                if (j <= i) { std::printf("     . "); continue; }
                std::printf(" %.3f ", t_inplace(blk[(size_t)probe[i]], blk[(size_t)probe[j]]));
            }
            std::printf("\n");
        }
    }
    std::fflush(stdout);
    // ---------------- E3 ----------------
    quads_over_groups(blk, g, (int)reps.size(), "E3 hipMalloc blocks");
    for (void *p : blk) (void)hipFree(p);
    blk.clear();

    // ---------------- E7: one large allocation, by 1 GiB sub-blocks ----------------
    {
        char *arena = nullptr;
        if (hipMalloc((void **)&arena, (size_t)arena_gib * GiB) == hipSuccess) {
            CK(hipMemset(arena, 0, (size_t)arena_gib * GiB));
            std::vector<void *> sub;
            for (int i = 0; i < arena_gib; ++i) sub.push_back(arena + (size_t)i * GiB);
            std::vector<int> sreps;
            const std::vector<int> sg = group_blocks(sub, sreps, "E7 sub-blocks of ONE hipMalloc");
            quads_over_groups(sub, sg, (int)sreps.size(), "E7 sub-blocks");
            (void)hipFree(arena);
        } else {
            (void)hipGetLastError();
            std::printf("E7: no %d GiB allocation\n", arena_gib);
        }
    }

    // ---------------- E6: the virtual-memory API ----------------
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    std::printf("E6: allocation granularity: minimum %zu, recommended %zu bytes; hipMemMap of a handle's sub-range (offset != 0): invalid argument (run 1)\n",
                gmin, grec);
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
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
    std::printf("E6: %zu chunk handles of %zu MiB created in %.3f s (%d aggregates of 1 GiB in allocation order)\n", hs.size(), chunk / MiB, t_create, G);
    if (G < 4) { std::printf("E6: too few aggregates\n"); return 0; }
    char *view = nullptr;
    CK(hipMemAddressReserve((void **)&view, (size_t)G * GiB, 0, nullptr, 0));
    // (every chunk mapped, given access and unmapped on its own: ranges that span several mappings are not relied on)
    auto map_chunk = [&](char *va, hipMemGenericAllocationHandle_t h) -> hipError_t {
        hipError_t e = hipMemMap(va, chunk, 0, h, 0);
        return e == hipSuccess ? hipMemSetAccess(va, chunk, &acc, 1) : e;
    };
    double t_map = now();
    for (size_t k = 0; k < (size_t)G * per_group; ++k) CK(map_chunk(view + k * chunk, hs[k]));
    t_map = now() - t_map;
    std::printf("E6: mapped in %.3f s\n", t_map);
    CK(hipMemset(view, 0, (size_t)G * GiB));
    std::vector<void *> agg;
    for (int gi = 0; gi < G; ++gi) agg.push_back(view + (size_t)gi * GiB);
    std::vector<int> areps;
    const std::vector<int> ag = group_blocks(agg, areps, "E6 aggregates of chunk handles");
    quads_over_groups(agg, ag, (int)areps.size(), "E6 aggregates");
    if (areps.size() >= 2) {
        // four planes, each alternating chunks of two aggregates of DIFFERENT groups (x0, y0 for planes 0 / 1; x1, y1 for 2 / 3)
        std::vector<std::vector<int>> m(areps.size());
        for (int i = 0; i < G; ++i) m[(size_t)ag[(size_t)i]].push_back(i);
        std::sort(m.begin(), m.end(), [](const std::vector<int> &x, const std::vector<int> &y) { return x.size() > y.size(); });
        int src[4] = {-1, -1, -1, -1};
        if (m.size() >= 4) { src[0] = m[0][0]; src[1] = m[1][0]; src[2] = m[2][0]; src[3] = m[3][0]; }
        else if (m[0].size() >= 2 && m[1].size() >= 2) { src[0] = m[0][0]; src[1] = m[1][0]; src[2] = m[0][1]; src[3] = m[1][1]; }
        if (src[0] >= 0) {
            char *mix = nullptr;
            CK(hipMemAddressReserve((void **)&mix, 4 * GiB, 0, nullptr, 0));
            std::printf("E6: four planes alternating chunks of aggregates #%d #%d (planes 0, 1) and #%d #%d (planes 2, 3)\n", src[0], src[1], src[2], src[3]);
            for (int stride_chunks = 1; stride_chunks <= (int)per_group / 2; stride_chunks *= 8) {
                for (int q = 0; q < 4; ++q)
                    for (size_t s = 0; s < per_group; ++s) CK(hipMemUnmap(view + (size_t)src[q] * GiB + s * chunk, chunk));
                for (int p = 0; p < 4; ++p)
                    for (size_t s = 0; s < per_group; ++s) {
                        const int pair = p / 2, flip = (int)((s / (size_t)stride_chunks) & 1);
                        const int from = src[pair * 2 + ((p & 1) ^ flip)];
                        CK(map_chunk(mix + (size_t)p * GiB + s * chunk, hs[(size_t)from * per_group + s]));
                    }
                void *P[4] = {mix, mix + GiB, mix + 2 * GiB, mix + 3 * GiB};
                const float t01 = t_quad(P[0], P[1], P[2], P[3]), t10 = t_quad(P[2], P[3], P[0], P[1]);
                const float u01 = t_quad(P[0], P[2], P[1], P[3]), u10 = t_quad(P[1], P[3], P[0], P[2]);
                std::printf("  alternating every %4zu MiB: (0 1 -> 2 3) %.4f / %.4f ms = %.0f GB/s; (0 2 -> 1 3) %.4f / %.4f ms = %.0f GB/s\n",
                            (size_t)stride_chunks * chunk / MiB, t01, t10, 8.0 * GiB / (t01 + t10) * 1e-6, u01, u10, 8.0 * GiB / (u01 + u10) * 1e-6);
                std::fflush(stdout);
                for (size_t k = 0; k < 4 * per_group; ++k) CK(hipMemUnmap(mix + k * chunk, chunk));
                for (int q = 0; q < 4; ++q)
                    for (size_t s = 0; s < per_group; ++s)
                        CK(map_chunk(view + (size_t)src[q] * GiB + s * chunk, hs[(size_t)src[q] * per_group + s]));
            }
            (void)hipMemAddressFree(mix, 4 * GiB);
        }
    }
    for (size_t k = 0; k < (size_t)G * per_group; ++k) CK(hipMemUnmap(view + k * chunk, chunk));
    for (auto h : hs) (void)hipMemRelease(h);
    (void)hipMemAddressFree(view, (size_t)G * GiB);
    std::printf("done\n");
    return 0;
}
