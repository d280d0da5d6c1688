// gs_fields.cpp -- planes (gs_field): one f32 array per species, slot and row slab in HBM, with ghost rows; the
// Concentration contract of the reference (data/src/concentration/mod.rs:198-296) on the device: creation, fills, uploads,
// downloads (blocking and overlapped), colour mapping, and placement by measurement.
#include <algorithm>
#include "gs_internal.h"

using namespace gsi;

extern "C" {

int32_t gs_field_destroy(gs_ctx *ctx, gs_field *f)
{
    if (!f) return GS_OK;
    if (ctx) (void)sync_all(ctx);
    for (size_t i = 0; i < f->s.size(); ++i)
        if (f->s[i].alloc) {
            if (ctx && i < ctx->slabs.size()) (void)hipSetDevice(ctx->slabs[i].device);
            (void)hipFree(f->s[i].alloc);
        }
    delete f;
    return GS_OK;
}

int32_t gs_field_create(gs_ctx *ctx, gs_field **out, uint64_t rows, uint64_t cols)
{
    if (!ctx || !out) return fail(GS_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t S = (uint64_t)ctx->total_slabs();
    // An empty grid is legal in the reference (ndarray holds zero-sized arrays and every step is a
    // no-op on them); here it is a single slab with no rows or no columns that no kernel touches.
    if (rows < S && !(rows == 0 && S == 1))
        return fail(GS_ERR_INVALID, "%llu rows cannot be split over %llu slabs",
                    (unsigned long long)rows, (unsigned long long)S);
    const int pad = ctx->o.pitch_pad > 0 ? ((ctx->o.pitch_pad + 3) / 4) * 4 : 0;
    const uint64_t pitch = (cols == 0 ? 64 : ((cols + 63) / 64) * 64) + (uint64_t)pad;
    if (pitch > 0x7ffffff0ull) return fail(GS_ERR_UNSUPPORTED, "too many columns");
    gs_field *f = new (std::nothrow) gs_field();
    if (!f) return fail(GS_ERR_NOMEM, "out of host memory");
    f->ctx = ctx;
    f->rows = rows;
    f->cols = cols;
    f->pitch = (int32_t)pitch;
    f->s.resize(ctx->slabs.size());
    for (size_t i = 0; i < ctx->slabs.size(); ++i) {
        const uint64_t k = (uint64_t)ctx->global_index((int)i);
        const uint64_t r0 = k * rows / S, r1 = (k + 1) * rows / S;
        if (r1 - r0 > 0x7ffffff0ull || (r1 - r0 + 2 * kGhostRows) * pitch > 0x7ffffff0ull * 4ull) {
            gs_field_destroy(ctx, f);
            return fail(GS_ERR_UNSUPPORTED, "slab too large for 32-bit row indexing");
        }
        FieldSlab &fs = f->s[i];
        fs.g_row0 = r0;
        fs.rows = (int32_t)(r1 - r0);
        const size_t n = (size_t)(fs.rows + 2 * kGhostRows) * pitch + 2 * kGuardFloats;
        hipError_t e = hipSetDevice(ctx->slabs[i].device);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&fs.alloc), n * sizeof(float));
        // Concentration::default() is zero-filled in the reference (ndarray default); ghost
        // rows and padding start as zeros too.
        if (e == hipSuccess) e = hipMemsetAsync(fs.alloc, 0, n * sizeof(float), ctx->slabs[i].compute);
        if (e != hipSuccess) {
            gs_field_destroy(ctx, f);
            return fail(e == hipErrorOutOfMemory ? GS_ERR_NOMEM : GS_ERR_HIP, "plane allocation failed: %s",
                        hipGetErrorString(e));
        }
        fs.row0 = fs.alloc + kGuardFloats + (size_t)kGhostRows * pitch;
    }
    int32_t st = sync_all(ctx);
    if (st != GS_OK) { gs_field_destroy(ctx, f); return st; }
    f->ghost_depth = kGhostRows; // all zeros, ghosts included
    *out = f;
    return GS_OK;
}

// Placement by measurement (gs_hip.h).  Where an allocation lands in HBM is below what a process controls (physical
// frames), and it decides the level at which four 1 GiB planes read -- 0.65 ... 0.75 of 8 TB/s for the HBM-bound
// single-step kernel at 16384^2, 1.09 ... 1.2 M Mcells x steps/s for the marching kernel.  What a process CAN do is draw
// more blocks than it needs, find out by timed probes which of the two kinds each is, and keep a set split over both.
int32_t gs_fields_place(gs_ctx *ctx, gs_field *const planes[4], int32_t candidates, float *first_ms, float *best_ms)
{
    if (!ctx || !planes) return fail(GS_ERR_INVALID, "null argument");
    if (candidates < 1 || candidates > 124) return fail(GS_ERR_INVALID, "1 to 124 extra candidate blocks, not %d", candidates);
    if (ctx->slabs.size() != 1) return fail(GS_ERR_UNSUPPORTED, "placement by measurement is for contexts with one slab per process");
    for (int i = 0; i < 4; ++i) {
        if (!planes[i] || planes[i]->ctx != ctx) return fail(GS_ERR_INVALID, "bad plane %d", i);
        GS_TRY(same_shape(planes[0], planes[i]));
        for (int j = 0; j < i; ++j)
            if (planes[i] == planes[j]) return fail(GS_ERR_INVALID, "the four planes must be distinct");
    }
    if (first_ms) *first_ms = 0.0f;
    if (best_ms) *best_ms = 0.0f;
    const gs_field *f0 = planes[0];
    if (f0->rows == 0 || f0->cols == 0) return GS_OK;
    GS_TRY(sync_all(ctx));
    SlabRt &sl = ctx->slabs[0];
    GS_HIP(hipSetDevice(sl.device));
    const size_t pitch = (size_t)f0->pitch;
    const size_t n = (size_t)(f0->s[0].rows + 2 * kGhostRows) * pitch + 2 * kGuardFloats;
    std::vector<float *> blocks;
    for (int i = 0; i < 4; ++i) blocks.push_back(planes[i]->s[0].alloc);
    auto release = [&](int keep_from) { // frees the blocks from index keep_from on
        for (size_t i = (size_t)keep_from; i < blocks.size(); ++i)
            if (blocks[i]) (void)hipFree(blocks[i]);
        blocks.resize((size_t)keep_from);
    };
    auto row0_of = [&](float *b) { return b + kGuardFloats + (size_t)kGhostRows * pitch; };
    // a probe: four single steps ping-ponging between (a, b) and (c, d), timed with the context's events
    GsStepArgs base = make_args(ctx, planes[0], planes[1], planes[2], planes[3], 0, 1);
    base.ra0 = 0;
    base.ra1 = base.rows;
    const bool fused = ctx->o.math == GS_MATH_FUSED;
    static const bool trace = gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1) != 0;
    auto probe = [&](const int (&pick)[4], float *ms) -> int32_t {
        float *p[4];
        for (int i = 0; i < 4; ++i) p[i] = row0_of(blocks[(size_t)pick[i]]);
        for (int rep = 0; rep < 5; ++rep) { // the first step is not timed
            if (rep == 1) GS_HIP(hipEventRecord(sl.t0, sl.compute));
            GsStepArgs a = base;
            const int in = (rep & 1) * 2, out = 2 - in;
            a.in_u = p[in]; a.in_v = p[in + 1]; a.out_u = p[out]; a.out_v = p[out + 1];
            const char *name = nullptr;
            const hipError_t e = fused ? gs_launch_stream_fused(a, sl.compute, &name) : gs_launch_stream_strict(a, sl.compute, &name);
            if (e != hipSuccess) return fail(GS_ERR_HIP, "probe launch failed: %s", hipGetErrorString(e));
        }
        GS_HIP(hipEventRecord(sl.t1, sl.compute));
        GS_HIP(hipEventSynchronize(sl.t1));
        GS_HIP(hipEventElapsedTime(ms, sl.t0, sl.t1));
        if (trace)
            std::fprintf(stderr, "gs_hip placement: blocks %3d %3d %3d %3d (%p %p %p %p): %.4f ms per step\n", pick[0], pick[1], pick[2],
                         pick[3], (void *)blocks[(size_t)pick[0]], (void *)blocks[(size_t)pick[1]], (void *)blocks[(size_t)pick[2]],
                         (void *)blocks[(size_t)pick[3]], *ms / 4.0f);
        return GS_OK;
    };
    // What the probes of rounds 4 and 5 say about this memory (profiles/r05_cross_lane.md, section 4): the blocks hipMalloc
    // hands out come from TWO pools, most of them from one; four planes read by how they are split over the two -- 4 + 0:
    // 0.82 ms per single step at 16384^2, 3 + 1: 0.77, 2 + 2: 0.72 when the inputs (and so the outputs) are one of each,
    // 0.76-0.80 when both inputs are of one pool.  So: draw blocks in batches, time random 4-subsets, class the blocks by
    // the mean time of the subsets they were in (the rare kind pulls its subsets down), stop drawing once there are two of
    // each class, and time the arrangements (common, rare, common, rare) of the clearest members.
    int best[4] = {0, 1, 2, 3};
    float best_t = 0.0f, first_t = 0.0f;
    // The planes keep their contents: a copy is set aside before the first probe writes into their blocks and goes into
    // the blocks they end up with (or back, if anything fails).  Without room for the copy nothing is done.
    float *stash[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i)
        if (hipMalloc(reinterpret_cast<void **>(&stash[i]), n * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; ++j) (void)hipFree(stash[j]);
            return GS_OK;
        }
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = hipMemcpyAsync(stash[i], blocks[(size_t)i], n * sizeof(float), hipMemcpyDeviceToDevice, sl.compute);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(sl.compute);
            for (float *p : stash) (void)hipFree(p);
            return fail(GS_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
        }
        (void)hipMemsetAsync(blocks[(size_t)i], 0, n * sizeof(float), sl.compute); // every probe runs on zeros, like the candidates
    }
    auto search = [&]() -> int32_t {
        uint32_t rng = 0x9e3779b9u;
        auto draw = [&](uint32_t n) { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 8) % n); };
        std::vector<float> sum;
        std::vector<int> cnt;
        std::vector<int> rare, common; // block indices by class, clearest first
        auto timed = [&](const int (&pick)[4], bool count) -> int32_t {
            float ms = 0.0f;
            GS_TRY(probe(pick, &ms));
            if (count)
                for (int i = 0; i < 4; ++i) { sum[(size_t)pick[i]] += ms; ++cnt[(size_t)pick[i]]; }
            if (best_t == 0.0f || ms < 0.995f * best_t) { best_t = ms; std::memcpy(best, pick, sizeof best); }
            return GS_OK;
        };
        int drawn = 0;
        while (true) {
            const int old = (int)blocks.size();
            const int batch = old == 4 ? (candidates < 12 ? candidates : 12) : (candidates - drawn < 16 ? candidates - drawn : 16);
            for (int i = 0; i < batch; ++i) {
                float *b = nullptr;
                if (hipMalloc(reinterpret_cast<void **>(&b), n * sizeof(float)) != hipSuccess) { // fewer candidates: fine
                    (void)hipGetLastError();
                    drawn = candidates;
                    break;
                }
                const hipError_t e = hipMemsetAsync(b, 0, n * sizeof(float), sl.compute); // zeros, as gs_field_create leaves a plane
                blocks.push_back(b);
                if (e != hipSuccess) { release(4); return fail(GS_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)); }
                ++drawn;
            }
            const int have = (int)blocks.size();
            sum.resize((size_t)have, 0.0f);
            cnt.resize((size_t)have, 0);
            if (old == 4) { // the four that are there
                const int pick[4] = {0, 1, 2, 3};
                const int32_t st = timed(pick, true);
                if (st != GS_OK) { release(4); return st; }
                first_t = best_t;
            }
            if (have == 4) break; // nothing could be drawn
            // three random 4-subsets per new block, each holding it
            for (int nb = old == 4 ? 0 : old; nb < have; ++nb)
                for (int rep = 0; rep < 3; ++rep) {
                    int pick[4] = {nb, nb, nb, nb};
                    const int at = draw(4);
                    for (int i = 0; i < 4; ++i) {
                        if (i == at) continue;
                        bool fresh;
                        do {
                            pick[i] = draw((uint32_t)have);
                            fresh = pick[i] != nb;
                            for (int j = 0; j < i; ++j) fresh = fresh && (j == at || pick[j] != pick[i]);
                        } while (!fresh);
                    }
                    const int32_t st = timed(pick, true);
                    if (st != GS_OK) { release(4); return st; }
                }
            // two classes?  by the means, split half way between the extremes when they are at least 3 % apart
            std::vector<int> order;
            for (int b = 0; b < have; ++b)
                if (cnt[(size_t)b] > 0) order.push_back(b);
            auto mean = [&](int b) { return sum[(size_t)b] / (float)cnt[(size_t)b]; };
            std::sort(order.begin(), order.end(), [&](int x, int y) { return mean(x) < mean(y); });
            rare.clear();
            common.clear();
            const float lo = mean(order.front()), hi = mean(order.back());
            if (hi - lo > 0.03f * hi)
                for (int b : order) (mean(b) < 0.5f * (lo + hi) ? rare : common).push_back(b);
            std::reverse(common.begin(), common.end()); // clearest first
            if (trace)
                std::fprintf(stderr, "gs_hip placement: %d blocks, means %.4f ... %.4f ms per step, %zu of the rarer kind\n", have, lo / 4.0f,
                             hi / 4.0f, rare.size());
            static const bool draw_all = gs_env_int("GS_HIP_PLACE_ALL", 0, 0, 1) != 0; // diagnostics: never stop early
            if ((rare.size() >= 2 && common.size() >= 2 && !draw_all) || drawn >= candidates) break;
        }
        // (common, rare, common, rare) over the three clearest of each class: U's planes of one pool, V's of the other
        if (rare.size() >= 2 && common.size() >= 2) {
            const int nr = (int)rare.size() < 3 ? (int)rare.size() : 3, nc = (int)common.size() < 3 ? (int)common.size() : 3;
            for (int r0 = 0; r0 < nr; ++r0)
                for (int r1 = r0 + 1; r1 < nr; ++r1)
                    for (int c0 = 0; c0 < nc; ++c0)
                        for (int c1 = c0 + 1; c1 < nc; ++c1) {
                            const int pick[4] = {common[(size_t)c0], rare[(size_t)r0], common[(size_t)c1], rare[(size_t)r1]};
                            const int32_t st = timed(pick, false);
                            if (st != GS_OK) { release(4); return st; }
                        }
        }
        return GS_OK;
    };
    const int32_t searched = search(); // (on failure only the planes' own four blocks are left)
    if (searched != GS_OK) { best[0] = 0; best[1] = 1; best[2] = 2; best[3] = 3; }
    // hand the chosen blocks to the planes, with the contents set aside
    std::vector<float *> chosen(4);
    for (int i = 0; i < 4; ++i) chosen[(size_t)i] = blocks[(size_t)best[i]];
    hipError_t copied = hipSuccess;
    for (int i = 0; i < 4; ++i) {
        FieldSlab &fs = planes[i]->s[0];
        fs.alloc = chosen[(size_t)i];
        fs.row0 = row0_of(fs.alloc);
        const hipError_t e = hipMemcpyAsync(fs.alloc, stash[i], n * sizeof(float), hipMemcpyDeviceToDevice, sl.compute);
        if (e != hipSuccess) copied = e;
    }
    const hipError_t synced = hipStreamSynchronize(sl.compute);
    for (float *b : blocks) {
        bool used = false;
        for (float *c : chosen) used = used || c == b;
        if (!used) (void)hipFree(b);
    }
    for (float *p : stash) (void)hipFree(p);
    (void)hipGetLastError();
    if (copied != hipSuccess || synced != hipSuccess)
        return fail(GS_ERR_HIP, "restoring the planes' contents failed: %s", hipGetErrorString(copied != hipSuccess ? copied : synced));
    if (searched != GS_OK) return searched;
    if (first_ms) *first_ms = first_t / 4.0f;
    if (best_ms) *best_ms = best_t / 4.0f;
    return GS_OK;
}

int32_t gs_field_shape(const gs_field *f, uint64_t *rows, uint64_t *cols)
{
    if (!f) return fail(GS_ERR_INVALID, "null field");
    if (rows) *rows = f->rows;
    if (cols) *cols = f->cols;
    return GS_OK;
}

int32_t gs_field_local_rows(const gs_field *f, uint64_t *row0, uint64_t *row1)
{
    if (!f || f->s.empty()) return fail(GS_ERR_INVALID, "null field");
    if (row0) *row0 = f->s.front().g_row0;
    if (row1) *row1 = f->s.back().g_row0 + (uint64_t)f->s.back().rows;
    return GS_OK;
}

int32_t gs_field_raw_shape(const gs_field *f, uint64_t *raw_rows, uint64_t *pitch)
{
    if (!f) return fail(GS_ERR_INVALID, "null field");
    uint64_t n = 0;
    for (auto &fs : f->s) n += (uint64_t)fs.rows + 2 * kGhostRows;
    if (raw_rows) *raw_rows = n;
    if (pitch) *pitch = (uint64_t)f->pitch;
    return GS_OK;
}

int32_t gs_field_fill_slice(gs_ctx *ctx, gs_field *f, uint64_t r0, uint64_t r1, uint64_t c0, uint64_t c1,
                            float value)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad handle");
    // ndarray slicing panics on out-of-range or reversed ranges (concentration/mod.rs:333-334)
    if (r0 > r1 || c0 > c1 || r1 > f->rows || c1 > f->cols)
        return fail(GS_ERR_INVALID, "slice [%llu..%llu, %llu..%llu] outside [%llu, %llu]",
                    (unsigned long long)r0, (unsigned long long)r1, (unsigned long long)c0,
                    (unsigned long long)c1, (unsigned long long)f->rows, (unsigned long long)f->cols);
    // After an asynchronous run on a slab chain (or on row bands) the last pass's boundary kernels and
    // ghost pushes may still be in flight on the halo / band streams: the fill below must not race them.
    GS_TRY(sync_all(ctx));
    for (size_t i = 0; i < f->s.size(); ++i) {
        const FieldSlab &fs = f->s[i];
        const uint64_t lo = r0 > fs.g_row0 ? r0 : fs.g_row0;
        const uint64_t hi = r1 < fs.g_row0 + fs.rows ? r1 : fs.g_row0 + fs.rows;
        if (lo >= hi || c0 >= c1) continue;
        GS_HIP(hipSetDevice(ctx->slabs[i].device));
        hipError_t e = gs_launch_fill_rect(fs.row0, f->pitch, (int32_t)(lo - fs.g_row0),
                                           (int32_t)(hi - fs.g_row0), (int32_t)c0, (int32_t)c1, value,
                                           ctx->slabs[i].compute);
        if (e != hipSuccess) return fail(GS_ERR_HIP, "fill launch failed: %s", hipGetErrorString(e));
    }
    GS_TRY(sync_all(ctx));
    f->ghost_depth = 0;
    return GS_OK;
}

int32_t gs_field_fill(gs_ctx *ctx, gs_field *f, float value)
{
    if (!f) return fail(GS_ERR_INVALID, "null field");
    return gs_field_fill_slice(ctx, f, 0, f->rows, 0, f->cols, value);
}

int32_t gs_field_finalize(gs_ctx *ctx, gs_field *f)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad handle");
    if (f->ghost_depth == 0) GS_TRY(refresh_ghosts(ctx, f)); // deeper needs are met lazily by gs_step / gs_run
    return GS_OK;
}

int32_t gs_field_upload(gs_ctx *ctx, gs_field *f, const float *host)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad argument");
    if (f->rows == 0 || f->cols == 0) return GS_OK; // nothing to copy (host may be null)
    if (!host) return fail(GS_ERR_INVALID, "bad argument");
    GS_TRY(sync_all(ctx));
    const uint64_t first = f->s.front().g_row0;
    for (size_t i = 0; i < f->s.size(); ++i) {
        const FieldSlab &fs = f->s[i];
        GS_HIP(hipSetDevice(ctx->slabs[i].device));
        GS_HIP(hipMemcpy2D(fs.row0, (size_t)f->pitch * sizeof(float), host + (fs.g_row0 - first) * f->cols,
                           (size_t)f->cols * sizeof(float), (size_t)f->cols * sizeof(float), (size_t)fs.rows,
                           hipMemcpyHostToDevice));
    }
    f->ghost_depth = 0;
    return GS_OK;
}

int32_t gs_field_download(gs_ctx *ctx, gs_field *f, float *host)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad argument");
    GS_TRY(sync_all(ctx));
    if (f->rows == 0 || f->cols == 0) return GS_OK; // nothing to copy (host may be null)
    if (!host) return fail(GS_ERR_INVALID, "bad argument");
    const uint64_t first = f->s.front().g_row0;
    for (size_t i = 0; i < f->s.size(); ++i) {
        const FieldSlab &fs = f->s[i];
        GS_HIP(hipSetDevice(ctx->slabs[i].device));
        GS_HIP(hipMemcpy2D(host + (fs.g_row0 - first) * f->cols, (size_t)f->cols * sizeof(float), fs.row0,
                           (size_t)f->pitch * sizeof(float), (size_t)f->cols * sizeof(float), (size_t)fs.rows,
                           hipMemcpyDeviceToHost));
    }
    return GS_OK;
}

int32_t gs_field_device_ptr(const gs_field *f, int32_t slab, void **ptr, uint64_t *pitch, uint64_t *slab_row0,
                            uint64_t *slab_rows, int32_t *device)
{
    if (!f || slab < 0 || (size_t)slab >= f->s.size()) return fail(GS_ERR_INVALID, "bad slab index");
    if (ptr) *ptr = f->s[slab].row0;
    if (pitch) *pitch = (uint64_t)f->pitch;
    if (slab_row0) *slab_row0 = f->s[slab].g_row0;
    if (slab_rows) *slab_rows = (uint64_t)f->s[slab].rows;
    if (device) *device = f->ctx->slabs[slab].device;
    return GS_OK;
}

int32_t gs_field_mark_written(gs_ctx *ctx, gs_field *f)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad handle");
    f->ghost_depth = 0; // as after gs_field_upload: the next step (or gs_field_finalize) refreshes the ghost rows
    return GS_OK;
}

int32_t gs_host_alloc(void **out, uint64_t bytes)
{
    if (!out || bytes == 0) return fail(GS_ERR_INVALID, "bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(GS_ERR_NOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes,
                                     hipGetErrorString(e));
    return GS_OK;
}

int32_t gs_host_free(void *p)
{
    if (!p) return GS_OK;
    GS_HIP(hipHostFree(p));
    return GS_OK;
}

int32_t gs_field_download_async(gs_ctx *ctx, gs_field *f, float *host)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad argument");
    if (f->rows == 0 || f->cols == 0) return GS_OK; // nothing to copy (host may be null)
    if (!host) return fail(GS_ERR_INVALID, "bad argument");
    GS_TRY(resolve_window(ctx)); // (waits for a persistent window launch in flight: its result must be known to be valid)
    const uint64_t first = f->s.front().g_row0;
    const int last = (int)((ctx->step_no + 1) & 1); // parity of the most recent pass
    for (size_t i = 0; i < f->s.size(); ++i) {
        SlabRt &sl = ctx->slabs[i];
        const FieldSlab &fs = f->s[i];
        GS_HIP(hipSetDevice(sl.device));
        const size_t need = (size_t)fs.rows * f->cols;
        if (sl.stage_floats < need) {
            GS_HIP(hipStreamSynchronize(sl.copy));
            if (sl.stage) GS_HIP(hipFree(sl.stage));
            sl.stage = nullptr;
            sl.stage_floats = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&sl.stage), need * sizeof(float));
            if (e != hipSuccess) return fail(GS_ERR_NOMEM, "staging buffer: %s", hipGetErrorString(e));
            sl.stage_floats = need;
        }
        // the previous image must have left the staging buffer; on a slab chain the boundary
        // rows of the newest plane come from the halo stream
        GS_HIP(hipStreamWaitEvent(sl.compute, sl.copied, 0));
        if (ctx->total_slabs() > 1 && ctx->step_no > 0) GS_HIP(hipStreamWaitEvent(sl.compute, sl.halod[last], 0));
        if (i == 0) GS_TRY(join_bands(ctx, sl.compute));
        GS_HIP(hipMemcpy2DAsync(sl.stage, (size_t)f->cols * sizeof(float), fs.row0, (size_t)f->pitch * sizeof(float),
                                (size_t)f->cols * sizeof(float), (size_t)fs.rows, hipMemcpyDeviceToDevice, sl.compute));
        GS_HIP(hipEventRecord(sl.staged, sl.compute));
        GS_HIP(hipStreamWaitEvent(sl.copy, sl.staged, 0));
        GS_HIP(hipMemcpyAsync(host + (fs.g_row0 - first) * f->cols, sl.stage, need * sizeof(float),
                              hipMemcpyDeviceToHost, sl.copy));
        GS_HIP(hipEventRecord(sl.copied, sl.copy));
    }
    return GS_OK;
}

int32_t gs_field_colormap(gs_ctx *ctx, gs_field *f, float scale, const uint8_t *palette_rgb, int32_t n_colors,
                          uint8_t *host_rgb)
{
    if (!ctx || !f || f->ctx != ctx) return fail(GS_ERR_INVALID, "bad argument");
    if (!palette_rgb || n_colors < 1 || n_colors > 65536) return fail(GS_ERR_INVALID, "bad palette (%d colours)", n_colors);
    GS_TRY(sync_all(ctx));
    if (f->rows == 0 || f->cols == 0) return GS_OK; // nothing to paint (host may be null)
    if (!host_rgb) return fail(GS_ERR_INVALID, "bad argument");
    const uint64_t first = f->s.front().g_row0;
    for (size_t i = 0; i < f->s.size(); ++i) {
        SlabRt &sl = ctx->slabs[i];
        const FieldSlab &fs = f->s[i];
        GS_HIP(hipSetDevice(sl.device));
        const size_t bytes = (size_t)fs.rows * f->cols * 3;
        uint8_t *dev = nullptr;
        GS_HIP(hipMalloc(reinterpret_cast<void **>(&dev), bytes + (size_t)n_colors * 3));
        uint8_t *pal = dev + bytes;
        hipError_t e = hipMemcpyAsync(pal, palette_rgb, (size_t)n_colors * 3, hipMemcpyHostToDevice, sl.compute);
        if (e == hipSuccess)
            e = gs_launch_colormap(fs.row0, f->pitch, fs.rows, (int32_t)f->cols, scale, pal, n_colors, dev, sl.compute);
        if (e == hipSuccess)
            e = hipMemcpyAsync(host_rgb + (fs.g_row0 - first) * f->cols * 3, dev, bytes, hipMemcpyDeviceToHost, sl.compute);
        if (e == hipSuccess) e = hipStreamSynchronize(sl.compute);
        (void)hipFree(dev);
        if (e != hipSuccess) return fail(GS_ERR_HIP, "colour mapping failed: %s", hipGetErrorString(e));
    }
    return GS_OK;
}

int32_t gs_download_wait(gs_ctx *ctx)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipStreamSynchronize(sl.copy));
    }
    return GS_OK;
}

} // extern "C"
