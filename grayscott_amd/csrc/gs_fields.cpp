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

// Placement by measurement (gs_hip.h).  hipMalloc's blocks lie in physical regions ("groups") that a process cannot
// see or choose, and two planes of ONE group that are written (and read) in the same pass cost 0.86-0.96 ms per GiB
// pair where two of different groups cost 0.72-0.79 (tools/ubench/hbm_kinds.hip; profiles/r06_placement.md): four planes of
// one group run the single-step kernel at 0.58-0.65 of 8 TB/s, U's planes in one group and V's in another at 0.73-0.76.
// A pass that reads two blocks and writes them back unchanged (gs_launch_pair_probe) tells the two cases apart in 3 ms
// without touching the planes' contents.  So: time the pairs among the planes' four blocks; if each slot's (U, V) pair
// is already as fast as the fastest pair seen -- and a slower pair has been seen, i.e. the fast ones are known to be
// cross-group pairs -- nothing moves.  Else draw blocks one at a time (at most `candidates`), time each against all held,
// and stop as soon as two disjoint fast pairs exist; the planes move (device copies) to the best two pairs.
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
    const size_t n = (size_t)(f0->s[0].rows + 2 * kGhostRows) * pitch + 2 * kGuardFloats; // (a multiple of 64 floats)
    const size_t bytes = n * sizeof(float);
    // (read at every call: the tests switch them between calls of one process)
    const bool trace = gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1) != 0;
    const bool draw_all = gs_env_int("GS_HIP_PLACE_ALL", 0, 0, 1) != 0; // diagnostics: never stop early
    std::vector<float *> blocks;
    for (int i = 0; i < 4; ++i) blocks.push_back(planes[i]->s[0].alloc);
    constexpr float kNotTimed = 1.0e9f;                                // (a pair the deep stage did not time)
    std::vector<std::vector<float>> T(4, std::vector<float>(4, 0.0f)); // pair times, ms
    int probes = 0;
    auto pair_ms = [&](int i, int j, float *ms) -> int32_t {
        float best = 0.0f;
        for (int rep = 0; rep < 3; ++rep) { // the first pass is not timed
            if (rep > 0) GS_HIP(hipEventRecord(sl.t0, sl.compute));
            const hipError_t e = gs_launch_pair_probe(blocks[(size_t)i], blocks[(size_t)j], bytes, sl.compute);
            if (e != hipSuccess) return fail(GS_ERR_HIP, "probe launch failed: %s", hipGetErrorString(e));
            if (rep > 0) {
                float t = 0.0f;
                GS_HIP(hipEventRecord(sl.t1, sl.compute));
                GS_HIP(hipEventSynchronize(sl.t1));
                GS_HIP(hipEventElapsedTime(&t, sl.t0, sl.t1));
                if (best == 0.0f || t < best) best = t;
            }
        }
        ++probes;
        if (trace) std::fprintf(stderr, "gs_hip placement: blocks %2d %2d (%p %p): %.4f ms per pass\n", i, j, (void *)blocks[(size_t)i], (void *)blocks[(size_t)j], best);
        *ms = best;
        return GS_OK;
    };
    auto release_extra = [&](const int (&keep)[4]) {
        for (size_t i = 4; i < blocks.size(); ++i) {
            bool used = false;
            for (int k : keep) used = used || (size_t)k == i;
            if (!used && blocks[i]) (void)hipFree(blocks[i]);
        }
    };
    // the best two disjoint pairs (p0, p1) and (p2, p3) held: the cost of an arrangement is T(U0, V0) + T(U1, V1) --
    // each pass writes one slot's two planes and reads the other's
    int best[4] = {0, 1, 2, 3};
    auto cost_of = [&](const int (&p)[4]) { return T[(size_t)p[0]][(size_t)p[1]] + T[(size_t)p[2]][(size_t)p[3]]; };
    auto search = [&]() {
        const int m = (int)blocks.size();
        float c_best = cost_of(best);
        for (int a = 0; a < m; ++a)
            for (int b = a + 1; b < m; ++b)
                for (int c = a + 1; c < m; ++c)
                    for (int d = c + 1; d < m; ++d) {
                        if (c == b || d == b) continue;
                        const int p[4] = {a, b, c, d};
                        // (an arrangement that moves fewer planes wins a tie of 0.5 %)
                        if (cost_of(p) < 0.995f * c_best) { c_best = cost_of(p); std::memcpy(best, p, sizeof best); }
                    }
    };
    int32_t st = GS_OK;
    for (int i = 0; i < 4 && st == GS_OK; ++i)
        for (int j = i + 1; j < 4 && st == GS_OK; ++j) {
            st = pair_ms(i, j, &T[(size_t)i][(size_t)j]);
            T[(size_t)j][(size_t)i] = T[(size_t)i][(size_t)j];
        }
    if (st != GS_OK) return st;
    const float first_cost = cost_of(best);
    int drawn = 0;
    const bool deep_allowed = gs_env_int("GS_HIP_PLACE_DEEP", 1, 0, 1) != 0;
    const int deep_cap = deep_allowed && bytes >= ((size_t)512 << 20) ? std::min(124, 4 * candidates) : candidates;
    // Test hook (tests/test_gpu_placement.py): GS_HIP_PLACE_FORCE="a,b,c,d" draws `candidates` blocks and then moves the
    // planes to blocks a, b, c, d of those held (0-3: the planes' own, 4 and up: drawn), whatever the probes say -- every
    // shape of move (a plane into a drawn block, chains, two planes swapping, a cycle through all four) on demand.
    int forced[4] = {-1, -1, -1, -1};
    if (const char *e = std::getenv("GS_HIP_PLACE_FORCE")) {
        if (std::sscanf(e, "%d,%d,%d,%d", &forced[0], &forced[1], &forced[2], &forced[3]) != 4) forced[0] = -1;
    }
    while (true) {
        if (forced[0] >= 0) {
            while (drawn < candidates) {
                float *b = nullptr;
                if (hipMalloc(reinterpret_cast<void **>(&b), bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                blocks.push_back(b);
                ++drawn;
            }
            bool ok = true;
            for (int i = 0; i < 4; ++i) {
                ok = ok && forced[i] >= 0 && forced[i] < (int)blocks.size();
                for (int j = 0; j < i; ++j) ok = ok && forced[i] != forced[j];
            }
            if (!ok) { const int keep[4] = {0, 1, 2, 3}; release_extra(keep); return fail(GS_ERR_INVALID, "GS_HIP_PLACE_FORCE: four distinct blocks of those held"); }
            std::memcpy(best, forced, sizeof best);
            T.assign(blocks.size(), std::vector<float>(blocks.size(), T[0][1])); // (no probes: every pair reads alike)
            break;
        }
        search();
        // good enough?  Planes of >= 512 MiB are judged by the rate of the probe pass itself: 4 x bytes per pass at
        // >= 5.25 TB/s is a cross-group pair (5.4-6.0 TB/s measured on every box), less is a pair of one group
        // (4.3-5.0): the spread INSIDE either class (0.72-0.80 ms, 0.86-0.99 ms per GiB pair) is as wide as a small
        // gap between them, so relative criteria alone mistook five blocks of one group for a settled case.  Smaller
        // planes partly live in the 256 MB last-level cache and have no absolute scale: both pairs within 7 % of the
        // fastest pair seen, and a pair at least 14 % slower seen.
        float lo = 0.0f, hi = 0.0f;
        const int m = (int)blocks.size();
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) {
                const float t = T[(size_t)i][(size_t)j];
                if (t >= kNotTimed) continue;
                if (lo == 0.0f || t < lo) lo = t;
                if (t > hi) hi = t;
            }
        const float t01 = T[(size_t)best[0]][(size_t)best[1]], t23 = T[(size_t)best[2]][(size_t)best[3]];
        bool both_fast, contrast;
        if (bytes >= ((size_t)512 << 20)) {
            const float fast_ms = (float)(4.0 * (double)bytes / 5.25e12 * 1e3);
            both_fast = t01 <= fast_ms && t23 <= fast_ms;
            contrast = true;
        } else {
            both_fast = t01 <= 1.07f * lo && t23 <= 1.07f * lo;
            contrast = hi >= 1.14f * lo;
        }
        if (trace)
            std::fprintf(stderr, "gs_hip placement: %d blocks, pairs %.4f ... %.4f ms, best arrangement %d %d | %d %d = %.4f ms per pass%s\n", m, lo,
                         hi, best[0], best[1], best[2], best[3], 0.5f * cost_of(best), both_fast && contrast ? " (settled)" : "");
        if ((both_fast && contrast && !draw_all) || drawn >= deep_cap) break;
        // Beyond `candidates` draws: the DEEP stage.  A fresh box hands out memory in physical order and a region can be
        // tens of GiB long (bench line r06z: 4 + 12 consecutive blocks of one region, single step 0.654 of 8 TB/s).  While
        // more than half of the device's memory is free -- nobody else needs it -- the search goes on, up to 4 x
        // `candidates` blocks, with ONE probe per block (against a block that still lacks a partner); only a block of
        // another region is timed against the rest of the best arrangement.  Large planes only (their probes have an absolute scale).
        const bool deep = drawn >= candidates;
        if (deep) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 2 + bytes) { (void)hipGetLastError(); break; }
        }
        float *b = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&b), bytes) != hipSuccess) { (void)hipGetLastError(); break; } // fewer candidates: fine
        blocks.push_back(b);
        ++drawn;
        for (auto &row : T) row.push_back(kNotTimed);
        T.emplace_back(blocks.size(), kNotTimed);
        const int nb = (int)blocks.size() - 1;
        auto time_pair = [&](int i) {
            if (st != GS_OK) return;
            st = pair_ms(i, nb, &T[(size_t)i][(size_t)nb]);
            T[(size_t)nb][(size_t)i] = T[(size_t)i][(size_t)nb];
        };
        if (!deep) {
            for (int i = 0; i < nb; ++i) time_pair(i);
        } else {
            // the block that still lacks a partner: a member of the slow pair of the best arrangement so far
            const float fast_ms = (float)(4.0 * (double)bytes / 5.25e12 * 1e3);
            const int ref = t01 > fast_ms ? best[0] : best[2];
            time_pair(ref);
            if (st == GS_OK && T[(size_t)ref][(size_t)nb] <= fast_ms) // a block of another region: time it against the whole arrangement
                for (int i = 0; i < 4; ++i)
                    if (best[i] != ref) time_pair(best[i]);
        }
        if (st != GS_OK) { const int keep[4] = {0, 1, 2, 3}; release_extra(keep); return st; }
    }
    // hand the chosen blocks to the planes, with their contents: a plane that stays among the chosen four keeps its
    // block where it can (no copy), the others are copied into the blocks nobody holds
    std::vector<float *> chosen(4);
    for (int i = 0; i < 4; ++i) chosen[(size_t)i] = blocks[(size_t)best[i]];
    // slot pairs are unordered, and so is the order of the two pairs: try the 8 equivalent arrangements, keep the one
    // that leaves most planes where they are
    if (forced[0] < 0) {
        int keep_best = -1;
        std::vector<float *> pick = chosen;
        for (int v = 0; v < 8; ++v) {
            int q[4] = {best[0], best[1], best[2], best[3]};
            if (v & 1) std::swap(q[0], q[1]);
            if (v & 2) std::swap(q[2], q[3]);
            if (v & 4) { std::swap(q[0], q[2]); std::swap(q[1], q[3]); }
            int kept = 0;
            for (int i = 0; i < 4; ++i) kept += q[i] == i;
            if (kept > keep_best) {
                keep_best = kept;
                for (int i = 0; i < 4; ++i) pick[(size_t)i] = blocks[(size_t)q[i]];
            }
        }
        chosen = pick;
    }
    // Moves, one synchronous device copy at a time, the plane's handle updated after each: at every moment (and after a
    // failure) every plane holds its contents in the block its handle names.  A plane moves when nobody holds its
    // chosen block; planes that wait for each other (two planes swapping blocks) go through a spare block.
    hipError_t err = hipSuccess;
    auto holder = [&](float *b) { for (int k = 0; k < 4; ++k) if (planes[k]->s[0].alloc == b) return k; return -1; };
    auto move = [&](int i, float *to) {
        FieldSlab &fs = planes[i]->s[0];
        const hipError_t e = hipMemcpy(to, fs.alloc, bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { err = e; return false; }
        fs.alloc = to;
        fs.row0 = to + kGuardFloats + (size_t)kGhostRows * pitch;
        return true;
    };
    while (err == hipSuccess) {
        bool progress = false, unsettled = false;
        for (int i = 0; i < 4 && err == hipSuccess; ++i) {
            if (planes[i]->s[0].alloc == chosen[(size_t)i]) continue;
            if (holder(chosen[(size_t)i]) < 0) progress = move(i, chosen[(size_t)i]) || progress;
            else unsettled = true;
        }
        if (progress || !unsettled || err != hipSuccess) { if (!progress) break; continue; }
        // a cycle: park one of its planes in a spare block
        float *spare = nullptr;
        for (size_t k = 0; k < blocks.size() && !spare; ++k) {
            bool is_chosen = false;
            for (float *c : chosen) is_chosen = is_chosen || c == blocks[k];
            if (!is_chosen && holder(blocks[k]) < 0) spare = blocks[k];
        }
        if (!spare && hipMalloc(reinterpret_cast<void **>(&spare), bytes) == hipSuccess) blocks.push_back(spare);
        if (!spare) { // no room to swap: the planes of the cycle stay where they are
            (void)hipGetLastError();
            for (int k = 0; k < 4; ++k) chosen[(size_t)k] = planes[k]->s[0].alloc;
            break;
        }
        for (int i = 0; i < 4; ++i)
            if (planes[i]->s[0].alloc != chosen[(size_t)i]) { (void)move(i, spare); break; }
    }
    // release what no plane holds now (drawn blocks that were not chosen, original blocks that were left, a spare)
    for (float *b : blocks)
        if (holder(b) < 0) (void)hipFree(b);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(GS_ERR_HIP, "moving a plane failed (every plane still holds its contents): %s", hipGetErrorString(err));
    }
    (void)hipGetLastError();
    ctx->place_probes += (uint64_t)probes;
    ctx->place_drawn += (uint64_t)drawn;
    if (first_ms) *first_ms = 0.5f * first_cost;
    if (best_ms) *best_ms = 0.5f * cost_of(best);
    return GS_OK;
}

int32_t gs_debug_place_stats(const gs_ctx *ctx, uint64_t *probes, uint64_t *blocks_drawn)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    if (probes) *probes = ctx->place_probes;
    if (blocks_drawn) *blocks_drawn = ctx->place_drawn;
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
    // A persistent window launch in flight may still give up (its workgroups not all resident): the copy is enqueued
    // behind it all the same -- this call never waits -- and the image is validated when it is waited for
    // (gs_download_wait; anything else that waits for results does it too: resolve_window).
    gs_ctx::WindowRt &w = ctx->win;
    if (w.pending) {
        if (!w.seen) {
            GS_HIP(hipSetDevice(ctx->slabs[0].device));
            GS_HIP(hipHostMalloc(reinterpret_cast<void **>(&w.seen), 2 * sizeof(int32_t), hipHostMallocDefault));
            w.seen[0] = w.seen[1] = 0;
        }
        w.images.push_back(gs_ctx::WindowRt::Image{f, host, w.seq});
    }
    const uint64_t first = f->s.front().g_row0;
    const int last = (int)((ctx->step_no + 1) & 1); // parity of the most recent pass
    const int k = (int)(ctx->downloads & 1);        // the staging buffer of this image
    for (size_t i = 0; i < f->s.size(); ++i) {
        SlabRt &sl = ctx->slabs[i];
        const FieldSlab &fs = f->s[i];
        GS_HIP(hipSetDevice(sl.device));
        const size_t need = (size_t)fs.rows * f->cols;
        if (sl.stage_floats[k] < need) {
            GS_HIP(hipStreamSynchronize(sl.image_stream(k)));
            if (sl.stage[k]) GS_HIP(hipFree(sl.stage[k]));
            sl.stage[k] = nullptr;
            sl.stage_floats[k] = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&sl.stage[k]), need * sizeof(float));
            if (e != hipSuccess) return fail(GS_ERR_NOMEM, "staging buffer: %s", hipGetErrorString(e));
            sl.stage_floats[k] = need;
        }
        // the image before the previous one must have left this staging buffer (a wait on the GPU, not on the host: the
        // previous image's host copy goes on meanwhile); on a slab chain the boundary rows of the newest plane come from
        // the halo stream
        GS_HIP(hipStreamWaitEvent(sl.compute, sl.copied[k], 0));
        if (ctx->total_slabs() > 1 && ctx->step_no > 0) GS_HIP(hipStreamWaitEvent(sl.compute, sl.halod[last], 0));
        if (i == 0) GS_TRY(join_bands(ctx, sl.compute));
        GS_HIP(gs_launch_pack_rows(fs.row0, f->pitch, (int32_t)fs.rows, (int32_t)f->cols, sl.stage[k], sl.compute));
        GS_HIP(hipEventRecord(sl.staged, sl.compute));
        // (every other image on a stream of its own: behind one another on ONE stream two host copies leave 13 us of the link
        // unused between them -- an 8.3 MB image every 168 us where the link takes 152, tools/ubench/d2h_probe.hip)
        const hipStream_t cs = sl.image_stream(k);
        GS_HIP(hipStreamWaitEvent(cs, sl.staged, 0));
        GS_HIP(hipMemcpyAsync(host + (fs.g_row0 - first) * f->cols, sl.stage[k], need * sizeof(float),
                              hipMemcpyDeviceToHost, cs));
        if (w.pending && i == 0) // the abort word as it stands once the launches this image depends on have ended
            GS_HIP(hipMemcpyAsync(w.seen + k, w.words + kWindowMaxTiles, sizeof(int32_t), hipMemcpyDeviceToHost, cs));
        GS_HIP(hipEventRecord(sl.copied[k], cs));
    }
    ctx->downloads++;
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

// Wait until at most `in_flight` (0 or 1) of the images enqueued so far are still on their way.  1 is what a driver
// loop with two images in flight calls (the reference's writer thread takes images through a channel two deep,
// simulate/src/main.rs:29-43, 73-87): the host copy of the newest image goes on while the one before it is handed on, so the
// PCIe link never idles between two images.
int32_t gs_download_wait_but(gs_ctx *ctx, int32_t in_flight)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    if (in_flight < 0 || in_flight > 1) return fail(GS_ERR_INVALID, "0 or 1 images may stay in flight, not %d", in_flight);
    if (in_flight == 1 && ctx->downloads < 2) return GS_OK; // nothing older than the newest
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        if (in_flight == 0) {
            GS_HIP(hipStreamSynchronize(sl.copy));
            GS_HIP(hipStreamSynchronize(sl.copy2));
        }
        else GS_HIP(hipEventSynchronize(sl.copied[(ctx->downloads - 2) & 1])); // the image before the newest
    }
    // images enqueued behind persistent window launches: did one of those launches give up?  (Then the launches are run
    // again with the marching kernel and the images fetched again: resolve_window.)  The launches that are still running
    // stay pending; only the images waited for are settled here.
    gs_ctx::WindowRt &w = ctx->win;
    if (!w.images.empty()) {
        if (w.seen && (w.seen[0] != 0 || w.seen[1] != 0)) GS_TRY(resolve_window(ctx)); // (sticky: either word will do)
        if (w.images.size() > (size_t)in_flight) w.images.erase(w.images.begin(), w.images.end() - in_flight);
    }
    return GS_OK;
}

int32_t gs_download_wait(gs_ctx *ctx) { return gs_download_wait_but(ctx, 0); }

} // extern "C"
