// gs_window.cpp -- host side of the persistent window kernel (gs_run_window_k): the tiling of a grid into one round of
// register-resident windows, exchange planes and flags, the launch, and what happens when a launch gives up.
#include "gs_internal.h"

namespace gsi {

std::vector<GsWindowDesc> plan_windows(const gs_ctx *ctx, uint64_t rows, uint64_t cols, int want_rpw, int want_k, int *rpw_out, int *k_out)
{
    // (the launcher builds the cheap kinds of edge window for the default side weights AND dt == 1 only: gs_launch_window)
    const bool cheap = (fast_of(ctx) & 3) == 3 && ctx->o.math == GS_MATH_STRICT && gs_env_int("GS_HIP_EDGE_KINDS", 1, 0, 1) != 0;
    return plan_windows(ctx->cu_count, ctx->o.boundary == GS_BOUNDARY_ZERO_HALO, cheap, rows, cols, want_rpw, want_k, rpw_out, k_out);
}

// (the geometry alone: no device needed -- tests/test_capi_cpu.py checks it through gs_debug_window_plan)
std::vector<GsWindowDesc> plan_windows(int cu_count, bool zero_halo, bool cheap, uint64_t rows, uint64_t cols, int want_rpw, int want_k,
                                       int *rpw_out, int *k_out)
{
    // Steps per exchange when the caller does not say: as many as still leave one window per compute unit.  A step takes the
    // same time whatever the number of windows (a window's waves wait for each other's rows and for their aprons, not for
    // issue slots) and an exchange costs about one step and a half, so the deeper apron pays wherever it fits: 8 rather than 4
    // steps per exchange is +8-10 % on grids of 0.9-1.5 M cells (profiles/r06_window_kernel.md, "smaller grids"); the
    // reference's 1080 x 1920 fits with 4 only, its criterion grid's 1024 x 2048 with 2 only (533 k against the marching kernel's
    // 421 k in 1000-step calls, 453 k against 354 k in 64-step calls).
    if (want_k <= 0) {
        for (int k : {8, 6, 4, 2}) {
            std::vector<GsWindowDesc> plan = plan_windows(cu_count, zero_halo, cheap, rows, cols, want_rpw, k, rpw_out, k_out);
            if (!plan.empty()) return plan;
        }
        return std::vector<GsWindowDesc>();
    }
    std::vector<GsWindowDesc> plan;
    if (cu_count <= 0 || rows == 0 || cols == 0 || rows > 0x7fffff || cols > 0x7fffff) return plan;
    const int k = want_k;
    if (k < 2 || k > 8 || (k & 1)) return plan;
    const long wo = 128 - 2 * k;
    const long tiles_c = (long)((cols + wo - 1) / wo);
    int forced[3] = {0, 0, 0};
    if (const char *e = std::getenv("GS_HIP_WINDOW_WAVES")) (void)std::sscanf(e, "%d,%d,%d", &forced[0], &forced[1], &forced[2]);
    for (int rpw : {5}) { // rows per wave: 80-row windows (gs_launch_window)
        if (want_rpw > 0 && want_rpw != rpw) continue;
        const int min_waves = (2 * k + rpw) / rpw; // at least one owned row
        plan.clear();
        bool ok = true;
        std::vector<long> col_first; // index of the first window of every tile column
        std::vector<int> col_oh;
        for (long tc = 0; tc < tiles_c && ok; ++tc) {
            const bool left = tc == 0, right = tc == tiles_c - 1;
            double cost = 1.0;
            if (left || right) {
                if (zero_halo) cost = 1.05;
                else if (!cheap || (left && right)) cost = 1.6;
                else cost = left ? 1.20 : 1.13;
            }
            // whole rounds of the 4 SIMDs only: 13 waves take the time of 16 (one SIMD holds four of them)
            int waves = cost <= 1.08 ? 16 : 12;
            const int f = left ? forced[0] : (right ? forced[2] : forced[1]);
            if (f > 0) waves = f;
            if (waves > 16) waves = 16;
            if (waves < min_waves) waves = min_waves;
            const int active = waves * rpw, oh = active - 2 * k;
            col_first.push_back((long)plan.size());
            col_oh.push_back(oh);
            for (uint64_t r0 = 0; r0 < rows; r0 += (uint64_t)oh) {
                GsWindowDesc d;
                std::memset(&d, 0, sizeof d);
                d.r0 = (int32_t)r0;
                d.c0 = (int32_t)(tc * wo);
                d.oh = oh;
                d.ow = (int32_t)wo;
                d.active = active;
                plan.push_back(d);
                if ((long)plan.size() > cu_count || plan.size() > (size_t)kWindowMaxTiles) { ok = false; break; }
            }
        }
        if (!ok) continue;
        // neighbours: every window whose owned cells (inside the grid) lie in this window's apron
        for (size_t i = 0; i < plan.size() && ok; ++i) {
            GsWindowDesc &d = plan[i];
            const long tc = d.c0 / wo;
            for (long nc = tc - 1; nc <= tc + 1 && ok; ++nc) {
                if (nc < 0 || nc >= tiles_c) continue;
                const long first = col_first[(size_t)nc], oh = col_oh[(size_t)nc];
                const long last = (nc + 1 < tiles_c ? col_first[(size_t)nc + 1] : (long)plan.size()) - 1;
                // rows [r0 - k, r0 + oh + k) clipped to the grid, in that column's windows
                long lo = (long)d.r0 - k, hi = (long)d.r0 + d.oh + k - 1;
                if (lo < 0) lo = 0;
                if (hi > (long)rows - 1) hi = (long)rows - 1;
                for (long j = first + lo / oh; j <= first + hi / oh && j <= last; ++j) {
                    if ((size_t)j == i) continue;
                    if (d.n_nbr >= kGsWindowMaxNbr) { ok = false; break; }
                    d.nbr[d.n_nbr++] = (int32_t)j;
                }
            }
        }
        if (!ok) continue;
        *rpw_out = rpw;
        *k_out = k;
        return plan;
    }
    plan.clear();
    return plan;
}

// Exchange planes, flags and abort word for planes of this shape (allocated on first use, re-made when the shape changes).
int32_t ensure_window_rt(gs_ctx *ctx, const gs_field *f)
{
    gs_ctx::WindowRt &w = ctx->win;
    GS_HIP(hipSetDevice(ctx->slabs[0].device));
    if (!w.words) {
        GS_HIP(hipMalloc(reinterpret_cast<void **>(&w.words), (kWindowMaxTiles + 1) * sizeof(int32_t)));
        // on the stream the launches use: the context's streams are non-blocking, a hipMemset on the null stream
        // would not be ordered before them (and the words may hold a freed context's flags)
        GS_HIP(hipMemsetAsync(w.words, 0, (kWindowMaxTiles + 1) * sizeof(int32_t), ctx->slabs[0].compute));
        GS_HIP(hipMalloc(reinterpret_cast<void **>(&w.desc), kWindowMaxTiles * sizeof(GsWindowDesc)));
        w.epoch = 0;
    }
    if (w.rows != f->rows || w.pitch != (uint64_t)f->pitch || !w.planes[0]) {
        GS_TRY(sync_all(ctx)); // nothing may still be exchanging through the old planes
        for (auto &p : w.planes) {
            if (p) GS_HIP(hipFree(p));
            p = nullptr;
        }
        // granules {value, exchange number}: 8 bytes per cell (GS_WIN_TAGGED); zeroed: no cell may carry a number by chance
        const size_t bytes = (size_t)(f->rows + 1) * (size_t)f->pitch * 2 * sizeof(float);
        for (auto &p : w.planes) {
            GS_HIP(hipMalloc(reinterpret_cast<void **>(&p), bytes));
            GS_HIP(hipMemsetAsync(p, 0, bytes, ctx->slabs[0].compute));
        }
        w.plane_bytes = bytes;
        w.rows = f->rows;
        w.pitch = (uint64_t)f->pitch;
    }
    if (w.epoch > (1 << 30)) { // keep flag arithmetic far from wrapping: start over behind everything enqueued
        GS_HIP(hipMemsetAsync(w.words, 0, kWindowMaxTiles * sizeof(int32_t), ctx->slabs[0].compute));
        for (auto p : w.planes) GS_HIP(hipMemsetAsync(p, 0, w.plane_bytes, ctx->slabs[0].compute)); // (granules carry exchange numbers)
        w.epoch = 0;
    }
    return GS_OK;
}

// Did a persistent window launch (gs_run_window_k) give up?  Its workgroups poll each other's flags with bounded
// patience; they only run out of it when they are not all resident, i.e. when another long-running kernel holds CUs.
// The abort word then holds the number of the launch that gave up.  Every launch before it ran to its end; that launch
// may have stored some of its windows (workgroups far from the stalled one finish long before the patience runs out),
// but only into its OUTPUT planes, and the launches behind it left at once.  So it and the later ones are run again, in
// order, with the marching kernel -- each from its own input planes, which are intact -- their results put where gs_run
// said they would be, and the context stays with the marching kernel.  Called by everything that waits for or reads
// results.
int32_t resolve_window(gs_ctx *ctx)
{
    gs_ctx::WindowRt &w = ctx->win;
    if (!w.pending) return GS_OK;
    SlabRt &sl = ctx->slabs[0];
    GS_HIP(hipSetDevice(sl.device));
    GS_HIP(hipStreamSynchronize(sl.compute));
    int32_t gave_up = 0;
    GS_HIP(hipMemcpy(&gave_up, w.words + kWindowMaxTiles, sizeof gave_up, hipMemcpyDeviceToHost));
    // (only now: a failure above leaves the launches on record for the next call)
    w.pending = false;
    std::vector<gs_ctx::WindowRt::Launch> launched;
    launched.swap(w.launched);
    std::vector<gs_ctx::WindowRt::Image> images;
    images.swap(w.images);
    if (w.seen) w.seen[0] = w.seen[1] = 0;
    if (!gave_up) return GS_OK;
    if (gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1)) {
        std::vector<int32_t> words((size_t)kWindowMaxTiles);
        if (hipMemcpy(words.data(), w.words, words.size() * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess)
            for (int g = 0; g < kWindowMaxTiles / 2; ++g)
                if (words[(size_t)2 * g] || words[(size_t)2 * g + 1])
                    std::fprintf(stderr, "gs_hip window: launch %d gave up: workgroup %d wave %d at exchange %d after %d polls\n", gave_up, g,
                                 words[(size_t)2 * g] >> 24, words[(size_t)2 * g] & 0xffffff, words[(size_t)2 * g + 1]);
    }
    GS_HIP(hipMemsetAsync(w.words, 0, (kWindowMaxTiles + 1) * sizeof(int32_t), sl.compute));
    w.epoch = 0;
    w.disabled = true;
    w.fallbacks++;
    for (const auto &l : launched) {
        if (l.seq < gave_up) continue; // ran to its end
        // what the launch was counted as when it was enqueued (gs_ctx_stats): the replay counts its own passes
        ctx->launches -= 1;
        ctx->passes -= (uint64_t)l.passes;
        ctx->steps_done -= (uint64_t)l.steps;
        int32_t slot = 0;
        GS_TRY(run_steps(ctx, l.in[0], l.in[1], l.out[0], l.out[1], (uint64_t)l.steps, &slot, false));
        if (slot != 1) { // the marching kernel ends in the slot of the steps' parity: move the planes over
            for (int sp = 0; sp < 2; ++sp) {
                const gs_field *src = l.in[sp];
                gs_field *dst = l.out[sp];
                const size_t bytes = (size_t)src->s[0].rows * (size_t)src->pitch * sizeof(float);
                GS_HIP(hipMemcpyAsync(dst->s[0].row0, src->s[0].row0, bytes, hipMemcpyDeviceToDevice, sl.compute));
                dst->ghost_depth = src->ghost_depth;
            }
        }
        // an image that was enqueued behind this launch (gs_field_download_async) copied what the launch that gave up
        // had left: fetch it again now, before a later launch's replay overwrites the planes
        for (const auto &im : images)
            if (im.after_seq == l.seq) {
                GS_HIP(hipStreamSynchronize(sl.compute));
                GS_HIP(hipStreamSynchronize(sl.copy));
                GS_HIP(hipStreamSynchronize(sl.copy2));
                const FieldSlab &fs = im.f->s[0];
                GS_HIP(hipMemcpy2D(im.host, (size_t)im.f->cols * sizeof(float), fs.row0, (size_t)im.f->pitch * sizeof(float),
                                   (size_t)im.f->cols * sizeof(float), (size_t)fs.rows, hipMemcpyDeviceToHost));
            }
    }
    GS_HIP(hipStreamSynchronize(sl.compute));
    return GS_OK;
}

// gs_run through ONE persistent launch of gs_run_window_k per 2^20 steps (see run_steps).  *launched = 0 when the grid is
// not one round of windows (an error if the kernel was forced).
int32_t run_window(gs_ctx *ctx, Run &r, uint64_t steps, bool forced, int32_t *launched, int32_t *result_slot)
{
    gs_ctx::WindowRt &w = ctx->win;
    gs_field *u0 = r.u[0];
    SlabRt &sl = ctx->slabs[0];
    *launched = 0;
    GS_HIP(hipSetDevice(sl.device));
    // the tiling of this grid (made once per shape and configuration, kept on the device)
    // (kernel = auto only takes grids that 80-row windows cover: with 96-row windows -- 1200 x 2000: 450 k against the
    // marching kernel's 452 k -- nothing is gained, profiles/r04_window_kernel.md)
    const int want_rpw = forced ? (ctx->o.rows_per_block > 0 ? ctx->o.rows_per_block / 16 : 0) : 5;
    const int key = ((ctx->o.boundary * 2 + (ctx->o.math == GS_MATH_FUSED)) * 4 + (fast_of(ctx) & 3)) * 64 + want_rpw * 8 + ctx->o.fuse_steps;
    if (w.plan_rows != u0->rows || w.plan_cols != u0->cols || w.plan_key != key) {
        int rpw = 0, wk = 0;
        const std::vector<GsWindowDesc> plan = plan_windows(ctx, u0->rows, u0->cols, want_rpw, ctx->o.fuse_steps, &rpw, &wk);
        GS_TRY(sync_all(ctx)); // no launch may still be reading the old tiling (or its flags)
        w.plan_rows = u0->rows; w.plan_cols = u0->cols; w.plan_key = key;
        w.plan_rpw = rpw; w.plan_k = wk; w.plan_n = (int)plan.size(); // (0: remembered as "not this grid")
        if (!plan.empty()) {
            GS_TRY(ensure_window_rt(ctx, u0));
            GS_HIP(hipMemcpy(w.desc, plan.data(), plan.size() * sizeof(GsWindowDesc), hipMemcpyHostToDevice));
            // the flags belong to the workgroups of the old tiling: start over (the exchange numbers with them: no granule
            // of the old tiling may be taken for one of the new)
            GS_HIP(hipMemsetAsync(w.words, 0, kWindowMaxTiles * sizeof(int32_t), sl.compute));
            for (auto p : w.planes) GS_HIP(hipMemsetAsync(p, 0, w.plane_bytes, sl.compute));
            w.epoch = 0;
        }
    }
    if (w.plan_n == 0) {
        if (forced)
            return fail(GS_ERR_UNSUPPORTED, "GS_KERNEL_WINDOW needs a grid of at most one window per compute unit (%d); "
                                            "%llu x %llu cells do not fit", ctx->cu_count, (unsigned long long)u0->rows,
                        (unsigned long long)u0->cols);
        return GS_OK;
    }
    GS_TRY(join_bands(ctx, sl.compute));
    ctx->bands_active = false;
    GS_TRY(ensure_window_rt(ctx, u0));
    if (w.seq >= 0x7ffffff0) { // launch numbers only order the launches pending at one time: start over behind them
        GS_TRY(sync_all(ctx));
        w.seq = 0;
    }
    // the waits above may have found that an earlier launch gave up: the context then stays with the marching kernel
    if (w.disabled) return forced ? fail(GS_ERR_UNSUPPORTED, "the persistent window kernel gave up on this context before (another "
                                                             "kernel held compute units): it stays with GS_KERNEL_TB") : GS_OK;
    uint64_t left = steps;
    int slot = 0;
    while (left > 0) { // (the step count is an int in the kernel; a launch goes in-planes -> out-planes)
        const int n = left > (1u << 20) ? (1 << 20) : (int)left;
        GsStepArgs a = make_args(ctx, r.u[slot], r.v[slot], r.u[1 - slot], r.v[1 - slot], 0, 1);
        GsWindowArgs x;
        std::memset(&x, 0, sizeof x);
        x.xu[0] = w.planes[0]; x.xu[1] = w.planes[1];
        x.xv[0] = w.planes[2]; x.xv[1] = w.planes[3];
        x.flags = w.words;
        x.abort = w.words + kWindowMaxTiles;
        x.desc = w.desc;
        x.n_windows = w.plan_n;
        x.steps = n;
        x.k = w.plan_k;
        x.epoch = w.epoch;
        x.patience = gs_env_int("GS_HIP_WINDOW_PATIENCE", 1 << 20, 1, 1 << 30); // polls of 2-3 us each: 2-3 s
        x.seq = ++w.seq;
        const char *name = nullptr;
        const hipError_t e = ctx->o.math == GS_MATH_FUSED ? gs_launch_window_fused(a, x, w.plan_rpw, sl.compute, &name)
                                                           : gs_launch_window_strict(a, x, w.plan_rpw, sl.compute, &name);
        if (e != hipSuccess) return fail(GS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        const int supers = (n + w.plan_k - 1) / w.plan_k;
        w.epoch += supers;
        w.pending = true;
        w.launched.push_back(gs_ctx::WindowRt::Launch{{r.u[slot], r.v[slot]}, {r.u[1 - slot], r.v[1 - slot]}, n, x.seq, supers});
        ctx->last_kernel = name;
        ctx->launches++;
        ctx->passes += (uint64_t)supers;
        ctx->steps_done += (uint64_t)n;
        ctx->step_no++;
        slot ^= 1;
        left -= (uint64_t)n;
    }
    if (result_slot) *result_slot = slot;
    *launched = 1;
    return GS_OK;
}

} // namespace gsi

using namespace gsi;

extern "C" {

int32_t gs_debug_window_plan(uint64_t rows, uint64_t cols, int32_t compute_units, int32_t boundary, int32_t cheap_edge_kinds,
                             int32_t window_rows, int32_t k, int32_t *out, int32_t cap_windows, int32_t *rows_per_wave, int32_t *k_out)
{
    int rpw = 0, kk = 0;
    const std::vector<GsWindowDesc> plan = plan_windows(compute_units, boundary == GS_BOUNDARY_ZERO_HALO, cheap_edge_kinds != 0, rows, cols,
                                                        window_rows > 0 ? window_rows / 16 : 0, k, &rpw, &kk);
    if (rows_per_wave) *rows_per_wave = rpw;
    if (k_out) *k_out = kk;
    constexpr int words = (int)(sizeof(GsWindowDesc) / sizeof(int32_t));
    if (out)
        for (size_t i = 0; i < plan.size() && (int)i < cap_windows; ++i) std::memcpy(out + i * words, &plan[i], sizeof(GsWindowDesc));
    return (int32_t)plan.size();
}

} // extern "C"
