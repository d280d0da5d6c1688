// gs_tuner.cpp -- which configuration of the temporally blocked kernel a launch uses: parameter specialisation
// (fast_of), the launch-geometry model (unit heights that make a launch a whole number of rounds of the chip's wave
// slots), the on-line tuner of gs_run (timed passes of the simulation itself), and the choices handed between contexts.
#include "gs_internal.h"

namespace gsi {

// Output columns per wave of the temporally blocked kernel (gs_march.h: tb_cols_per_wave).
long tb_strips(int32_t cols, int fuse, int cpl)
{
    const long w = (64 - 2 * ((fuse + cpl - 1) / cpl)) * (long)cpl;
    return (cols + w - 1) / w;
}

// The form of difference sharing in force (when the parameters allow it): 0 = none, 1 = within a lane (the halo-board
// march), 2 = also across lanes.  Pinned by gs_options.share_taps (1 / 2 / 3 = within / none / across), else what the
// on-line tuner last chose or is trying (gs_ctx::share_now), else 2 (kShareDefault).
int share_mode(const gs_ctx *ctx)
{
    switch (ctx->o.share_taps) {
    case 1: return 1;
    case 2: return 0;
    case 3: return 2;
    default: return ctx->share_now;
    }
}

// GsStepArgs::fast for this context's parameters: bit 0 = the four side weights are 0.5, bit 1 = dt == 1, bit 2 = both
// and the diagonal weights are pairwise equal and the context wants full difference sharing, bit 3 = also across lanes.
// ... as the parameters alone decide it: bit 2 = full difference sharing is possible (cells_vshare: the diagonal taps of a
// row pair are each other's negatives when w00 == w22 and w02 == w20)
static int fast_possible(const gs_ctx *ctx)
{
    int fast = 0;
    if (!ctx->o.general_kernels) {
        const float(*w)[3] = ctx->p.w;
        if (w[0][1] == 0.5f && w[1][0] == 0.5f && w[1][2] == 0.5f && w[2][1] == 0.5f) fast |= 1;
        if (ctx->p.dt == 1.0f) fast |= 2;
        if (fast == 3 && w[0][0] == w[2][2] && w[0][2] == w[2][0] && ctx->o.math == GS_MATH_STRICT) fast |= 4;
    }
    return fast;
}
int fast_of(const gs_ctx *ctx)
{
    const int fast = fast_possible(ctx), mode = share_mode(ctx);
    if (!(fast & 4) || mode == 0) return fast & 3;
    return mode == 2 ? fast | 8 : fast;
}

// Unit heights that make a launch of the temporally blocked kernel exactly `r` rounds of the chip's wave
// slots (256 CUs x 4 SIMDs x the kernel entry's waves per SIMD): strips x chunks <= r x slots with the
// chunks as short as that allows.  A launch that misses such a height by one chunk runs a nearly empty
// extra round: at 4096^2 with 2 columns per lane 36 rows give 749 k Mcells x steps/s, 32 rows 677 k, 40
// rows 687 k (profiles/archive/r02_sweeps.md, section 9).  From two rounds up the launcher tapers the last two
// rounds (an eighth and a half as tall: 0.625 rounds' worth of rows), which the formula accounts for.
// Writes up to `max` heights (the single-round one first); returns their number.
int fit_heights(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl, int fast, int *out, int max, bool partial)
{
    const int slots = ctx->o.math == GS_MATH_FUSED ? gs_tb_wave_slots_fused(fuse, fast, cpl) : gs_tb_wave_slots_strict(fuse, fast, cpl);
    const long strips = tb_strips(cols, fuse, cpl);
    if (slots <= 0 || strips <= 0) return 0;
    const long per_round = slots / strips; // chunks per round
    if (per_round < 1) return 0;
    const long per_round_up = (slots + strips - 1) / strips; // what the launcher tapers (gs_launch_tb)
    int n = 0;
    auto push = [&](long h) {
        if (h > rows) h = rows;
        for (int i = 0; i < n; ++i)
            if (out[i] == (int)h) return;
        if (n < max) out[n++] = (int)h;
    };
    // A launch of at most one round dispatches its edge units -- up to 3 strips of every chunk, all strips of
    // the top and bottom chunk rows -- as two halves each (gs_launch_tb): count them.  `partial`: also the
    // heights that leave every SIMD w = waves - 1, ..., 1 waves instead of a full round (1080 x 1920, 1 column
    // per lane, 5 waves per SIMD: 8 rows fill the round, 10 rows give every SIMD 4 waves and are 6 % faster).
    const int waves = slots / 1024;
    const long wcols = (64 - 2 * ((fuse + cpl - 1) / cpl)) * (long)cpl, scols = ((fuse + cpl - 1) / cpl) * (long)cpl;
    const long ne = strips <= 1 ? strips : (((strips - 1) * wcols + scols >= cols && strips >= 2) ? 3 : 2); // edge strips (gs_step_tb_k)
    for (int w = waves; w >= (partial ? 1 : waves); --w) {
        // units = chunks x (strips + ne) + 2 x (strips - ne): the halves of the edge strips of every chunk and
        // of the other strips of the top and bottom chunk rows
        const long chunks = strips <= ne ? 1024L * w / (2 * strips) : (1024L * w - 2 * (strips - ne)) / (strips + ne);
        if (chunks < 1) continue;
        const long h = (rows + chunks - 1) / chunks;
        if (h >= 2) push(h);
    }
    // r = 2: an un-tapered launch of two full rounds; r >= 3: (r - 1) full rounds + the two tapered ones (the
    // launcher tapers from two rounds' worth of full-height units up)
    for (int r = 2; r <= 8 && n < max; ++r) {
        const double chunks = r < 3 ? (double)(per_round * r) : (double)(per_round * (r - 1)) + 0.625 * (double)per_round_up;
        const long h = (long)std::ceil((double)rows / chunks - 1e-9);
        if (h < 2L * fuse) break;
        push(h);
    }
    return n;
}

// `rows` = rows of one slab.  Slabs of an uneven partition differ by one row: same configuration.
bool tuned_for(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse)
{
    const uint64_t r = (uint64_t)rows;
    return ctx->tuned_rpu > 0 && ctx->tuned_k == fuse && ctx->tuned_cols == (uint64_t)cols &&
           (ctx->tuned_rows == r || (ctx->total_slabs() > 1 && (ctx->tuned_rows == r + 1 || ctx->tuned_rows + 1 == r)));
}

// Columns per lane of the temporally blocked kernel when nothing was tuned on line: 2 (measured
// fastest from 4096^2 up, profiles/archive/r01_sweeps.md runs 54-57) unless that cannot give every SIMD a
// wave at a unit height of 8 * fuse rows, then 1.
int32_t pick_cols_per_lane(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse)
{
    if (ctx->o.cols_per_lane > 0) return ctx->o.cols_per_lane;
    if (tuned_for(ctx, rows, cols, fuse) && ctx->tuned_cpl > 0) return ctx->tuned_cpl;
    if (fuse < 2) return 2;
    return (long)rows * tb_strips(cols, fuse, 2) / (8L * fuse) >= 2048 ? 2 : 1;
}

int32_t pick_rows_per_unit(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse)
{
    if (ctx->o.rows_per_block > 0) return ctx->o.rows_per_block;
    if (tuned_for(ctx, rows, cols, fuse)) return ctx->tuned_rpu;
    const int cpl = pick_cols_per_lane(ctx, rows, cols, fuse);
    const int32_t own = model_rows_per_unit(ctx, rows, cols, fuse, cpl);
    // Several slabs of one process on ONE device share its wave slots: their launches run side by side and
    // together fill many rounds.  Where a slab's own height is its ONE-round height (the slab alone does not
    // fill two rounds), the height follows the rows the device holds instead.  16384^2 as N slabs on one GPU,
    // own / device-wide height: 8 slabs (76 / 122 rows) 865-885 k / 1000-1030 k = 0.94-0.98 of the single slab,
    // 4 slabs (152 / 122) 935-970 k / 1007-1015 k; 2 slabs keep their own 142 rows = two rounds each: 1019-1077 k
    // against 1000-1008 k with 122 (profiles/archive/r03_sweeps.md, section 5).
    if (ctx->slabs.size() > 1 && fuse > 1) {
        int fit[2];
        const int nf = fit_heights(ctx, rows, cols, fuse, cpl, fast_of(ctx), fit, 2);
        if (nf > 0 && own == fit[0]) {
            int same = 0;
            for (const auto &sl : ctx->slabs) same += sl.device == ctx->slabs[0].device;
            int64_t rows_on_device = (int64_t)rows * same;
            if (rows_on_device > 0x7fffffff) rows_on_device = 0x7fffffff;
            const int32_t h = model_rows_per_unit(ctx, (int32_t)rows_on_device, cols, fuse, cpl);
            return h > rows ? (rows > 0 ? rows : 1) : h;
        }
    }
    return own;
}

// ... for a given lane layout, from the launch geometry alone (also the tuner's first candidate).
int32_t model_rows_per_unit(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl)
{
    const long strips = fuse > 1 ? tb_strips(cols, fuse, cpl) : (cols + 255) / 256;
    const long want = fuse > 1 ? 32L * fuse : 16;
    if (fuse > 1) {
        // A launch of a whole number of rounds of the chip's wave slots (fit_heights): of the heights of
        // at least 4K rows (at most a third of a unit's rows recomputed) the one nearest to 32K, else the
        // single-round height.
        int fit[8];
        const int nf = fit_heights(ctx, rows, cols, fuse, cpl, fast_of(ctx), fit, 8);
        long best = 0;
        for (int i = 0; i < nf; ++i)
            if (fit[i] >= 4 * fuse && (best == 0 || std::labs(fit[i] - want) < std::labs(best - want))) best = fit[i];
        if (best == 0 && nf > 0 && fit[0] >= 2 * fuse) best = fit[0];
        if (best > 0) return (int32_t)best;
    }
    long rpu = ((long)rows * strips + 16383) / 16384; // keep >= 16384 waves per launch when possible
    if (rpu > want) rpu = want;
    // small grids are bound by the length of a wave's march: units of K rows there (runs 120-122)
    const long least = (long)rows * cols <= (1L << 19) ? fuse : 2L * fuse;
    if (rpu < least) rpu = least;
    if (rpu < 4) rpu = 4;
    return (int32_t)rpu;
}

// Window shape and steps per launch of the LDS-window kernel (gs_run_tile_k) for a grid, from a cost model
// fitted to the measured launches (profiles/archive/r02_sweeps.md, section 10): a launch costs T0 = 3.4 / 2.6 / 3.7 us
// (launch gap, weights, window load and store) plus K steps of 0.74 / 0.585 / 1.38 us for the 32 / 16 / 64-row
// window while every workgroup has a CU to itself; beyond 256 workgroups they run in rounds (two share a CU
// at 0.87 of the time of two turns).  The model is within ~15 % of the measured rates from 64 x 128 to 1024 x
// 1024 and picks the measured-best or second-best configuration on every grid of that table.
void pick_tile_config(long rows, long cols, int *shape, int *k)
{
    static const int window_rows[3] = {32, 16, 64};
    static const double launch_us[3] = {3.4, 2.6, 3.7}, step_us[3] = {0.74, 0.585, 1.38};
    static const int ks[3] = {4, 6, 8};
    double best = 0.0;
    for (int sh = 0; sh < 3; ++sh)
        for (int kk : ks) {
            if (2 * kk >= window_rows[sh]) continue;
            const long ho = window_rows[sh] - 2 * kk, wo = 64 - 2 * kk;
            const long wgs = ((rows + ho - 1) / ho) * ((cols + wo - 1) / wo);
            const double rounds = wgs <= 256 ? 1.0 : 0.87 * (double)((wgs + 255) / 256);
            const double us_per_step = (launch_us[sh] + kk * step_us[sh] * rounds) / kk;
            if (best == 0.0 || us_per_step < best) { best = us_per_step; *shape = sh; *k = kk; }
        }
}

// Rows of the slabs of `f` as the tuning tables key them (the first local slab's; the others differ by
// at most one row).
uint64_t slab_rows_of(const gs_field *f) { return f->s.empty() ? f->rows : (uint64_t)f->s.front().rows; }

// Same slab shape as far as tuning goes: the slabs of an uneven partition differ by one row and must
// all run the same configuration (the steps per pass above all: the exchange is that many rows deep).
bool same_slab_shape(const gs_ctx *ctx, uint64_t rows_a, uint64_t cols_a, uint64_t rows_b, uint64_t cols_b)
{
    if (cols_a != cols_b) return false;
    return rows_a == rows_b || (ctx->total_slabs() > 1 && (rows_a + 1 == rows_b || rows_b + 1 == rows_a));
}

bool tuned_shape(const gs_ctx *ctx, const gs_field *f, int fuse)
{
    return ctx->tuned_rpu > 0 && ctx->tuned_fuse == fuse &&
           same_slab_shape(ctx, ctx->tuned_rows, ctx->tuned_cols, slab_rows_of(f), f->cols);
}

// Make the remembered choice for this shape (if any) the active one.
void recall_tuned(gs_ctx *ctx, const gs_field *f, int fuse)
{
    if (tuned_shape(ctx, f, fuse)) return;
    const uint64_t rows = slab_rows_of(f);
    for (const gs_ctx::Tuned &t : ctx->tuned_cache)
        if (same_slab_shape(ctx, t.rows, t.cols, rows, f->cols) && t.fuse == fuse) {
            ctx->tuned_rows = t.rows; ctx->tuned_cols = t.cols; ctx->tuned_fuse = t.fuse;
            ctx->tuned_rpu = t.rpu; ctx->tuned_split = t.split; ctx->tuned_k = t.k; ctx->tuned_cpl = t.cpl;
            ctx->tuned_share = t.share;
            ctx->share_now = t.share;
            return;
        }
    // nothing chosen for this shape yet: its untuned passes and its tuning start from the default form, not from the
    // form another shape of this context was tuned to (share_now is the context's, the choices are per shape)
    ctx->share_now = kShareDefault;
}

void remember_tuned(gs_ctx *ctx, const gs_ctx::Tuned &t)
{
    for (auto it = ctx->tuned_cache.begin(); it != ctx->tuned_cache.end(); ++it)
        if (it->rows == t.rows && it->cols == t.cols && it->fuse == t.fuse) { ctx->tuned_cache.erase(it); break; }
    if (ctx->tuned_cache.size() >= 64) ctx->tuned_cache.erase(ctx->tuned_cache.begin());
    ctx->tuned_cache.push_back(t);
    if (ctx->tuned_rows == t.rows && ctx->tuned_cols == t.cols && ctx->tuned_fuse == t.fuse) ctx->tuned_rpu = 0; // re-recall
}

// On-line choice of unit height, fused steps per pass and columns per lane (single slab, fused
// passes, unit height not pinned).  The best values depend on how a launch tiles the chip (tail
// effects vs 2K redundant rows per unit vs occupancy), so the first passes of a run on a new shape
// are timed with a few candidates -- they are real passes of the simulation, nothing is recomputed
// -- and the fastest combination is kept for this context and shape.  Continues in the next gs_run
// when this one is too short.
//   phase A: unit heights; (phase B, band counts: retired, see bands_for;) phase C: fewer fused
//   steps per pass (when fuse_steps is not pinned) -- on small, cache-resident grids the 2K
//   redundant rows per unit can cost more than the extra passes; phase D (columns per lane not
//   pinned): 1 and 4 columns per lane -- more, narrower waves for small grids; fewer, wider ones
//   with 16-byte accesses -- with a few unit heights each (large grids skip the candidates that
//   would only multiply tiny units); phase E (gs_options.share_taps = 0 and the parameters allow it): the
//   chosen configuration without difference sharing -- A-D run with it (the default form: across lanes too).  A-C run with the untuned layout
//   (pick_cols_per_lane).  Every list of heights is a fixed ladder plus the heights that make a launch a whole
//   number of rounds of the chip's wave slots (fit_heights).
int32_t tune_online(Run &r, int fuse)
{
    gs_ctx *ctx = r.ctx;
    const gs_field *f = r.u[0];
    static const int cand0[] = {2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192};
    static const int altk[] = {3, 2};
    static const int candn0[] = {2, 4, 8, 16, 32, 64, 128};
    const uint64_t cells = f->rows * f->cols;
    const bool large = cells > (1ull << 26);
    const int user_cpl = ctx->o.cols_per_lane;
    // phases A-C run with the layout an untuned context would use (1 column per lane on small grids, 2
    // from about 1024 x 2048 up), so that what a run is given before its tuning has finished is already
    // close; phase D tries the other two layouts
    const int base_cpl = user_cpl ? user_cpl : ((long)f->rows * tb_strips((int32_t)f->cols, fuse, 2) / (8L * fuse) >= 2048 ? 2 : 1);
    const int cpls[2] = {base_cpl == 1 ? 2 : 1, base_cpl == 4 ? 2 : 4};
    const int fast = fast_of(ctx);
    // a fixed ladder of heights plus the heights that make a launch a whole number of rounds of the
    // chip's wave slots (fit_heights), in ascending order without near-duplicates
    auto heights = [&](const int *fixed, int nfixed, int k, int cpl) {
        std::vector<int> v(fixed, fixed + nfixed);
        int fit[16];
        // (not on large grids: from about six rounds per launch up the rate is flat in the unit height --
        // 16384^2: 0.2469-0.2511 ms per step from 96 to 214 rows -- and every candidate costs passes)
        const int nf = large ? 0 : fit_heights(ctx, (int32_t)f->rows, (int32_t)f->cols, k, cpl, fast, fit, 16, true);
        for (int i = 0; i < nf; ++i) {
            bool dup = false;
            for (int x : v) dup = dup || x == fit[i];
            if (!dup) v.push_back(fit[i]);
        }
        std::sort(v.begin(), v.end());
        return v;
    };
    std::vector<int> cand = heights(cand0, (int)(sizeof cand0 / sizeof cand0[0]), fuse, base_cpl);
    // The height an untuned context would use (the model's: a whole number of rounds, ~32K rows) is timed
    // last in phase A, when the chip has warmed up, and is preferred on large grids unless it is 2 % slower
    // than the best of the ladder (see the margins in evaluate()).
    const int dflt = model_rows_per_unit(ctx, (int32_t)f->rows, (int32_t)f->cols, fuse, base_cpl);
    cand.erase(std::remove(cand.begin(), cand.end(), dflt), cand.end());
    cand.push_back(dflt);
    // ... and, with 2 columns per lane, where it is the height of ONE round of 16-wave workgroups kept in step
    // (gs_launch_tb): that form is 6-11 % ahead of every other height of such a grid, but two-pass windows of
    // 20-80 us passes scatter by more than that while the tuning runs inside short calls (the criterion grid's
    // 16-step calls left 2048 x 4096 on 24-row units, 627 k, where long calls find 19-20 rows, 730 k).
    int one_round[1] = {0};
    const bool prefer_model = large || (base_cpl == 2 && fuse == 4 &&
                                        fit_heights(ctx, (int32_t)f->rows, (int32_t)f->cols, fuse, base_cpl, fast, one_round, 1) == 1 &&
                                        one_round[0] == dflt);
    const int ncand = (int)cand.size();
    const int nalt = 0; // phase B is empty
    const int nk = ctx->o.fuse_steps == 0 ? (int)(sizeof altk / sizeof altk[0]) : 0;
    // phase D: (columns per lane, height) pairs.  (Fitted for `fuse` steps per pass whatever phase C kept:
    // the list must not change while the phases advance, and the strips of 3 and 4 fused steps are the
    // same width for 2 and 4 columns per lane, 58 against 56 columns for 1.)
    std::vector<int> d_cpl, d_rpu;
    if (user_cpl == 0)
        for (int c : cpls)
            for (int h : heights(candn0, (int)(sizeof candn0 / sizeof candn0[0]), fuse, c)) { d_cpl.push_back(c); d_rpu.push_back(h); }
    const int nn = (int)d_cpl.size();
    // phase E: one candidate (no sharing), where the choice is open and a variant with full difference sharing exists at
    // all.  Sharing within a lane only (share_taps = 1) is not a candidate: against the default form it is within 1 % where
    // it wins and 3-5 % behind on every developed pattern once the power cap has set the clock -- which a timing window of
    // a few passes on a chip that was idle a moment ago does not show (at 4096^2 the windows preferred it on every input
    // and the run then lost 5 %, profiles/r05_cross_lane.md, section 3).
    const bool share_open = ctx->o.share_taps == 0 && (fast_possible(ctx) & 4) != 0;
    const int ne = share_open ? 1 : 0;
    // timed passes per candidate: short passes need more of them for a stable comparison
    const int reps = cells >= (1ull << 27) ? 2 : (cells >= (1ull << 24) ? 6 : 8);
    constexpr int kMaxBatch = (int)(sizeof(gs_ctx::Tuning::batch) / sizeof(gs_ctx::Trial));
    // A call with at least this many passes still to come is a batch job: it waits for each phase's
    // windows, so a long first run is tuned when it returns.  Shorter calls -- a driver loop with a
    // few passes per image -- never wait: their windows are read by a later gs_run.
    constexpr uint64_t kWaitPasses = 16;

    gs_ctx::Tuning *tu = nullptr;
    for (auto &t : ctx->tunings)
        if (t.rows == f->rows && t.cols == f->cols && t.fuse == fuse) tu = &t;
    if (!tu) {
        if (ctx->tunings.size() >= 16) { // the oldest unfinished tuning makes room
            for (auto e : ctx->tunings.front().events)
                if (e) (void)hipEventDestroy(e);
            ctx->tunings.erase(ctx->tunings.begin());
        }
        ctx->tunings.emplace_back();
        tu = &ctx->tunings.back();
        tu->rows = f->rows;
        tu->cols = f->cols;
        tu->fuse = fuse;
    }
    const int V0 = bands_for(ctx, f, fuse);
    SlabRt &sl = ctx->slabs[0];
    GS_HIP(hipSetDevice(sl.device));
    // Candidates of one phase do not depend on each other, so a whole phase is enqueued back to
    // back -- per candidate: [an untimed pass when the kernel changes,] event, `reps` passes,
    // event, `reps` passes, event -- and read once: no idle gaps (clock ramps) between the timing
    // windows.  A candidate's time is the shorter of its two windows.
    // Timestamp "everything enqueued so far has finished" without holding anything back: after a
    // banded pass the event is recorded on the copy stream, which is made to wait for the bands (a
    // record on the compute stream would turn every window boundary into a barrier between passes,
    // and hide exactly the overlap that bands are for).
    auto mark = [&](hipEvent_t ev) -> int32_t {
        hipStream_t ts = sl.compute;
        if (ctx->bands_active) {
            ts = sl.copy;
            GS_TRY(join_bands(ctx, ts));
        }
        GS_HIP(hipEventRecord(ev, ts));
        return GS_OK;
    };
    if (tu->events.empty()) {
        tu->events.resize(3 * kMaxBatch, nullptr);
        for (auto &e : tu->events) GS_HIP(hipEventCreate(&e));
    }
    // read the windows of the batch in flight
    auto evaluate = [&]() -> int32_t {
        for (int b = 0; b < tu->nb; ++b) {
            const gs_ctx::Trial &t = tu->batch[b];
            float w0 = 0.f, w1 = 0.f;
            if (hipEventElapsedTime(&w0, tu->events[3 * b], tu->events[3 * b + 1]) != hipSuccess ||
                hipEventElapsedTime(&w1, tu->events[3 * b + 1], tu->events[3 * b + 2]) != hipSuccess)
                return fail(GS_ERR_HIP, "timing a tuning pass failed");
            const float ms = (w0 < w1 ? w0 : w1) / (float)(t.reps * t.k); // per time step
            static const bool trace = gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1) != 0;
            if (trace)
                std::fprintf(stderr, "gs_hip tuner %llux%llu: unit %3d rows, %d band(s), %d steps/pass, %d col/lane%s: "
                                     "%.4f ms/step (windows %.3f %.3f ms)\n",
                             (unsigned long long)f->rows, (unsigned long long)f->cols, t.rpu, t.V, t.k, t.cpl,
                             t.share == 2 ? "" : (t.share ? ", taps shared within lanes only" : ", taps not shared"), ms, w0, w1);
            // prefer the incumbent unless the newcomer is clearly faster: by 1 %, or by 3 % when it
            // fuses fewer steps (more HBM traffic, slower remainder passes: a tie is not worth it)
            // ... and a taller unit of the same layout wins a near-tie: it recomputes fewer rows, and on large
            // grids the rate is flat over a wide range of heights, where a 1 % margin would keep the first
            // (shortest) height of the plateau's edge
            const bool taller = t.k == tu->best_k && t.cpl == tu->best_cpl && t.V == tu->best_split && t.rpu > tu->best_rpu;
            // On large grids (a plateau from 96 to 214 rows at 16384^2, windows of two passes that scatter by
            // 1-2 %, more while the chip warms up) picking inside the plateau by such measurements is a lottery
            // (64 or 256 rows, 2-3 % below the plateau, in two of six runs): the model's height, timed last
            // in phase A, wins unless it is 2 % slower than the best of the ladder.
            float margin = t.k < tu->best_k ? 0.97f : (taller ? 0.998f : 0.99f);
            if (prefer_model && t.rpu == dflt && t.cpl == base_cpl && t.k == fuse) margin = 1.02f;
            if (tu->best_rpu == 0 || ms < margin * tu->best_ms) {
                tu->best_ms = ms;
                tu->best_rpu = t.rpu;
                tu->best_split = t.V;
                tu->best_k = t.k;
                tu->best_cpl = t.cpl;
                tu->best_share = t.share;
            }
        }
        tu->nb = 0;
        return GS_OK;
    };
    if (tu->nb > 0) { // windows of an earlier call
        const hipError_t q = hipEventQuery(tu->events[3 * (tu->nb - 1) + 2]);
        if (q == hipErrorNotReady) return GS_OK; // still running: this call runs the incumbent
        if (q != hipSuccess) return fail(GS_ERR_HIP, "a tuning pass failed: %s", hipGetErrorString(q));
        GS_TRY(evaluate());
    }
    const int phase_end[5] = {ncand, ncand + nalt, ncand + nalt + nk, ncand + nalt + nk + nn, ncand + nalt + nk + nn + ne};
    constexpr int kLast = 4;
    int warm_cpl = 0, warm_k = 0, warm_share = kShareDefault; // kernel of the newest pass enqueued by this call
    bool out_of_steps = false;
    // The first milliseconds of work on an idle chip run slow (the first windows of a 16384^2 context measured
    // 0.32 ms per step against 0.255 a few passes later: clocks, first touches), which used to cost whichever
    // candidate was timed first its chance.  A tuning therefore starts with ~20 ms of untimed passes (real
    // passes of the run, like all the others) in the model's configuration.
    if (tu->next == 0 && tu->best_rpu == 0 && tu->nb == 0) {
        uint64_t want = 2500000000ull / (cells ? cells : 1); // ~20 ms at 500 k Mcells x steps / s
        if (want < 8) want = 8;
        if (want > 2000) want = 2000;
        const uint64_t have = (r.steps - r.n) / (uint64_t)fuse;
        const uint64_t n = have > 4 * want ? want : have / 4;
        ctx->o.cols_per_lane = base_cpl;
        int32_t st = GS_OK;
        for (uint64_t i = 0; i < n && st == GS_OK; ++i) st = r.advance(V0, fuse);
        ctx->o.cols_per_lane = user_cpl;
        if (st != GS_OK) return st;
        warm_cpl = base_cpl;
        warm_k = fuse;
    }
    while (tu->next < phase_end[kLast] && !out_of_steps) {
        int phase = 0;
        while (tu->next >= phase_end[phase]) ++phase;
        int nb = 0;
        int32_t st = GS_OK;
        for (; tu->next < phase_end[phase] && nb < kMaxBatch && st == GS_OK; ++tu->next) {
            gs_ctx::Trial t{0, V0, fuse, base_cpl, reps, kShareDefault};
            const int i = tu->next - (phase ? phase_end[phase - 1] : 0);
            if (phase == 0) {
                t.rpu = cand[i];
                // units shorter than 2K rows recompute more rows than they produce: only worth it
                // where a pass is latency-bound, i.e. on small grids
                if ((t.rpu < 2 * fuse && cells > (1ull << 21)) || (uint64_t)t.rpu > f->rows || (large && t.rpu < 32)) continue;
            } else if (phase == 2) {
                t.rpu = tu->best_rpu;
                t.V = tu->best_split;
                t.k = altk[i];
                if (t.rpu == 0 || t.k >= fuse) continue;
            } else if (phase == 4) { // what phases A-D chose, without difference sharing
                t.rpu = tu->best_rpu;
                t.V = tu->best_split;
                t.k = tu->best_k;
                t.cpl = tu->best_cpl;
                t.share = 0;
                // (only 2 columns per lane and 2 to 4 fused steps have a sharing variant: elsewhere nothing to compare)
                if (t.rpu == 0 || t.cpl != 2 || t.k < 2) continue;
            } else { // phase 3 (phase 1 has no candidates)
                t.cpl = d_cpl[i];
                t.rpu = d_rpu[i];
                t.V = tu->best_split;
                t.k = tu->best_k;
                if (tu->best_rpu == 0 || (t.rpu < 2 * t.k && cells > (1ull << 21)) || (uint64_t)t.rpu > f->rows ||
                    (large && (t.cpl == 1 || t.rpu < 32)))
                    continue;
            }
            // short calls get shorter windows rather than no tuning at all, but not shorter than two
            // passes per window: single-pass windows are noise, and a mis-tuned configuration is worse
            // than the untuned default (criterion grid, 16-step calls: profiles/archive/r02_criterion_grid.md).
            // With less than 5 passes left the candidate waits for the next gs_run.
            const uint64_t passes_left = (r.steps - r.n) / (uint64_t)t.k;
            while (t.reps > 2 && passes_left < (uint64_t)(2 * t.reps + 1)) --t.reps;
            if (passes_left < (uint64_t)(2 * t.reps + 1)) {
                out_of_steps = true;
                break;
            }
            ctx->o.rows_per_block = t.rpu;
            ctx->o.cols_per_lane = t.cpl;
            ctx->share_now = t.share;
            if (t.cpl != warm_cpl || t.k != warm_k || t.share != warm_share) { // another kernel: one untimed pass first
                st = r.advance(t.V, t.k);
                warm_cpl = t.cpl;
                warm_k = t.k;
                warm_share = t.share;
            }
            for (int w = 0; w < 3 && st == GS_OK; ++w) {
                st = mark(tu->events[3 * nb + w]);
                for (int p = 0; p < t.reps && w < 2 && st == GS_OK; ++p) st = r.advance(t.V, t.k);
            }
            ctx->o.rows_per_block = 0;
            ctx->o.cols_per_lane = user_cpl;
            ctx->share_now = kShareDefault;
            tu->batch[nb++] = t;
        }
        if (st != GS_OK) return st;
        tu->nb = nb;
        if (nb == 0) continue;
        if ((r.steps - r.n) / (uint64_t)fuse < kWaitPasses) break; // short call: read them next time
        if (hipEventSynchronize(tu->events[3 * (nb - 1) + 2]) != hipSuccess)
            return fail(GS_ERR_HIP, "waiting for the tuning passes failed");
        GS_TRY(evaluate());
    }
    if (tu->next >= phase_end[kLast] && tu->nb == 0 && tu->best_rpu > 0) {
        const gs_ctx::Tuned done{f->rows, f->cols, fuse, tu->best_rpu, tu->best_split, tu->best_k, tu->best_cpl, tu->best_share};
        remember_tuned(ctx, done);
        recall_tuned(ctx, f, fuse);
        if (gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1))
            std::fprintf(stderr, "gs_hip tuner %llux%llu: chose unit %d rows, %d steps/pass, %d col/lane, taps %s\n",
                         (unsigned long long)f->rows, (unsigned long long)f->cols, done.rpu, done.k, done.cpl,
                         done.share == 2 ? "shared, across lanes too" : (done.share ? "shared within lanes" : "not shared"));
        for (auto e : tu->events)
            if (e) (void)hipEventDestroy(e);
        ctx->tunings.erase(ctx->tunings.begin() + (tu - ctx->tunings.data()));
    }
    return GS_OK;
}

} // namespace gsi

using namespace gsi;

extern "C" {

int32_t gs_ctx_get_tuned(const gs_ctx *ctx, uint64_t slab_rows, uint64_t cols, int32_t *rows_per_block,
                         int32_t *fuse_steps, int32_t *cols_per_lane, int32_t *share_taps)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    int rpu = 0, k = 0, cpl = 0, share = 0;
    for (const gs_ctx::Tuned &t : ctx->tuned_cache)
        if (t.rows == slab_rows && t.cols == cols) { // the newest entry wins
            rpu = t.rpu; k = t.k; cpl = t.cpl;
            // the EFFECTIVE form: only 2 columns per lane with 2 to 4 fused steps, strict math and a stencil whose diagonal
            // weights pair up have a sharing variant -- everything else runs without, whatever the entry carries
            const bool has_variant = t.cpl == 2 && t.k >= 2 && (fast_possible(ctx) & 4);
            share = !has_variant ? 2 : (t.share == 1 ? 1 : (t.share ? 3 : 2));
        }
    if (rows_per_block) *rows_per_block = rpu;
    if (fuse_steps) *fuse_steps = k;
    if (cols_per_lane) *cols_per_lane = cpl;
    if (share_taps) *share_taps = share;
    return GS_OK;
}

int32_t gs_ctx_set_tuned(gs_ctx *ctx, uint64_t slab_rows, uint64_t cols, int32_t rows_per_block, int32_t fuse_steps,
                         int32_t cols_per_lane, int32_t share_taps)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    if (rows_per_block < 1 || fuse_steps < 1 || fuse_steps > kGhostRows ||
        (cols_per_lane != 1 && cols_per_lane != 2 && cols_per_lane != 4) || share_taps < 0 || share_taps > 3)
        return fail(GS_ERR_INVALID, "bad configuration (unit %d rows, %d steps per pass, %d columns per lane, share_taps %d)",
                    rows_per_block, fuse_steps, cols_per_lane, share_taps);
    // keyed like gs_run's own choices: by the steps per pass it was asked to fuse
    const int fuse = ctx->o.fuse_steps > 0 ? (ctx->o.fuse_steps > kGhostRows ? kGhostRows : ctx->o.fuse_steps) : kGhostRows;
    if (fuse_steps > fuse) return fail(GS_ERR_INVALID, "%d steps per pass exceed fuse_steps = %d", fuse_steps, fuse);
    remember_tuned(ctx, gs_ctx::Tuned{slab_rows, cols, fuse, rows_per_block, 1, fuse_steps, cols_per_lane, share_taps == 2 ? 0 : (share_taps == 1 ? 1 : (share_taps == 3 ? 2 : kShareDefault))});
    return GS_OK;
}

} // extern "C"
