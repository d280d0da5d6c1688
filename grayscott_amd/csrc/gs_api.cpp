// gs_api.cpp -- contexts, the schedule of a pass, gs_step / gs_run: the core of the C ABI in include/gs_hip.h (planes:
// gs_fields.cpp, kernel configuration: gs_tuner.cpp, the window kernel's host side: gs_window.cpp, RCCL: gs_rccl.cpp).
//
// Host-side runtime of the backend: contexts (devices, streams, row partition, RCCL
// communicator), planes (row slabs with ghost rows in HBM), the per-step launch and
// ghost-row exchange schedule, and the small amount of plumbing the reference's
// Concentration contract needs (fill, fill_slice, finalize, upload, download).
//
// A "pass" advances the state by K <= 4 time steps with one sweep over the planes (K = 1 for
// gs_step).  Schedule of one pass on a chain of S > 1 slabs (per slab i; p = parity of the pass
// counter; the same schedule runs on the opt-in row bands of a single slab, without the copies):
//
//   halo stream (high priority)                     compute stream
//   ---------------------------                     --------------
//   wait done[p^1][i], halo[p^1][local nbrs]        wait halo[p^1][i]
//   kernel: rows [0,K) and [rows-K,rows) -> out     kernel: rows [K, rows-K)  -> out
//   out rows [0,K)       -> upper nbr's out ghost   record done[p][i]
//   out rows [rows-K,..) -> lower nbr's out ghost
//     (same process: device-to-device copy; other process: ncclSend / ncclRecv pair)
//   record halo[p][i]
//
// so the exchange of pass n overlaps the interior update of pass n, and pass n+1's interior only
// waits for its own slab's boundary rows.  Every dependency is an event on the consumer's stream;
// the host never blocks inside gs_step / gs_run (except while gs_run's on-line tuner reads the
// timings of a phase of candidate configurations, a few times per context and shape).
#include "gs_internal.h"

namespace gsi {

bool is_pow2_or_zero(float w)
{
    if (w == 0.0f) return true;
    int e = 0;
    const float m = std::frexp(std::fabs(w), &e);
    return m == 0.5f;
}

int32_t check_math(const gs_params &p, int32_t math)
{
    if (math != GS_MATH_STRICT && math != GS_MATH_FUSED)
        return fail(GS_ERR_INVALID, "unknown math flavour %d", math);
    if (math == GS_MATH_FUSED)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                if (!is_pow2_or_zero(p.w[i][j]))
                    return fail(GS_ERR_UNSUPPORTED,
                                "GS_MATH_FUSED needs stencil weights that are 0 or a power of two "
                                "(w[%d][%d] = %g); use GS_MATH_STRICT", i, j, (double)p.w[i][j]);
    return GS_OK;
}

int32_t same_shape(const gs_field *a, const gs_field *b)
{
    if (a->rows != b->rows || a->cols != b->cols || a->pitch != b->pitch || a->ctx != b->ctx)
        return fail(GS_ERR_INVALID, "fields of one step must share context and shape "
                                    "([%llu,%llu] vs [%llu,%llu])",
                    (unsigned long long)a->rows, (unsigned long long)a->cols,
                    (unsigned long long)b->rows, (unsigned long long)b->cols);
    return GS_OK;
}

int32_t sync_all(gs_ctx *ctx)
{
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipStreamSynchronize(sl.halo));
        GS_HIP(hipStreamSynchronize(sl.compute));
        GS_HIP(hipStreamSynchronize(sl.copy));
        GS_HIP(hipStreamSynchronize(sl.copy2));
    }
    for (auto &b : ctx->bands) {
        GS_HIP(hipSetDevice(b.device));
        GS_HIP(hipStreamSynchronize(b.halo));
        GS_HIP(hipStreamSynchronize(b.compute));
    }
    GS_TRY(resolve_window(ctx));
    return GS_OK;
}

int32_t copy_row(gs_ctx *ctx, int src_slab, const float *src, int dst_slab, float *dst, size_t bytes,
                 hipStream_t stream)
{
    const int sd = ctx->slabs[src_slab].device, dd = ctx->slabs[dst_slab].device;
    if (sd == dd)
        GS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    else
        GS_HIP(hipMemcpyPeerAsync(dst, dd, src, sd, bytes, stream));
    return GS_OK;
}

// Push the `depth` boundary rows of `planes` to the ghost rows of the neighbouring slabs,
// from local slab i, on `stream`.  Rows travel whole (pitch floats) and the `depth` rows of a
// side are contiguous, so every plane needs one message per direction.
int32_t push_halo(gs_ctx *ctx, gs_field *const *planes, int nplanes, int i, hipStream_t stream, int depth)
{
    const int n_local = (int)ctx->slabs.size();
    const int k = ctx->global_index(i), S = ctx->total_slabs();
    const bool up_remote = (k > 0) && (i == 0);
    const bool down_remote = (k < S - 1) && (i == n_local - 1);
    for (int f = 0; f < nplanes; ++f) {
        gs_field *pl = planes[f];
        const FieldSlab &me = pl->s[i];
        const size_t bytes = (size_t)depth * pl->pitch * sizeof(float);
        if (i > 0) { // my first rows -> bottom ghost rows of the slab above
            const FieldSlab &nb = pl->s[i - 1];
            GS_TRY(copy_row(ctx, i, me.row0, i - 1, nb.row0 + (ptrdiff_t)nb.rows * pl->pitch, bytes, stream));
        }
        if (i < n_local - 1) { // my last rows -> top ghost rows of the slab below
            const FieldSlab &nb = pl->s[i + 1];
            GS_TRY(copy_row(ctx, i, me.row0 + (ptrdiff_t)(me.rows - depth) * pl->pitch, i + 1,
                            nb.row0 - (ptrdiff_t)depth * pl->pitch, bytes, stream));
        }
    }
    if (up_remote || down_remote) {
        Rccl *R = rccl();
        if (!R || !ctx->comm) return fail(GS_ERR_RCCL, "RCCL communicator missing");
        GS_NCCL(R, R->GroupStart());
        for (int f = 0; f < nplanes; ++f) {
            gs_field *pl = planes[f];
            const FieldSlab &me = pl->s[i];
            const size_t n = (size_t)depth * pl->pitch;
            if (up_remote) {
                GS_NCCL(R, R->Send(me.row0, n, ncclFloat, ctx->rank - 1, ctx->comm, stream));
                GS_NCCL(R, R->Recv(me.row0 - (ptrdiff_t)depth * pl->pitch, n, ncclFloat, ctx->rank - 1, ctx->comm,
                                   stream));
            }
            if (down_remote) {
                GS_NCCL(R, R->Send(me.row0 + (ptrdiff_t)(me.rows - depth) * pl->pitch, n, ncclFloat, ctx->rank + 1,
                                   ctx->comm, stream));
                GS_NCCL(R, R->Recv(me.row0 + (ptrdiff_t)me.rows * pl->pitch, n, ncclFloat, ctx->rank + 1, ctx->comm,
                                   stream));
            }
        }
        GS_NCCL(R, R->GroupEnd());
    }
    return GS_OK;
}

// Smallest slab of the row partition of `f` (every process computes the same value).
int min_slab_rows(const gs_ctx *ctx, const gs_field *f)
{
    return (int)(f->rows / (uint64_t)ctx->total_slabs() < 0x7fffffffull ? f->rows / (uint64_t)ctx->total_slabs()
                                                                          : 0x7fffffffull);
}

// Bring the ghost rows of one plane up to date (after fill / fill_slice / upload).
int32_t refresh_ghosts(gs_ctx *ctx, gs_field *f)
{
    if (ctx->total_slabs() > 1) {
        ctx->ghost_refreshes++;
        GS_TRY(sync_all(ctx));
        gs_field *planes[1] = {f};
        const int depth = min_slab_rows(ctx, f) < kGhostRows ? min_slab_rows(ctx, f) : kGhostRows;
        for (int i = 0; i < (int)ctx->slabs.size(); ++i) {
            GS_HIP(hipSetDevice(ctx->slabs[i].device));
            GS_TRY(push_halo(ctx, planes, 1, i, ctx->slabs[i].halo, depth));
        }
        GS_TRY(sync_all(ctx));
        f->ghost_depth = depth; // what was exchanged: min(4, smallest slab)
        return GS_OK;
    }
    f->ghost_depth = kGhostRows;
    return GS_OK;
}

int32_t launch_rows(gs_ctx *ctx, const GsStepArgs &a, hipStream_t stream, int fuse)
{
    int32_t kernel = ctx->o.kernel;
    if (kernel == GS_KERNEL_AUTO || kernel == GS_KERNEL_TILE || kernel == GS_KERNEL_WINDOW) kernel = fuse > 1 ? GS_KERNEL_TB : GS_KERNEL_STREAM;
    if (fuse > 1 && kernel != GS_KERNEL_TB)
        return fail(GS_ERR_UNSUPPORTED, "only the temporally blocked kernel fuses steps");
    const bool fused = ctx->o.math == GS_MATH_FUSED;
    const char *name = nullptr;
    hipError_t e;
    switch (kernel) {
    case GS_KERNEL_TB:
        e = fused ? gs_launch_tb_fused(a, fuse, stream, &name) : gs_launch_tb_strict(a, fuse, stream, &name);
        break;
    case GS_KERNEL_SIMPLE:
        e = fused ? gs_launch_simple_fused(a, stream, &name) : gs_launch_simple_strict(a, stream, &name);
        break;
    case GS_KERNEL_STREAM:
        e = fused ? gs_launch_stream_fused(a, stream, &name) : gs_launch_stream_strict(a, stream, &name);
        break;
    case GS_KERNEL_LDS:
        e = fused ? gs_launch_lds_fused(a, stream, &name) : gs_launch_lds_strict(a, stream, &name);
        break;
    default:
        return fail(GS_ERR_UNSUPPORTED, "kernel variant %d is not built", kernel);
    }
    if (e != hipSuccess) return fail(GS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    ctx->last_kernel = name;
    ctx->launches++;
    static const bool trace = gs_env_int("GS_HIP_TRACE_LAUNCH", 0, 0, 1) != 0;
    static int traced = 0;
    if (trace && traced < 64 && ++traced)
        std::fprintf(stderr, "gs_hip launch %s: slab of %d rows x %d cols, rows [%d, %d) + [%d, %d), unit %d rows, %d col/lane, %d step(s)\n",
                     name, a.rows, a.cols, a.ra0, a.ra1, a.rb0, a.rb1, a.rows_per_unit, a.cpl, fuse);
    return GS_OK;
}

GsStepArgs make_args(const gs_ctx *ctx, const gs_field *in_u, const gs_field *in_v,
                     const gs_field *out_u, const gs_field *out_v, int i, int fuse)
{
    GsStepArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_u = in_u->s[i].row0;
    a.in_v = in_v->s[i].row0;
    a.out_u = out_u->s[i].row0;
    a.out_v = out_v->s[i].row0;
    a.rows = in_u->s[i].rows;
    a.cols = (int32_t)in_u->cols;
    a.pitch = in_u->pitch;
    const int k = ctx->global_index(i);
    a.top_present = k > 0;
    a.bottom_present = k < ctx->total_slabs() - 1;
    a.ghost = kGhostRows;
    a.rows_per_unit = pick_rows_per_unit(ctx, a.rows, a.cols, fuse);
    a.cpl = pick_cols_per_lane(ctx, a.rows, a.cols, fuse);
    a.allow_fair = ctx->total_slabs() == 1;
    static const bool edge_kinds = gs_env_int("GS_HIP_EDGE_KINDS", 1, 0, 1) != 0;
    a.edge_kinds = edge_kinds;
    a.zero_halo = ctx->o.boundary == GS_BOUNDARY_ZERO_HALO;
    std::memcpy(a.w, ctx->p.w, sizeof a.w);
    a.du = ctx->p.du;
    a.dv = ctx->p.dv;
    a.feed = ctx->p.feed;
    // (feed_rate + kill_rate) is an f32 add in the reference (compute/naive/src/lib.rs:77);
    // both operands are normal numbers, so forming it here in f32 gives the same bits.
    a.feed_plus_kill = ctx->p.feed + ctx->p.kill;
    a.dt = ctx->p.dt;
    a.fast = fast_of(ctx);
    return a;
}

// ---- in-place row bands of a single slab -------------------------------------------------
// One slab can be scheduled as V row bands that alias the same planes: a band's "ghost rows" are
// simply the neighbouring band's rows, so nothing is copied, but the dependency structure is
// that of a slab chain: the next pass of a band only waits for its own previous pass and for the
// K boundary rows of its neighbours.  The tail of pass n (waves draining at different times)
// then overlaps the head of pass n+1 instead of idling the chip between dependent launches.
int32_t ensure_bands(gs_ctx *ctx, int V)
{
    if ((int)ctx->bands.size() >= V) return GS_OK;
    const int device = ctx->slabs[0].device;
    GS_HIP(hipSetDevice(device));
    int least = 0, greatest = 0;
    GS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (!ctx->band_join) GS_HIP(hipEventCreateWithFlags(&ctx->band_join, hipEventDisableTiming));
    while ((int)ctx->bands.size() < V) {
        SlabRt b;
        b.device = device;
        GS_HIP(hipStreamCreateWithPriority(&b.compute, hipStreamNonBlocking, least));
        GS_HIP(hipStreamCreateWithPriority(&b.halo, hipStreamNonBlocking, greatest));
        for (int k = 0; k < 2; ++k) {
            GS_HIP(hipEventCreateWithFlags(&b.done[k], hipEventDisableTiming));
            GS_HIP(hipEventCreateWithFlags(&b.halod[k], hipEventDisableTiming));
        }
        ctx->bands.push_back(b);
    }
    return GS_OK;
}

// Make `stream` wait for everything the band streams were given (no-op when they are idle).
int32_t join_bands(gs_ctx *ctx, hipStream_t stream)
{
    if (!ctx->bands_active) return GS_OK;
    for (auto &b : ctx->bands)
        for (int k = 0; k < 2; ++k) {
            GS_HIP(hipStreamWaitEvent(stream, b.done[k], 0));
            GS_HIP(hipStreamWaitEvent(stream, b.halod[k], 0));
        }
    return GS_OK;
}

int bands_for(const gs_ctx *ctx, const gs_field *f, int fuse)
{
    if (ctx->total_slabs() != 1 || fuse < 2) return 1;
    // Opt-in (gs_options.split >= 2).  Bands reach +3 % on the 16384^2 grid on a good day, but how
    // the four streams of two bands share the chip varies from box to box and run to run (845 k to
    // 936 k for one configuration, profiles/archive/r01_sweeps.md runs 58-61); one launch per pass does not.
    int V = ctx->o.split;
    if (V == 0) V = 1;
    if (V > 8) V = 8;
    while (V > 1 && f->rows / (uint64_t)V < (uint64_t)(8 * fuse)) --V;
    return V < 1 ? 1 : V;
}

int32_t step_bands(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v, int fuse, int V)
{
    GS_TRY(ensure_bands(ctx, V));
    SlabRt &sl = ctx->slabs[0];
    GS_HIP(hipSetDevice(sl.device));
    const GsStepArgs full = make_args(ctx, in_u, in_v, out_u, out_v, 0, fuse);
    const int n = full.rows;
    // The band-to-band events below order consecutive passes of ONE layout: the same row ranges
    // and the same number of fused steps (a band waits for its neighbours' K-row boundary kernels
    // only, and its interior kernel overwrites everything but its own K boundary rows).  Any other
    // sequence starts behind a full barrier.
    if (ctx->bands_active && (V != ctx->bands_v || n != ctx->bands_rows || fuse != ctx->bands_k))
        GS_TRY(join_bands(ctx, sl.compute));
    ctx->bands_v = V;
    ctx->bands_rows = n;
    ctx->bands_k = fuse;
    // whatever was enqueued on the slab's own streams (fills, single steps, staging copies)
    GS_HIP(hipEventRecord(ctx->band_join, sl.compute));
    const int p = (int)(ctx->step_no & 1), q = p ^ 1;
    for (int k = 0; k < V; ++k) {
        SlabRt &b = ctx->bands[k];
        const int r0 = (int)((int64_t)k * n / V), r1 = (int)((int64_t)(k + 1) * n / V);
        GsStepArgs a = full;
        a.allow_fair = 0; // several launches share the chip
        const ptrdiff_t off = (ptrdiff_t)r0 * full.pitch;
        a.in_u += off; a.in_v += off; a.out_u += off; a.out_v += off;
        a.rows = r1 - r0;
        a.top_present = k > 0;
        a.bottom_present = k < V - 1;
        const int nk = a.rows;
        // band-edge rows first (high priority), so that the neighbours' next pass can start
        GS_HIP(hipStreamWaitEvent(b.halo, ctx->band_join, 0));
        GS_HIP(hipStreamWaitEvent(b.halo, b.done[q], 0));
        if (k > 0) GS_HIP(hipStreamWaitEvent(b.halo, ctx->bands[k - 1].halod[q], 0));
        if (k < V - 1) GS_HIP(hipStreamWaitEvent(b.halo, ctx->bands[k + 1].halod[q], 0));
        GsStepArgs e = a;
        e.ra0 = 0;
        e.ra1 = nk <= 2 * fuse ? nk : fuse;
        e.rb0 = nk <= 2 * fuse ? 0 : nk - fuse;
        e.rb1 = nk <= 2 * fuse ? 0 : nk;
        e.rows_per_unit = fuse;
        GS_TRY(launch_rows(ctx, e, b.halo, fuse));
        GS_HIP(hipEventRecord(b.halod[p], b.halo));
        GS_HIP(hipStreamWaitEvent(b.compute, ctx->band_join, 0));
        GS_HIP(hipStreamWaitEvent(b.compute, b.halod[q], 0));
        if (nk > 2 * fuse) {
            a.ra0 = fuse;
            a.ra1 = nk - fuse;
            GS_TRY(launch_rows(ctx, a, b.compute, fuse));
        }
        GS_HIP(hipEventRecord(b.done[p], b.compute));
    }
    ctx->bands_active = true;
    ctx->step_no++;
    ctx->passes++;
    ctx->steps_done += (uint64_t)fuse;
    out_u->ghost_depth = fuse;
    out_v->ghost_depth = fuse;
    return GS_OK;
}

// Advances the state by `fuse` time steps with ONE pass over the planes.  On a chain of slabs
// the ghost rows are `fuse` deep for that pass: the boundary kernel updates the first and last
// `fuse` rows, which are then pushed to the neighbours while the interior kernel runs.
int32_t step_impl(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v,
                  int fuse)
{
    const int n_local = (int)ctx->slabs.size();
    const int S = ctx->total_slabs();
    if (in_u->rows == 0 || in_u->cols == 0) { // empty grid: every step is a no-op
        ctx->step_no++;
        return GS_OK;
    }
    if (S == 1) {
        SlabRt &sl = ctx->slabs[0];
        GS_HIP(hipSetDevice(sl.device));
        GS_TRY(join_bands(ctx, sl.compute));
        ctx->bands_active = false;
        GsStepArgs a = make_args(ctx, in_u, in_v, out_u, out_v, 0, fuse);
        a.ra0 = 0;
        a.ra1 = a.rows;
        GS_TRY(launch_rows(ctx, a, sl.compute, fuse));
    } else {
        if (fuse > kGhostRows || fuse > min_slab_rows(ctx, in_u))
            return fail(GS_ERR_INVALID, "cannot fuse %d steps over slabs of %d rows", fuse, min_slab_rows(ctx, in_u));
        // The input planes need `fuse` valid ghost rows (after fill / upload, or after a pass that
        // fused fewer steps, they are refreshed from the neighbours first: a blocking exchange).
        if (in_u->ghost_depth < fuse) GS_TRY(refresh_ghosts(ctx, in_u));
        if (in_v->ghost_depth < fuse) GS_TRY(refresh_ghosts(ctx, in_v));
        const int p = (int)(ctx->step_no & 1), q = p ^ 1;
        gs_field *outs[2] = {out_u, out_v};
        // Rows per boundary band and per exchange: as deep as the ghost rows go (4, or the smallest slab),
        // whatever this pass fuses -- so the planes it leaves behind serve a pass of any depth, and a short
        // pass (a remainder, a single gs_step) is never followed by a blocking refresh.
        const int depth = min_slab_rows(ctx, in_u) < kGhostRows ? min_slab_rows(ctx, in_u) : kGhostRows;
        for (int i = 0; i < n_local; ++i) {
            SlabRt &sl = ctx->slabs[i];
            GS_HIP(hipSetDevice(sl.device));
            GsStepArgs a = make_args(ctx, in_u, in_v, out_u, out_v, i, fuse);
            const int n = a.rows;
            // gs_ctx_set_pass_timing: events around this pass's halo-stream work and interior kernel
            const bool timed = sl.timed < ctx->pass_timing;
            if (timed && (int)sl.th0.size() <= sl.timed) {
                hipEvent_t ev[4];
                for (auto &e : ev) GS_HIP(hipEventCreate(&e));
                sl.th0.push_back(ev[0]); sl.th1.push_back(ev[1]); sl.tc0.push_back(ev[2]); sl.tc1.push_back(ev[3]);
            }
            // halo stream: boundary rows, then the exchange
            GS_HIP(hipStreamWaitEvent(sl.halo, sl.done[q], 0));
            if (i > 0) GS_HIP(hipStreamWaitEvent(sl.halo, ctx->slabs[i - 1].halod[q], 0));
            if (i < n_local - 1) GS_HIP(hipStreamWaitEvent(sl.halo, ctx->slabs[i + 1].halod[q], 0));
            if (timed) GS_HIP(hipEventRecord(sl.th0[sl.timed], sl.halo));
            GsStepArgs b = a;
            b.ra0 = 0;
            b.ra1 = n <= 2 * depth ? n : depth;
            b.rb0 = n <= 2 * depth ? 0 : n - depth;
            b.rb1 = n <= 2 * depth ? 0 : n;
            b.rows_per_unit = depth; // one unit per boundary band and strip
            GS_TRY(launch_rows(ctx, b, sl.halo, fuse));
            GS_TRY(push_halo(ctx, outs, 2, i, sl.halo, depth));
            GS_HIP(hipEventRecord(sl.halod[p], sl.halo));
            if (timed) GS_HIP(hipEventRecord(sl.th1[sl.timed], sl.halo));
            // compute stream: interior rows
            GS_HIP(hipStreamWaitEvent(sl.compute, sl.halod[q], 0));
            if (timed) GS_HIP(hipEventRecord(sl.tc0[sl.timed], sl.compute));
            if (n > 2 * depth) {
                a.ra0 = depth;
                a.ra1 = n - depth;
                GS_TRY(launch_rows(ctx, a, sl.compute, fuse));
            }
            GS_HIP(hipEventRecord(sl.done[p], sl.compute));
            if (timed) {
                GS_HIP(hipEventRecord(sl.tc1[sl.timed], sl.compute));
                sl.timed++;
            }
        }
    }
    ctx->step_no++;
    ctx->passes++;
    ctx->steps_done += (uint64_t)fuse;
    // the depth actually exchanged: a chain pushed min(4, smallest slab) rows (fuse never exceeds that)
    const int left = S > 1 ? (min_slab_rows(ctx, in_u) < kGhostRows ? min_slab_rows(ctx, in_u) : kGhostRows) : fuse;
    out_u->ghost_depth = left;
    out_v->ghost_depth = left;
    return GS_OK;
}

int32_t check_step_fields(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v)
{
    if (!ctx || !in_u || !in_v || !out_u || !out_v) return fail(GS_ERR_INVALID, "null handle");
    if (in_u->ctx != ctx) return fail(GS_ERR_INVALID, "field belongs to another context");
    GS_TRY(same_shape(in_u, in_v));
    GS_TRY(same_shape(in_u, out_u));
    GS_TRY(same_shape(in_u, out_v));
    if (in_u == out_u || in_v == out_v || in_u == in_v || out_u == out_v || in_u == out_v || in_v == out_u)
        return fail(GS_ERR_INVALID, "the four planes of a step must be distinct");
    return GS_OK;
}

// hipGraph replay (gs_options.use_graph): passes of kk fused steps are captured in batches of
// kGraphBatch -- an even number, so a batch ends on the planes it started from and can be replayed
// as is -- and each batch costs one hipGraphLaunch instead of kGraphBatch kernel launches on the
// host side.  The captured launches carry plane addresses, tuning and parameters: the key holds
// all of them and a mismatch rebuilds the graph.
int32_t replay_graph_batches(Run &r, int kk)
{
    constexpr int kGraphBatch = 16;
    gs_ctx *ctx = r.ctx;
    if ((r.steps - r.n) / (uint64_t)kk < (uint64_t)kGraphBatch) return GS_OK;
    SlabRt &sl = ctx->slabs[0];
    GS_HIP(hipSetDevice(sl.device));
    GS_TRY(join_bands(ctx, sl.compute));
    ctx->bands_active = false;
    const gs_field *f = r.u[0];
    gs_ctx::GraphKey key;
    key.planes[0] = r.u[r.in]->s[0].row0; key.planes[1] = r.v[r.in]->s[0].row0;
    key.planes[2] = r.u[1 - r.in]->s[0].row0; key.planes[3] = r.v[1 - r.in]->s[0].row0;
    key.rows = f->rows; key.cols = f->cols;
    key.k = kk;
    key.rpu = pick_rows_per_unit(ctx, (int32_t)f->rows, (int32_t)f->cols, kk);
    key.cpl = pick_cols_per_lane(ctx, (int32_t)f->rows, (int32_t)f->cols, kk);
    key.batch = kGraphBatch;
    key.p = ctx->p;
    if (!ctx->graph_exec || !(ctx->graph_key == key)) {
        if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
        if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
        const uint64_t n0 = r.n, step0 = ctx->step_no, launches0 = ctx->launches, passes0 = ctx->passes, sd0 = ctx->steps_done;
        const int in0 = r.in;
        GS_HIP(hipStreamBeginCapture(sl.compute, hipStreamCaptureModeThreadLocal));
        int32_t st = GS_OK;
        for (int b = 0; b < kGraphBatch && st == GS_OK; ++b) st = r.advance(1, kk);
        const hipError_t e = hipStreamEndCapture(sl.compute, &ctx->graph);
        // nothing ran: the captured passes are accounted for when the graph is launched
        r.n = n0; ctx->step_no = step0; ctx->launches = launches0; r.in = in0;
        ctx->passes = passes0; ctx->steps_done = sd0;
        if (st != GS_OK) return st;
        if (e != hipSuccess) return fail(GS_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        GS_HIP(hipGraphInstantiate(&ctx->graph_exec, ctx->graph, nullptr, nullptr, 0));
        ctx->graph_key = key;
    }
    while ((r.steps - r.n) / (uint64_t)kk >= (uint64_t)kGraphBatch) {
        GS_HIP(hipGraphLaunch(ctx->graph_exec, sl.compute));
        r.n += (uint64_t)kGraphBatch * kk;
        ctx->step_no += kGraphBatch;
        ctx->launches += kGraphBatch;
        ctx->passes += kGraphBatch;
        ctx->steps_done += (uint64_t)kGraphBatch * kk;
    }
    for (int i = 0; i < 2; ++i) { // as after the last pass of a batch
        r.u[i]->ghost_depth = kk;
        r.v[i]->ghost_depth = kk;
    }
    return GS_OK;
}

} // namespace gsi

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
using namespace gsi;

extern "C" {

void gs_default_params(gs_params *out)
{
    if (!out) return;
    static const float w[3][3] = {{0.25f, 0.5f, 0.25f}, {0.5f, 0.0f, 0.5f}, {0.25f, 0.5f, 0.25f}};
    std::memcpy(out->w, w, sizeof w);
    out->du = 0.1f;
    out->dv = 0.05f;
    out->feed = 0.014f;
    out->kill = 0.054f;
    out->dt = 1.0f;
}

void gs_default_options(gs_options *out)
{
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    out->math = GS_MATH_STRICT;
    out->kernel = GS_KERNEL_AUTO;
}

int32_t gs_abi_version(void) { return GS_ABI_VERSION; }

int32_t gs_device_count(int32_t *out)
{
    if (!out) return fail(GS_ERR_INVALID, "null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return fail(GS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out = n;
    return GS_OK;
}

int32_t gs_ctx_destroy(gs_ctx *ctx)
{
    if (!ctx) return GS_OK;
    for (auto &sl : ctx->slabs) {
        if (!sl.compute && !sl.halo) continue; // never initialised (creation failed early)
        if (hipSetDevice(sl.device) != hipSuccess) continue;
        if (sl.halo) (void)hipStreamSynchronize(sl.halo);
        if (sl.compute) (void)hipStreamSynchronize(sl.compute);
    }
    if (ctx->comm) {
        if (Rccl *R = rccl()) R->CommDestroy(ctx->comm);
    }
    for (auto &sl : ctx->slabs) {
        if (!sl.compute && !sl.halo) continue;
        if (hipSetDevice(sl.device) != hipSuccess) continue;
        for (int k = 0; k < 2; ++k) {
            if (sl.done[k]) (void)hipEventDestroy(sl.done[k]);
            if (sl.halod[k]) (void)hipEventDestroy(sl.halod[k]);
        }
        if (sl.t0) (void)hipEventDestroy(sl.t0);
        if (sl.t1) (void)hipEventDestroy(sl.t1);
        if (sl.staged) (void)hipEventDestroy(sl.staged);
        for (auto e : sl.copied)
            if (e) (void)hipEventDestroy(e);
        for (auto *v : {&sl.th0, &sl.th1, &sl.tc0, &sl.tc1})
            for (auto e : *v)
                if (e) (void)hipEventDestroy(e);
        if (sl.copy) { (void)hipStreamSynchronize(sl.copy); (void)hipStreamDestroy(sl.copy); }
        if (sl.copy2) { (void)hipStreamSynchronize(sl.copy2); (void)hipStreamDestroy(sl.copy2); }
        for (auto p : sl.stage)
            if (p) (void)hipFree(p);
        if (sl.halo) (void)hipStreamDestroy(sl.halo);
        if (sl.compute) (void)hipStreamDestroy(sl.compute);
    }
    for (auto &b : ctx->bands) {
        if (hipSetDevice(b.device) != hipSuccess) continue;
        if (b.halo) { (void)hipStreamSynchronize(b.halo); (void)hipStreamDestroy(b.halo); }
        if (b.compute) { (void)hipStreamSynchronize(b.compute); (void)hipStreamDestroy(b.compute); }
        for (int k = 0; k < 2; ++k) {
            if (b.done[k]) (void)hipEventDestroy(b.done[k]);
            if (b.halod[k]) (void)hipEventDestroy(b.halod[k]);
        }
    }
    if (!ctx->slabs.empty() && (ctx->win.words || ctx->win.planes[0] || ctx->win.desc) && hipSetDevice(ctx->slabs[0].device) == hipSuccess) {
        for (auto p : ctx->win.planes)
            if (p) (void)hipFree(p);
        if (ctx->win.words) (void)hipFree(ctx->win.words);
        if (ctx->win.desc) (void)hipFree(ctx->win.desc);
    }
    if (ctx->win.seen) (void)hipHostFree(ctx->win.seen);
    if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->graph) (void)hipGraphDestroy(ctx->graph);
    if (ctx->band_join) (void)hipEventDestroy(ctx->band_join);
    if (!ctx->tunings.empty() && !ctx->slabs.empty() && hipSetDevice(ctx->slabs[0].device) == hipSuccess)
        for (auto &t : ctx->tunings)
            for (auto e : t.events)
                if (e) (void)hipEventDestroy(e);
    (void)hipGetLastError(); // teardown failures must not leak into later calls' status
    delete ctx;
    return GS_OK;
}

int32_t gs_ctx_create(gs_ctx **out, const gs_params *params, const gs_options *opts,
                      const int32_t *device_ids, int32_t n_local, int32_t rank, int32_t world,
                      const void *unique_id)
{
    if (!out) return fail(GS_ERR_INVALID, "null output");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(GS_ERR_INVALID, "bad rank %d / world %d", rank, world);
    if (n_local < 0 || (n_local > 0 && !device_ids)) return fail(GS_ERR_INVALID, "bad device list");
    if (world > 1 && !unique_id) return fail(GS_ERR_INVALID, "world > 1 needs the RCCL unique id of rank 0");
    // One slab per process is the deployment (one process per GPU).  A process of a chain may also hold
    // several consecutive slabs as long as they live on ONE device -- the communicator is bound to it -- which
    // is how an 8-slab chain is rehearsed on boxes that admit fewer processes than slabs.
    if (world > 1) {
        for (int i = 1; i < n_local; ++i)
            if (device_ids[i] != device_ids[0])
                return fail(GS_ERR_UNSUPPORTED, "a process of a multi-process chain drives slabs of one device "
                                                "(one process per GPU); got devices %d and %d", device_ids[0], device_ids[i]);
        // ... and only over a transport named with GS_RCCL_LIBRARY (the tests' double): with several local slabs the
        // first and the last slab each issue their own send / recv group, from two streams, on the one communicator
        // -- whether the real RCCL orders two streams on one communicator is version-dependent and was never run.
        const char *user = std::getenv("GS_RCCL_LIBRARY");
        if (n_local > 1 && !(user && *user))
            return fail(GS_ERR_UNSUPPORTED, "several slabs per process of a multi-process chain are a rehearsal mode: "
                                            "set GS_RCCL_LIBRARY to the transport to use, or run one slab per process");
    }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(GS_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));

    gs_ctx *ctx = new (std::nothrow) gs_ctx();
    if (!ctx) return fail(GS_ERR_NOMEM, "out of host memory");
    if (params) ctx->p = *params; else gs_default_params(&ctx->p);
    if (opts) ctx->o = *opts; else gs_default_options(&ctx->o);
    ctx->rank = rank;
    ctx->world = world;
    int32_t st = check_math(ctx->p, ctx->o.math);
    if (st == GS_OK && ctx->o.cols_per_lane != 0 && ctx->o.cols_per_lane != 1 && ctx->o.cols_per_lane != 2 &&
        ctx->o.cols_per_lane != 4)
        st = fail(GS_ERR_INVALID, "cols_per_lane must be 0 (auto), 1, 2 or 4, not %d", ctx->o.cols_per_lane);
    if (st == GS_OK && (ctx->o.tile_shape < 0 || ctx->o.tile_shape > 3))
        st = fail(GS_ERR_INVALID, "tile_shape must be 0 (auto), 1 (32 x 64), 2 (16 x 64) or 3 (64 x 64), not %d", ctx->o.tile_shape);
    if (st == GS_OK && ctx->o.boundary != GS_BOUNDARY_CLIPPED && ctx->o.boundary != GS_BOUNDARY_ZERO_HALO)
        st = fail(GS_ERR_INVALID, "unknown boundary rule %d", ctx->o.boundary);
    if (st == GS_OK && (ctx->o.share_taps < 0 || ctx->o.share_taps > 3))
        st = fail(GS_ERR_INVALID, "share_taps must be 0 (chosen on line), 1 (within a lane), 2 (off) or 3 (across lanes too), not %d",
                  ctx->o.share_taps);
    if (st != GS_OK) { delete ctx; return st; }

    const int32_t one = 0;
    if (n_local == 0) { device_ids = &one; n_local = 1; }
    ctx->slabs.resize(n_local);
    auto bail = [&](int32_t code) { gs_ctx_destroy(ctx); return code; };
    for (int i = 0; i < n_local; ++i) {
        SlabRt &sl = ctx->slabs[i];
        sl.device = device_ids[i];
        if (sl.device < 0 || sl.device >= ndev)
            return bail(fail(GS_ERR_NO_DEVICE, "device %d out of range (have %d)", sl.device, ndev));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sl.device) != hipSuccess)
            return bail(fail(GS_ERR_HIP, "hipGetDeviceProperties(%d) failed", sl.device));
        if (i == 0) ctx->cu_count = prop.multiProcessorCount;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return bail(fail(GS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                             sl.device, prop.gcnArchName));
#define GS_HIP_B(expr)                                                                         \
    do {                                                                                       \
        hipError_t e2_ = (expr);                                                               \
        if (e2_ != hipSuccess)                                                                 \
            return bail(fail(GS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_)));     \
    } while (0)
        GS_HIP_B(hipSetDevice(sl.device));
        int least = 0, greatest = 0;
        GS_HIP_B(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // (A compute stream whose CU mask leaves 1 or 2 CUs per XCD to the halo stream was measured in round 4: 4 / 8
        // slabs on one GPU ran at 0.59 / 0.40 of the unmasked chain with 1 CU per XCD left out, 0.65 / 0.74 with 2
        // -- profiles/r04_sweeps.md, section 1 -- and is not offered.)
        GS_HIP_B(hipStreamCreateWithPriority(&sl.compute, hipStreamNonBlocking, least));
        GS_HIP_B(hipStreamCreateWithPriority(&sl.halo, hipStreamNonBlocking, greatest));
        GS_HIP_B(hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking));
        GS_HIP_B(hipStreamCreateWithFlags(&sl.copy2, hipStreamNonBlocking));
        GS_HIP_B(hipEventCreateWithFlags(&sl.staged, hipEventDisableTiming));
        GS_HIP_B(hipEventCreateWithFlags(&sl.copied[0], hipEventDisableTiming));
        GS_HIP_B(hipEventCreateWithFlags(&sl.copied[1], hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) {
            GS_HIP_B(hipEventCreateWithFlags(&sl.done[k], hipEventDisableTiming));
            GS_HIP_B(hipEventCreateWithFlags(&sl.halod[k], hipEventDisableTiming));
        }
        GS_HIP_B(hipEventCreate(&sl.t0));
        GS_HIP_B(hipEventCreate(&sl.t1));
    }
    // Peer access between neighbouring local slabs on different devices (best effort: the
    // copies fall back to staged transfers when it is unavailable).
    for (int i = 0; i + 1 < n_local; ++i) {
        const int a = ctx->slabs[i].device, b = ctx->slabs[i + 1].device;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
            (void)hipSetDevice(a);
            (void)hipDeviceEnablePeerAccess(b, 0);
            (void)hipSetDevice(b);
            (void)hipDeviceEnablePeerAccess(a, 0);
            (void)hipGetLastError();
        }
    }
    if (world > 1) {
        Rccl *R = rccl();
        if (!R) return bail(fail(GS_ERR_RCCL, "librccl could not be loaded"));
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        GS_HIP_B(hipSetDevice(ctx->slabs[0].device));
        ncclResult_t r = R->CommInitRank(&ctx->comm, world, id, rank);
        if (r != ncclSuccess)
            return bail(fail(GS_ERR_RCCL, "ncclCommInitRank failed: %s", R->GetErrorString(r)));
    }
#undef GS_HIP_B
    *out = ctx;
    return GS_OK;
}

int32_t gs_ctx_set_params(gs_ctx *ctx, const gs_params *params)
{
    if (!ctx || !params) return fail(GS_ERR_INVALID, "null argument");
    GS_TRY(check_math(*params, ctx->o.math));
    GS_TRY(resolve_window(ctx)); // (a window launch in flight that gave up is run again with the parameters it was enqueued with)
    ctx->p = *params;
    return GS_OK;
}

int32_t gs_step(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v)
{
    GS_TRY(check_step_fields(ctx, in_u, in_v, out_u, out_v));
    GS_TRY(resolve_window(ctx));
    return step_impl(ctx, in_u, in_v, out_u, out_v);
}

int32_t gs_run(gs_ctx *ctx, gs_field *u0, gs_field *v0, gs_field *u1, gs_field *v1, uint64_t steps,
               int32_t *result_slot)
{
    return run_steps(ctx, u0, v0, u1, v1, steps, result_slot, true);
}

} // extern "C"

namespace gsi {

// gs_run.  allow_window = false: never the persistent window kernel (the replay of launches that gave up).
int32_t run_steps(gs_ctx *ctx, gs_field *u0, gs_field *v0, gs_field *u1, gs_field *v1, uint64_t steps, int32_t *result_slot,
                  bool allow_window)
{
    GS_TRY(check_step_fields(ctx, u0, v0, u1, v1));
    Run r{ctx, {u0, u1}, {v0, v1}};
    r.steps = steps;
    // Temporal blocking: `fuse` steps per pass over HBM (default 4, the measured optimum);
    // bounded by the ghost depth and by the smallest slab of the partition.
    int fuse = 1;
    if (ctx->o.kernel == GS_KERNEL_AUTO || ctx->o.kernel == GS_KERNEL_TB || ctx->o.kernel == GS_KERNEL_TILE ||
        ctx->o.kernel == GS_KERNEL_WINDOW) {
        fuse = ctx->o.fuse_steps > 0 ? ctx->o.fuse_steps : kGhostRows;
        if (fuse > kGhostRows) fuse = kGhostRows;
        if (ctx->total_slabs() > 1 && fuse > min_slab_rows(ctx, u0)) fuse = min_slab_rows(ctx, u0);
        if (fuse < 1) fuse = 1;
    }
    const bool single = ctx->total_slabs() == 1;
    // Small grids (single slab, kernel = auto): the whole run is one launch with the grid resident
    // in LDS (gs_run_resident_k) -- up to kGsResidentCells = 1536 cells; above, the window kernel is faster (1536
    // cells: 1630 against 1558 Mcells x steps / s; 2048: 1596 against 2062; 4096: 1695 against 4153; run 48).
    if (single && ctx->o.kernel == GS_KERNEL_AUTO && u0->rows * u0->cols > 0 && steps > 0 &&
        u0->rows * u0->cols <= (uint64_t)kGsResidentCells) {
        SlabRt &sl = ctx->slabs[0];
        GS_HIP(hipSetDevice(sl.device));
        GS_TRY(join_bands(ctx, sl.compute));
        ctx->bands_active = false;
        uint64_t left = steps;
        int slot = 0;
        while (left > 0) { // the step count is an int in the kernel
            const int n = left > 0x40000000ull ? 0x40000000 : (int)left;
            GsStepArgs a = make_args(ctx, r.u[slot], r.v[slot], r.u[1 - slot], r.v[1 - slot], 0, 1);
            const char *name = nullptr;
            const hipError_t e = ctx->o.math == GS_MATH_FUSED ? gs_launch_resident_fused(a, n, sl.compute, &name)
                                                               : gs_launch_resident_strict(a, n, sl.compute, &name);
            if (e != hipSuccess) return fail(GS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
            ctx->last_kernel = name;
            ctx->launches++;
            ctx->passes++;
            ctx->steps_done += (uint64_t)n;
            ctx->step_no += (uint64_t)n;
            slot ^= n & 1;
            left -= (uint64_t)n;
        }
        if (result_slot) *result_slot = slot;
        return GS_OK;
    }
    // Grids of one round of register-resident windows (single slab; the reference's default 1080 x 1920 is 252 of them):
    // the whole call is ONE persistent launch of gs_run_window_k, which trades the windows' aprons between workgroups
    // itself every k steps.  kernel = auto takes it from kWindowAutoCells up to the largest grid that is one workgroup per CU
    // when nothing is pinned and the call is long enough to pay for the launch's fixed cost (kWindowAutoSteps);
    // GS_KERNEL_WINDOW forces it (fuse_steps = steps per exchange, rows_per_block = window rows: 80).
    // 640-660 k against 391-416 k Mcells x steps / s at 1080 x 1920, both boundary rules (profiles/r06_window_kernel.md).
    const uint64_t cells = u0->rows * u0->cols;
    if (allow_window && single && cells > 0 && steps > 0 && !ctx->win.disabled) {
        const bool forced = ctx->o.kernel == GS_KERNEL_WINDOW;
        const bool automatic = ctx->o.kernel == GS_KERNEL_AUTO && ctx->o.fuse_steps == 0 && ctx->o.rows_per_block == 0 &&
                               ctx->o.cols_per_lane == 0 && ctx->o.split <= 1 && !ctx->o.use_graph && cells >= kWindowAutoCells &&
                               steps >= kWindowAutoSteps;
        if (forced || automatic) {
            int32_t launched = 0;
            GS_TRY(run_window(ctx, r, steps, forced, &launched, result_slot));
            if (launched) return GS_OK;
        }
    }
    // every other kernel reads or overwrites planes that a window launch still in flight may own
    GS_TRY(resolve_window(ctx));
    // Mid-size grids (single slab): K <= 8 steps per launch on LDS-resident windows (gs_run_tile_k), where a
    // pass of the temporally blocked kernel is bound by the length of a wave's march and a launch per <= 4
    // steps.  kernel = auto picks it between the resident kernel's 1536 cells and 1.5 M cells when nothing
    // is pinned, with the window and steps per launch of pick_tile_config (profiles/archive/r02_sweeps.md, section
    // 10: 2.2x at 64 x 128 and 128 x 256, 1.8x at 256 x 512, 1.4x at 512 x 1024; at 1080 x 1920 the marching
    // kernel is ahead again); GS_KERNEL_TILE forces it (tile_shape and fuse_steps then choose the window
    // and the steps per launch).
    int auto_shape = -1, auto_k = 0;
    if (single && ctx->o.kernel == GS_KERNEL_AUTO && ctx->o.fuse_steps == 0 && ctx->o.rows_per_block == 0 &&
        ctx->o.cols_per_lane == 0 && ctx->o.split <= 1 && !ctx->o.use_graph && cells > (uint64_t)kGsResidentCells &&
        cells < kTileAutoCells)
        pick_tile_config((long)u0->rows, (long)u0->cols, &auto_shape, &auto_k);
    if (single && cells > 0 && steps > 0 && (ctx->o.kernel == GS_KERNEL_TILE || auto_shape >= 0)) {
        SlabRt &sl = ctx->slabs[0];
        GS_HIP(hipSetDevice(sl.device));
        GS_TRY(join_bands(ctx, sl.compute));
        ctx->bands_active = false;
        // window shape (gs_launch_tile): 0 = 32 rows x 64 columns, 1 = 16 x 64, 2 = 64 x 64; steps per launch:
        // 8, or 4 for the 16-row window, whose apron would otherwise outweigh what it produces
        int shape = auto_shape >= 0 ? auto_shape : 0;
        if (auto_shape < 0 && ctx->o.tile_shape >= 1 && ctx->o.tile_shape <= 3) shape = ctx->o.tile_shape - 1;
        const int window_rows = shape == 0 ? 32 : (shape == 1 ? 16 : 64);
        int kmax = auto_shape >= 0 ? auto_k : (shape == 1 ? 4 : kGsTileMaxSteps);
        if (auto_shape < 0 && ctx->o.fuse_steps > 0)
            kmax = ctx->o.fuse_steps > kGsTileMaxSteps ? kGsTileMaxSteps : ctx->o.fuse_steps;
        if (2 * kmax >= window_rows) kmax = window_rows / 2 - 1;
        uint64_t left = steps;
        int slot = 0;
        const char *full_name = nullptr;
        while (left > 0) { // the short launch first, then full ones
            const int n = left % (uint64_t)kmax ? (int)(left % (uint64_t)kmax) : kmax;
            GsStepArgs a = make_args(ctx, r.u[slot], r.v[slot], r.u[1 - slot], r.v[1 - slot], 0, 1);
            const char *name = nullptr;
            const hipError_t e = ctx->o.math == GS_MATH_FUSED ? gs_launch_tile_fused(a, n, shape, sl.compute, &name)
                                                               : gs_launch_tile_strict(a, n, shape, sl.compute, &name);
            if (e != hipSuccess) return fail(GS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
            if (!full_name || n == kmax) full_name = name;
            ctx->launches++;
            ctx->passes++;
            ctx->steps_done += (uint64_t)n;
            ctx->step_no++;
            slot ^= 1;
            left -= (uint64_t)n;
        }
        ctx->last_kernel = full_name;
        if (result_slot) *result_slot = slot;
        return GS_OK;
    }
    // The short pass goes first so that a run ends on a full pass -- a full-depth ghost exchange -- and the
    // next run can start without a blocking refresh.  It is sized with the steps per pass in force (a
    // configuration handed in through gs_ctx_set_tuned may fuse fewer steps than `fuse`), which is known
    // before anything runs on a slab chain; a single slab may still change it below (on-line tuning), where
    // a remainder pass at the end costs nothing.
    recall_tuned(ctx, u0, fuse);
    {
        const int kk0 = tuned_shape(ctx, u0, fuse) && ctx->tuned_k > 0 && ctx->tuned_k <= fuse ? ctx->tuned_k : fuse;
        if (steps % (uint64_t)kk0) GS_TRY(r.advance(1, (int)(steps % (uint64_t)kk0)));
    }
    if (single && fuse > 1 && ctx->o.rows_per_block == 0 && !ctx->o.no_tune && !tuned_shape(ctx, u0, fuse))
        GS_TRY(tune_online(r, fuse));
    // Steps per full pass: the tuned value -- on a slab chain every process must have been given the
    // same one (gs_ctx_set_tuned), since the ghost-row exchange is K rows deep.
    const int kk = tuned_shape(ctx, u0, fuse) && ctx->tuned_k > 0 && ctx->tuned_k <= fuse ? ctx->tuned_k : fuse;
    const int V = bands_for(ctx, u0, kk);
    if (ctx->o.use_graph && single && V == 1 && kk > 1) GS_TRY(replay_graph_batches(r, kk));
    const char *full_pass = nullptr;
    while (r.n < steps) {
        const int k = (steps - r.n) >= (uint64_t)kk ? kk : (int)(steps - r.n);
        GS_TRY(r.advance(k == kk ? V : 1, k));
        if (k == kk) full_pass = ctx->last_kernel;
    }
    if (full_pass) ctx->last_kernel = full_pass; // gs_ctx_info names the full pass, not a remainder
    if (result_slot) *result_slot = r.in;
    return GS_OK;
}

} // namespace gsi

extern "C" {

int32_t gs_sync(gs_ctx *ctx)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    return sync_all(ctx);
}

int32_t gs_timer_start(gs_ctx *ctx)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipEventRecord(sl.t0, sl.compute));
    }
    return GS_OK;
}

int32_t gs_timer_stop(gs_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return fail(GS_ERR_INVALID, "null argument");
    const int last = (int)((ctx->step_no + 1) & 1); // parity of the most recent step
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        if (ctx->total_slabs() > 1 && ctx->step_no > 0)
            GS_HIP(hipStreamWaitEvent(sl.compute, sl.halod[last], 0));
        if (&sl == &ctx->slabs[0]) GS_TRY(join_bands(ctx, sl.compute));
        GS_HIP(hipEventRecord(sl.t1, sl.compute));
    }
    float worst = 0.0f;
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipEventSynchronize(sl.t1));
        float ms = 0.0f;
        GS_HIP(hipEventElapsedTime(&ms, sl.t0, sl.t1));
        if (ms > worst) worst = ms;
    }
    *elapsed_ms = worst;
    return GS_OK;
}

int32_t gs_ctx_set_pass_timing(gs_ctx *ctx, int32_t passes)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    if (passes < 0 || passes > 4096) return fail(GS_ERR_INVALID, "pass timing covers 0 to 4096 passes, not %d", passes);
    GS_TRY(sync_all(ctx)); // the events of an earlier window must not be re-recorded while in flight
    ctx->pass_timing = passes;
    for (auto &sl : ctx->slabs) sl.timed = 0;
    return GS_OK;
}

int32_t gs_ctx_stats(gs_ctx *ctx, gs_stats *out)
{
    if (!ctx || !out) return fail(GS_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    out->passes = ctx->passes;
    out->steps = ctx->steps_done;
    out->launches = ctx->launches;
    out->ghost_refreshes = ctx->ghost_refreshes;
    out->window_fallbacks = ctx->win.fallbacks;
    // timed passes (slab chains only): the slowest local slab's sums
    for (auto &sl : ctx->slabs) {
        if (sl.timed == 0) continue;
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipEventSynchronize(sl.th1[sl.timed - 1]));
        GS_HIP(hipEventSynchronize(sl.tc1[sl.timed - 1]));
        double halo = 0.0, interior = 0.0, exposed = 0.0;
        for (int k = 0; k < sl.timed; ++k) {
            float h = 0.f, c = 0.f, x = 0.f;
            GS_HIP(hipEventElapsedTime(&h, sl.th0[k], sl.th1[k]));
            GS_HIP(hipEventElapsedTime(&c, sl.tc0[k], sl.tc1[k]));
            // how long after the interior kernel's end the halo stream's work ended (<= 0: hidden)
            GS_HIP(hipEventElapsedTime(&x, sl.tc1[k], sl.th1[k]));
            halo += h; interior += c; exposed += x > 0.f ? x : 0.f;
        }
        if (interior + exposed >= (double)out->interior_ms + (double)out->halo_exposed_ms) {
            out->timed_passes = (uint64_t)sl.timed;
            out->halo_ms = (float)halo;
            out->interior_ms = (float)interior;
            out->halo_exposed_ms = (float)exposed;
        }
    }
    return GS_OK;
}

int32_t gs_ctx_info(const gs_ctx *ctx, char *kernel_name, size_t cap, uint64_t *launches)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    if (kernel_name && cap) {
        if (ctx->tuned_rpu > 0)
            std::snprintf(kernel_name, cap, "%s@%dx%d", ctx->last_kernel, ctx->tuned_rpu,
                          ctx->tuned_split > 0 ? ctx->tuned_split : 1); // tuned unit height x row bands
        else
            std::snprintf(kernel_name, cap, "%s", ctx->last_kernel);
    }
    if (launches) *launches = ctx->launches;
    return GS_OK;
}

} // extern "C"
