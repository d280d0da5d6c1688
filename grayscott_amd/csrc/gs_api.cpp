// gs_api.cpp -- implementation of the C ABI in include/gs_hip.h.
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
#include "../../include/gs_hip.h"
#include "gs_kernels.h"
#include "gs_experiments.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define GS_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(GS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                        __FILE__, __LINE__);                                                   \
    } while (0)

#define GS_TRY(expr)                                                                           \
    do {                                                                                       \
        int32_t s_ = (expr);                                                                   \
        if (s_ != GS_OK) return s_;                                                            \
    } while (0)

// ---------------------------------------------------------------------------------------
// RCCL, loaded on first use so that single-process users never touch it
// ---------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // introspection (gs_ctx_comm_info); optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
};

Rccl *rccl()
{
    // function-local static: initialised once, thread-safe (C++11)
    static Rccl *const instance = []() -> Rccl * {
        static Rccl r;
        // GS_RCCL_LIBRARY names the library to bind instead of the system's librccl (a custom
        // RCCL build; the tests' shared-memory transport double, tests/cpp/shm_transport.cpp)
        const char *user = std::getenv("GS_RCCL_LIBRARY");
        for (const char *name : {user, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (!name || !*name) continue;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle || name == user) break; // an explicit choice never falls back silently
        }
        if (!r.handle) return nullptr;
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p) ok = false;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
            return nullptr;
        }
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.handle, "ncclCommCuDevice"));
        return &r;
    }();
    return instance;
}

#define GS_NCCL(R, expr)                                                                       \
    do {                                                                                       \
        ncclResult_t e_ = (expr);                                                              \
        if (e_ != ncclSuccess)                                                                 \
            return fail(GS_ERR_RCCL, "%s failed: %s", #expr, (R)->GetErrorString(e_));          \
    } while (0)

static_assert(sizeof(ncclUniqueId) == GS_UNIQUE_ID_BYTES, "RCCL unique id size changed");

} // namespace

// ---------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------
struct SlabRt {
    int device = 0;
    hipStream_t compute = nullptr, halo = nullptr, copy = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr}, halod[2] = {nullptr, nullptr};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t staged = nullptr, copied = nullptr; // asynchronous downloads
    float *stage = nullptr;                        // dense device staging buffer
    size_t stage_floats = 0;
    // gs_ctx_set_pass_timing: per timed pass, events around the halo stream's work (boundary-band kernel +
    // ghost-row exchange: th0, th1) and around the interior kernel on the compute stream (tc0, tc1)
    std::vector<hipEvent_t> th0, th1, tc0, tc1;
    int timed = 0; // passes recorded since the timing was switched on
};

struct gs_ctx {
    gs_params p;
    gs_options o;
    std::vector<SlabRt> slabs; // local slabs, top to bottom
    std::vector<SlabRt> bands; // stream/event sets for the in-place row bands of a single slab
    hipEvent_t band_join = nullptr;
    bool bands_active = false; // the newest pass ran on the band streams
    int bands_v = 0, bands_rows = 0, bands_k = 0; // layout of that pass
    int rank = 0, world = 1;
    uint64_t step_no = 0;
    ncclComm_t comm = nullptr;
    const char *last_kernel = "none";
    uint64_t launches = 0;
    uint64_t passes = 0, steps_done = 0, ghost_refreshes = 0; // gs_ctx_stats
    int pass_timing = 0;                                      // passes per slab still to be timed (0 = off)
    // Configuration of the temporally blocked kernel in force (tuned_rpu > 0): unit height, fused steps
    // per pass and columns per lane for slabs of tuned_rows x tuned_cols -- chosen by gs_run's on-line
    // tuner (single-slab contexts) or handed in through gs_ctx_set_tuned (slab chains).
    uint64_t tuned_rows = 0, tuned_cols = 0;
    int tuned_fuse = 0, tuned_rpu = 0, tuned_split = 0, tuned_k = 0; // tuned_k: fused steps per pass chosen
    int tuned_cpl = 0;                                               // columns per lane chosen
    int tuned_share = 1;                                             // full difference sharing chosen (0 / 1)
    // every finished choice (a context that alternates between grids does not re-tune)
    struct Tuned { uint64_t rows, cols; int fuse, rpu, split, k, cpl, share; };
    std::vector<Tuned> tuned_cache;
    // Tunings in progress, one per shape (each may span several gs_run calls; two grids driven
    // alternately advance independently).  `batch` / `nb`: timing windows enqueued but not read yet.
    struct Trial { int rpu, V, k, cpl, reps, share; };
    struct Tuning {
        uint64_t rows = 0, cols = 0;
        int fuse = 0, next = 0, best_rpu = 0, best_split = 0, best_k = 0, best_cpl = 0, best_share = 1;
        float best_ms = 0.f;
        Trial batch[16];
        int nb = 0;
        std::vector<hipEvent_t> events; // 3 per window (created on first use)
    };
    std::vector<Tuning> tunings;
    // gs_options.use_graph: a batch of passes captured once and replayed (single slab, no bands).
    // The captured launches carry plane addresses and parameters, so the key holds all of them.
    struct GraphKey {
        const void *planes[4] = {nullptr, nullptr, nullptr, nullptr};
        uint64_t rows = 0, cols = 0;
        int k = 0, rpu = 0, cpl = 0, batch = 0;
        gs_params p{};
        bool operator==(const GraphKey &o) const
        {
            return std::memcmp(planes, o.planes, sizeof planes) == 0 && rows == o.rows && cols == o.cols && k == o.k &&
                   rpu == o.rpu && cpl == o.cpl && batch == o.batch && std::memcmp(&p, &o.p, sizeof p) == 0;
        }
    } graph_key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // gs_run_window_k (one persistent launch per gs_run on grids of one round of windows): exchange planes, flags and
    // the abort word, sized for one plane shape at a time (single-slab contexts only)
    struct WindowRt {
        float *planes[4] = {nullptr, nullptr, nullptr, nullptr}; // xu[0], xu[1], xv[0], xv[1]
        int32_t *words = nullptr;                                // kWindowMaxTiles flags, then the abort word
        GsWindowDesc *desc = nullptr;                            // kWindowMaxTiles window descriptors (device)
        uint64_t plan_rows = 0, plan_cols = 0;                   // the tiling `desc` holds ...
        int plan_rpw = 0, plan_k = 0, plan_n = 0, plan_key = -1; // ... its windows, and what else it was made for
        uint64_t rows = 0, pitch = 0;
        int32_t epoch = 0;
        bool pending = false;  // a launch has been enqueued since the abort word was last read
        bool disabled = false; // a launch gave up once: this context stays with the marching kernel
        // The launches behind `pending`, in order (planes in -> planes out, `steps` time steps, numbered `seq`).  A launch
        // that gives up leaves its number in the abort word: the launches before it ran to their end and their results
        // stand; that launch and every later one (they leave at once: the word is sticky) are run again with the
        // marching kernel, from the input planes of the first of them, which no launch has written (resolve_window).
        struct Launch { gs_field *in[2], *out[2]; int steps; int32_t seq; int passes; };
        std::vector<Launch> launched;
        int32_t seq = 0;
        uint64_t fallbacks = 0;
    } win;
    int share_now = 1; // full difference sharing in force when gs_options.share_taps leaves the choice open (fast_of)
    int cu_count = 0; // compute units of the first slab's device
    int total_slabs() const { return world * (int)slabs.size(); }
    int global_index(int i) const { return rank * (int)slabs.size() + i; }
};

struct FieldSlab {
    float *alloc = nullptr; // hipMalloc'ed block: guard | 4 ghost rows | rows | 4 ghost rows | guard
    float *row0 = nullptr;  // local row 0, column 0
    uint64_t g_row0 = 0;    // global index of local row 0
    int32_t rows = 0;
};

struct gs_field {
    gs_ctx *ctx = nullptr;
    uint64_t rows = 0, cols = 0;
    int32_t pitch = 0;
    std::vector<FieldSlab> s;
    int ghost_depth = 0; // ghost rows currently holding the neighbours' data (0 = stale)
};

namespace {

constexpr int kWindowMaxTiles = 1024; // flags of gs_run_window_k (one per workgroup; a launch has at most one per CU)
constexpr int kGuardFloats = 64; // 256 B in front of / behind every plane
constexpr int kGhostRows = 4;    // ghost rows kept above and below every slab (= max fused steps)

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

int32_t run_steps(gs_ctx *ctx, gs_field *u0, gs_field *v0, gs_field *u1, gs_field *v1, uint64_t steps, int32_t *result_slot,
                  bool allow_window);

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
    if (!gave_up) return GS_OK;
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
    }
    GS_HIP(hipStreamSynchronize(sl.compute));
    return GS_OK;
}

int32_t sync_all(gs_ctx *ctx)
{
    for (auto &sl : ctx->slabs) {
        GS_HIP(hipSetDevice(sl.device));
        GS_HIP(hipStreamSynchronize(sl.halo));
        GS_HIP(hipStreamSynchronize(sl.compute));
        GS_HIP(hipStreamSynchronize(sl.copy));
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

// Output columns per wave of the temporally blocked kernel (gs_step_kernels.hip: tb_cols_per_wave).
long tb_strips(int32_t cols, int fuse, int cpl)
{
    const long w = (64 - 2 * ((fuse + cpl - 1) / cpl)) * (long)cpl;
    return (cols + w - 1) / w;
}

// Is full difference sharing in force (when the parameters allow it)?  Pinned by gs_options.share_taps, else what the
// on-line tuner last chose or is trying (gs_ctx::share_now), else on.
bool share_on(const gs_ctx *ctx) { return ctx->o.share_taps == 1 || (ctx->o.share_taps == 0 && ctx->share_now != 0); }

// GsStepArgs::fast for this context's parameters: bit 0 = the four side weights are 0.5, bit 1 = dt == 1, bit 2 = both
// and the diagonal weights are pairwise equal and the context wants full difference sharing.
int fast_of(const gs_ctx *ctx)
{
    int fast = 0;
    if (!ctx->o.general_kernels) {
        const float(*w)[3] = ctx->p.w;
        if (w[0][1] == 0.5f && w[1][0] == 0.5f && w[1][2] == 0.5f && w[2][1] == 0.5f) fast |= 1;
        if (ctx->p.dt == 1.0f) fast |= 2;
        // full difference sharing (cells_vshare): the diagonal taps of a row pair are each other's negatives
        if (fast == 3 && w[0][0] == w[2][2] && w[0][2] == w[2][0] && ctx->o.share_taps != 2 && share_on(ctx)) fast |= 4;
    }
    return fast;
}

// Unit heights that make a launch of the temporally blocked kernel exactly `r` rounds of the chip's wave
// slots (256 CUs x 4 SIMDs x the kernel entry's waves per SIMD): strips x chunks <= r x slots with the
// chunks as short as that allows.  A launch that misses such a height by one chunk runs a nearly empty
// extra round: at 4096^2 with 2 columns per lane 36 rows give 749 k Mcells x steps/s, 32 rows 677 k, 40
// rows 687 k (profiles/r02_sweeps.md, section 9).  From two rounds up the launcher tapers the last two
// rounds (an eighth and a half as tall: 0.625 rounds' worth of rows), which the formula accounts for.
// Writes up to `max` heights (the single-round one first); returns their number.
int fit_heights(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl, int fast, int *out, int max, bool partial = false)
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
// fastest from 4096^2 up, profiles/r01_sweeps.md runs 54-57) unless that cannot give every SIMD a
// wave at a unit height of 8 * fuse rows, then 1.
int32_t pick_cols_per_lane(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse)
{
    if (ctx->o.cols_per_lane > 0) return ctx->o.cols_per_lane;
    if (tuned_for(ctx, rows, cols, fuse) && ctx->tuned_cpl > 0) return ctx->tuned_cpl;
    if (fuse < 2) return 2;
    return (long)rows * tb_strips(cols, fuse, 2) / (8L * fuse) >= 2048 ? 2 : 1;
}

// Rows each wave marches over when nothing was tuned on line (slab chains, short runs).  Measured
// at 16384^2 (profiles/r01_sweeps.md, sweep17/18): 16 rows for single steps, 128 for 4 fused steps.
int32_t model_rows_per_unit(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl);
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
    // against 1000-1008 k with 122 (profiles/r03_sweeps.md, section 5).
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

// Window shape and steps per launch of the LDS-window kernel (gs_run_tile_k) for a grid, from a cost model
// fitted to the measured launches (profiles/r02_sweeps.md, section 10): a launch costs T0 = 3.4 / 2.6 / 3.7 us
// (launch gap, weights, window load and store) plus K steps of 0.74 / 0.585 / 1.38 us for the 32 / 16 / 64-row
// window while every workgroup has a CU to itself; beyond 256 workgroups they run in rounds (two share a CU
// at 0.87 of the time of two turns).  The model is within ~15 % of the measured rates from 64 x 128 to 1024 x
// 1024 and picks the measured-best or second-best configuration on every grid of that table.
constexpr uint64_t kTileAutoCells = 1536 * 1024; // above, the marching kernel is ahead (1080 x 1920: 380-420 k vs 350 k)
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

// ---- gs_run_window_k: grids of one round of register-resident windows ----------------------------------------
// The tiling of a grid into windows, or an empty plan when the grid is not one round of workgroups (one workgroup per
// CU at most, all of them resident for the whole launch).  Tile columns are 128 - 2 k owned columns wide; the windows of
// a tile column share their height.  Every workgroup waits for its neighbours at every exchange, so the slowest
// window sets the pace of the whole grid: columns whose cells cost more instructions get LOWER windows (fewer waves
// in use), so that a step takes every workgroup the same time.  Costs relative to an interior window (ISA and
// tools/window_timeline.py, profiles/r04_window_kernel.md): the grid's left edge under the clipped rule 1.20 (cell<2>: 18
// selects for the cell on column 0's lane), the right edge 1.13 (cell<3>), general path (general weights, the fused
// build, a grid of one tile column) 1.6, edge columns under the zero-halo rule 1.05 (a select per cell).
// `waves_env`: GS_HIP_WINDOW_WAVES = "left,interior,right" overrides the waves in use per column class (experiments).
std::vector<GsWindowDesc> plan_windows(int cu_count, bool zero_halo, bool cheap, uint64_t rows, uint64_t cols, int want_rpw, int want_k,
                                       int *rpw_out, int *k_out);
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
    std::vector<GsWindowDesc> plan;
    if (cu_count <= 0 || rows == 0 || cols == 0 || rows > 0x7fffff || cols > 0x7fffff) return plan;
    const int k = want_k > 0 ? want_k : 4;
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
        const size_t bytes = (size_t)(f->rows + 1) * (size_t)f->pitch * sizeof(float);
        for (auto &p : w.planes) GS_HIP(hipMalloc(reinterpret_cast<void **>(&p), bytes));
        w.rows = f->rows;
        w.pitch = (uint64_t)f->pitch;
    }
    if (w.epoch > (1 << 30)) { // keep flag arithmetic far from wrapping: start over behind everything enqueued
        GS_HIP(hipMemsetAsync(w.words, 0, kWindowMaxTiles * sizeof(int32_t), ctx->slabs[0].compute));
        w.epoch = 0;
    }
    return GS_OK;
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
    // 936 k for one configuration, profiles/r01_sweeps.md runs 58-61); one launch per pass does not.
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
                  int fuse = 1)
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

// ---- gs_run: state of one call, on-line tuning, graph replay ---------------------------------
struct Run {
    gs_ctx *ctx;
    gs_field *u[2], *v[2];
    int in = 0;          // slot that holds the newest state
    uint64_t n = 0;      // steps done
    uint64_t steps = 0;  // steps wanted

    // one pass of k fused steps, on V row bands when V > 1
    int32_t advance(int V, int k)
    {
        const int32_t st = V > 1 ? step_bands(ctx, u[in], v[in], u[1 - in], v[1 - in], k, V)
                                 : step_impl(ctx, u[in], v[in], u[1 - in], v[1 - in], k);
        in = 1 - in;
        n += (uint64_t)k;
        return st;
    }
};

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
//   chosen configuration without full difference sharing -- A-D run with it.  A-C run with the untuned layout
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
    // phase E: one candidate, where the choice is open and a variant with full difference sharing exists at all
    const bool share_open = ctx->o.share_taps == 0 && ctx->o.math == GS_MATH_STRICT && [&] {
        gs_ctx probe = *ctx; // (fast_of reads options and parameters only)
        probe.o.share_taps = 1;
        return (fast_of(&probe) & 4) != 0;
    }();
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
                             t.share ? "" : ", taps not shared", ms, w0, w1);
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
    int warm_cpl = 0, warm_k = 0, warm_share = 1; // kernel of the newest pass enqueued by this call
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
            gs_ctx::Trial t{0, V0, fuse, base_cpl, reps, 1};
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
            } else if (phase == 4) { // what phases A-D chose, without full difference sharing
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
            // than the untuned default (criterion grid, 16-step calls: profiles/r02_criterion_grid.md).
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
            ctx->share_now = 1;
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
                         done.share ? "shared" : "not shared");
        for (auto e : tu->events)
            if (e) (void)hipEventDestroy(e);
        ctx->tunings.erase(ctx->tunings.begin() + (tu - ctx->tunings.data()));
    }
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

} // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
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

const char *gs_last_error(void) { return g_last_error.c_str(); }

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

int32_t gs_get_unique_id(void *out128)
{
    if (!out128) return fail(GS_ERR_INVALID, "null output");
    Rccl *R = rccl();
    if (!R) return fail(GS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    GS_NCCL(R, R->GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof id);
    return GS_OK;
}

int32_t gs_rccl_selftest(int32_t device, uint64_t floats)
{
    Rccl *R = rccl();
    if (!R) return fail(GS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    if (floats == 0 || floats > (1ull << 28)) return fail(GS_ERR_INVALID, "message of %llu floats", (unsigned long long)floats);
    GS_HIP(hipSetDevice(device));
    ncclUniqueId id;
    GS_NCCL(R, R->GetUniqueId(&id));
    ncclComm_t comm = nullptr;
    GS_NCCL(R, R->CommInitRank(&comm, 1, id, 0));
    float *src = nullptr, *dst = nullptr;
    hipStream_t stream = nullptr;
    std::vector<float> host(floats), back(floats);
    for (uint64_t i = 0; i < floats; ++i) host[i] = (float)(i % 65521) * 0.25f + 1.0f;
    int32_t st = GS_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (st == GS_OK && e != hipSuccess) st = fail(GS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    int least = 0, greatest = 0;
    step(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    step(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    step(hipMalloc(reinterpret_cast<void **>(&src), floats * sizeof(float)), "hipMalloc");
    step(hipMalloc(reinterpret_cast<void **>(&dst), floats * sizeof(float)), "hipMalloc");
    step(hipMemcpy(src, host.data(), floats * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    step(hipMemset(dst, 0, floats * sizeof(float)), "hipMemset");
    if (st == GS_OK) {
        // the call pattern of push_halo: one group, a send and the matching receive, on the halo stream
        ncclResult_t r = R->GroupStart();
        if (r == ncclSuccess) r = R->Send(src, (size_t)floats, ncclFloat, 0, comm, stream);
        if (r == ncclSuccess) r = R->Recv(dst, (size_t)floats, ncclFloat, 0, comm, stream);
        const ncclResult_t e = R->GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) st = fail(GS_ERR_RCCL, "grouped ncclSend / ncclRecv to self failed: %s", R->GetErrorString(r));
    }
    step(hipStreamSynchronize(stream), "hipStreamSynchronize");
    step(hipMemcpy(back.data(), dst, floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (st == GS_OK && std::memcmp(back.data(), host.data(), floats * sizeof(float)) != 0)
        st = fail(GS_ERR_RCCL, "the message came back altered");
    if (stream) (void)hipStreamDestroy(stream);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    R->CommDestroy(comm);
    (void)hipGetLastError();
    return st;
}

// Which HIP runtime and which RCCL this process's libgs_hip.so is bound to (dladdr of an entry point of each), with
// their versions.  A process that imported torch first resolves libamdhip64.so.7 and librccl.so.1 by SONAME to the
// copies torch bundles -- the runtime that owns the device pointers the planes live at is then the one RCCL moves them
// with; a torch-free process gets /opt/rocm's.
int32_t gs_runtime_info(int32_t load_rccl, char *out, size_t cap)
{
    if (!out || cap == 0) return fail(GS_ERR_INVALID, "null output");
    Dl_info hip_so{}, rccl_so{};
    int hip_version = 0, rccl_version = 0;
    (void)dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &hip_so);
    if (hipRuntimeGetVersion(&hip_version) != hipSuccess) { hip_version = 0; (void)hipGetLastError(); }
    Rccl *R = load_rccl ? rccl() : nullptr;
    if (R) {
        (void)dladdr(reinterpret_cast<const void *>(R->Send), &rccl_so);
        auto get_version = reinterpret_cast<ncclResult_t (*)(int *)>(dlsym(R->handle, "ncclGetVersion"));
        if (get_version) (void)get_version(&rccl_version);
    }
    const char *user = std::getenv("GS_RCCL_LIBRARY");
    std::snprintf(out, cap, "{\"hip\": \"%s\", \"hip_runtime_version\": %d, \"rccl\": %s%s%s, \"rccl_version\": %d, "
                            "\"rccl_named_by_GS_RCCL_LIBRARY\": %s}",
                  hip_so.dli_fname ? hip_so.dli_fname : "", hip_version, rccl_so.dli_fname ? "\"" : "",
                  rccl_so.dli_fname ? rccl_so.dli_fname : "null", rccl_so.dli_fname ? "\"" : "", rccl_version,
                  user && *user ? "true" : "false");
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
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        for (auto *v : {&sl.th0, &sl.th1, &sl.tc0, &sl.tc1})
            for (auto e : *v)
                if (e) (void)hipEventDestroy(e);
        if (sl.copy) { (void)hipStreamSynchronize(sl.copy); (void)hipStreamDestroy(sl.copy); }
        if (sl.stage) (void)hipFree(sl.stage);
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
        GS_HIP_B(hipEventCreateWithFlags(&sl.staged, hipEventDisableTiming));
        GS_HIP_B(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
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
// frames, the channel hash over high address bits), and four 1 GiB planes land on one of three levels for the HBM-bound
// single-step kernel -- 0.66 / 0.70 / 0.75 of 8 TB/s at 16384^2, from one context to the next (profiles/r04_sweeps.md,
// section 8).  What a process CAN do is draw more blocks than it needs and keep the four that read best together.
int32_t gs_fields_place(gs_ctx *ctx, gs_field *const planes[4], int32_t candidates, float *first_ms, float *best_ms)
{
    if (!ctx || !planes) return fail(GS_ERR_INVALID, "null argument");
    if (candidates < 1 || candidates > 12) return fail(GS_ERR_INVALID, "1 to 12 extra candidate blocks, not %d", candidates);
    if (ctx->total_slabs() != 1) return fail(GS_ERR_UNSUPPORTED, "placement by measurement is for single-slab contexts");
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
    const int total = 4 + candidates;
    std::vector<float *> blocks;
    for (int i = 0; i < 4; ++i) blocks.push_back(planes[i]->s[0].alloc);
    auto release = [&](int keep_from) { // frees the blocks from index keep_from on
        for (size_t i = (size_t)keep_from; i < blocks.size(); ++i)
            if (blocks[i]) (void)hipFree(blocks[i]);
        blocks.resize((size_t)keep_from);
    };
    for (int i = 4; i < total; ++i) {
        float *b = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&b), n * sizeof(float)) != hipSuccess) { // fewer candidates: fine
            (void)hipGetLastError();
            break;
        }
        blocks.push_back(b);
    }
    const int have = (int)blocks.size();
    // zeros everywhere (what gs_field_create leaves; the four planes come out zero-filled whichever blocks they get)
    for (float *b : blocks) {
        const hipError_t e = hipMemsetAsync(b, 0, n * sizeof(float), sl.compute);
        if (e != hipSuccess) { release(4); return fail(GS_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)); }
    }
    auto row0_of = [&](float *b) { return b + kGuardFloats + (size_t)kGhostRows * pitch; };
    // a probe: four single steps ping-ponging between (a, b) and (c, d), timed with the context's events
    GsStepArgs base = make_args(ctx, planes[0], planes[1], planes[2], planes[3], 0, 1);
    base.ra0 = 0;
    base.ra1 = base.rows;
    const bool fused = ctx->o.math == GS_MATH_FUSED;
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
        return GS_OK;
    };
    int best[4] = {0, 1, 2, 3};
    float best_t = 0.0f, first_t = 0.0f;
    // the four that are there, then pseudo-random 4-subsets of the pool (a fixed sequence: the same candidates every time)
    uint32_t rng = 0x9e3779b9u;
    const int trials = have > 4 ? 3 * have : 1;
    for (int t = 0; t < trials; ++t) {
        int pick[4] = {0, 1, 2, 3};
        if (t > 0) {
            int order[16];
            for (int i = 0; i < have; ++i) order[i] = i;
            for (int i = 0; i < 4; ++i) { // partial Fisher-Yates
                rng = rng * 1664525u + 1013904223u;
                const int j = i + (int)((rng >> 8) % (uint32_t)(have - i));
                std::swap(order[i], order[j]);
                pick[i] = order[i];
            }
        }
        float ms = 0.0f;
        const int32_t st = probe(pick, &ms);
        if (st != GS_OK) { release(4); return st; }
        if (t == 0) first_t = ms;
        static const bool trace = gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1) != 0;
        if (trace)
            std::fprintf(stderr, "gs_hip placement: blocks %2d %2d %2d %2d (%p %p %p %p): %.4f ms per step\n", pick[0], pick[1], pick[2],
                         pick[3], (void *)blocks[(size_t)pick[0]], (void *)blocks[(size_t)pick[1]], (void *)blocks[(size_t)pick[2]],
                         (void *)blocks[(size_t)pick[3]], ms / 4.0f);
        if (t == 0 || ms < 0.995f * best_t) { best_t = ms; std::memcpy(best, pick, sizeof best); }
    }
    // one sweep of single-block exchanges around the best set found: every member against every block outside it
    if (have > 4) {
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < have; ++b) {
                bool member = false;
                for (int j = 0; j < 4; ++j) member = member || best[j] == b;
                if (member) continue;
                int pick[4];
                std::memcpy(pick, best, sizeof pick);
                pick[i] = b;
                float ms = 0.0f;
                const int32_t st = probe(pick, &ms);
                if (st != GS_OK) { release(4); return st; }
                if (ms < 0.995f * best_t) { best_t = ms; std::memcpy(best, pick, sizeof best); }
            }
    }
    // hand the chosen blocks to the planes; the probes have written into every block: zeros again
    std::vector<float *> chosen(4);
    for (int i = 0; i < 4; ++i) chosen[(size_t)i] = blocks[(size_t)best[i]];
    for (int i = 0; i < 4; ++i) {
        FieldSlab &fs = planes[i]->s[0];
        fs.alloc = chosen[(size_t)i];
        fs.row0 = row0_of(fs.alloc);
        planes[i]->ghost_depth = kGhostRows;
        GS_HIP(hipMemsetAsync(fs.alloc, 0, n * sizeof(float), sl.compute));
    }
    GS_HIP(hipStreamSynchronize(sl.compute));
    for (float *b : blocks) {
        bool used = false;
        for (float *c : chosen) used = used || c == b;
        if (!used) (void)hipFree(b);
    }
    (void)hipGetLastError();
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

namespace {

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
            // the flags belong to the workgroups of the old tiling: start over
            GS_HIP(hipMemsetAsync(w.words, 0, kWindowMaxTiles * sizeof(int32_t), sl.compute));
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
        x.patience = gs_env_int("GS_HIP_WINDOW_PATIENCE", 1 << 21, 1, 1 << 30); // polls of ~1 us each: ~2 s
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
    // Mid-size grids (single slab): K <= 8 steps per launch on LDS-resident windows (gs_run_tile_k), where a
    // pass of the temporally blocked kernel is bound by the length of a wave's march and a launch per <= 4
    // steps.  kernel = auto picks it between the resident kernel's 1536 cells and 1.5 M cells when nothing
    // is pinned, with the window and steps per launch of pick_tile_config (profiles/r02_sweeps.md, section
    // 10: 2.2x at 64 x 128 and 128 x 256, 1.8x at 256 x 512, 1.4x at 512 x 1024; at 1080 x 1920 the marching
    // kernel is ahead again); GS_KERNEL_TILE forces it (tile_shape and fuse_steps then choose the window
    // and the steps per launch).
    const uint64_t cells = u0->rows * u0->cols;
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
    // Grids of one round of register-resident windows (single slab; the reference's default 1080 x 1920 is 252 of them):
    // the whole call is ONE persistent launch of gs_run_window_k, which trades the windows' aprons between workgroups
    // itself every k steps.  kernel = auto takes it from the LDS-window kernel's upper end (1.5 M cells) up to the largest
    // grid that is one workgroup per CU when nothing is pinned and the call is long enough to pay for the launch's fixed
    // cost (64 steps: a launch costs ~9 us plus 4.5 us per step against 4.8 us per step for the marching kernel);
    // GS_KERNEL_WINDOW forces it (fuse_steps = steps per exchange, rows_per_block = window rows: 80 or 96).
    // 461 k against 435 k Mcells x steps / s at 1080 x 1920, both boundary rules (profiles/r04_window_kernel.md).
    if (allow_window && single && cells > 0 && steps > 0 && !ctx->win.disabled) {
        const bool forced = ctx->o.kernel == GS_KERNEL_WINDOW;
        const bool automatic = ctx->o.kernel == GS_KERNEL_AUTO && ctx->o.fuse_steps == 0 && ctx->o.rows_per_block == 0 &&
                               ctx->o.cols_per_lane == 0 && ctx->o.split <= 1 && !ctx->o.use_graph && cells >= kTileAutoCells &&
                               steps >= 64;
        if (forced || automatic) {
            int32_t launched = 0;
            GS_TRY(run_window(ctx, r, steps, forced, &launched, result_slot));
            if (launched) return GS_OK;
        }
    }
    // every other kernel reads or overwrites planes that a window launch still in flight may own
    // every other kernel reads or overwrites planes that a window launch still in flight may own
    GS_TRY(resolve_window(ctx));
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

} // namespace

extern "C" {

int32_t gs_sync(gs_ctx *ctx)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    return sync_all(ctx);
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

int32_t gs_ctx_get_tuned(const gs_ctx *ctx, uint64_t slab_rows, uint64_t cols, int32_t *rows_per_block,
                         int32_t *fuse_steps, int32_t *cols_per_lane, int32_t *share_taps)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    int rpu = 0, k = 0, cpl = 0, share = 0;
    for (const gs_ctx::Tuned &t : ctx->tuned_cache)
        if (t.rows == slab_rows && t.cols == cols) { rpu = t.rpu; k = t.k; cpl = t.cpl; share = t.share ? 1 : 2; } // the newest entry wins
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
        (cols_per_lane != 1 && cols_per_lane != 2 && cols_per_lane != 4) || share_taps < 0 || share_taps > 2)
        return fail(GS_ERR_INVALID, "bad configuration (unit %d rows, %d steps per pass, %d columns per lane, share_taps %d)",
                    rows_per_block, fuse_steps, cols_per_lane, share_taps);
    // keyed like gs_run's own choices: by the steps per pass it was asked to fuse
    const int fuse = ctx->o.fuse_steps > 0 ? (ctx->o.fuse_steps > kGhostRows ? kGhostRows : ctx->o.fuse_steps) : kGhostRows;
    if (fuse_steps > fuse) return fail(GS_ERR_INVALID, "%d steps per pass exceed fuse_steps = %d", fuse_steps, fuse);
    remember_tuned(ctx, gs_ctx::Tuned{slab_rows, cols, fuse, rows_per_block, 1, fuse_steps, cols_per_lane, share_taps == 2 ? 0 : 1});
    return GS_OK;
}

int32_t gs_ctx_comm_info(const gs_ctx *ctx, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *rccl_device)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    int n = 0, r = -1, d = -1;
    if (ctx->comm) {
        Rccl *R = rccl();
        if (!R) return fail(GS_ERR_RCCL, "RCCL is not loaded");
        if (R->CommCount) GS_NCCL(R, R->CommCount(ctx->comm, &n));
        if (R->CommUserRank) GS_NCCL(R, R->CommUserRank(ctx->comm, &r));
        if (R->CommCuDevice) GS_NCCL(R, R->CommCuDevice(ctx->comm, &d));
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = r;
    if (rccl_device) *rccl_device = d;
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
