// gs_internal.h -- what the translation units of libgs_hip.so's host side share: the objects behind the opaque handles
// of include/gs_hip.h (contexts, planes), error plumbing, and the functions that cross file boundaries.
//   gs_api.cpp     contexts, the launch + ghost-row exchange schedule of a pass, gs_step / gs_run, timers, counters
//   gs_fields.cpp  planes: creation, fills, uploads, downloads (blocking and overlapped), colour mapping, placement
//   gs_tuner.cpp   kernel configuration: the launch-geometry model, the on-line tuner of gs_run, gs_ctx_get/set_tuned
//   gs_window.cpp  the persistent window kernel's host side: tiling, exchange planes, give-up and replay
//   gs_rccl.cpp    RCCL (loaded on first use), its self-test, gs_runtime_info, gs_last_error
#pragma once
// (the host-side translation units are compiled with -fvisibility=hidden: only the C ABI leaves the library)
#pragma GCC visibility push(default)
#include "../../include/gs_hip.h"
#pragma GCC visibility pop
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

namespace gsi {

// ---- errors: integer status + a thread-local message (gs_last_error) -------------------------------------------
extern thread_local std::string g_last_error;
int32_t fail(int32_t code, const char *fmt, ...);

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

// ---- RCCL, loaded on first use so that single-process users never touch it ---------------------------------------
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
Rccl *rccl();

#define GS_NCCL(R, expr)                                                                       \
    do {                                                                                       \
        ncclResult_t e_ = (expr);                                                              \
        if (e_ != ncclSuccess)                                                                 \
            return fail(GS_ERR_RCCL, "%s failed: %s", #expr, (R)->GetErrorString(e_));          \
    } while (0)

static_assert(sizeof(ncclUniqueId) == GS_UNIQUE_ID_BYTES, "RCCL unique id size changed");

constexpr int kWindowMaxTiles = 1024; // flags of gs_run_window_k (one per workgroup; a launch has at most one per CU)
constexpr int kGuardFloats = 64; // 256 B in front of / behind every plane
constexpr int kGhostRows = 4;    // ghost rows kept above and below every slab (= max fused steps)

} // namespace gsi

// ---- the objects behind the handles -----------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------
struct SlabRt {
    int device = 0;
    hipStream_t compute = nullptr, halo = nullptr, copy = nullptr;
    hipStream_t copy2 = nullptr;   // the host copies of every other image (gs_field_download_async): two images share the link
    hipStream_t image_stream(int k) const { return k ? copy2 : copy; }
    hipEvent_t done[2] = {nullptr, nullptr}, halod[2] = {nullptr, nullptr};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // asynchronous downloads: two dense device staging buffers used in turn, so that the host copy of one image and the
    // staging copy of the next overlap (copied[k]: the host copy that last read stage[k] is done)
    hipEvent_t staged = nullptr, copied[2] = {nullptr, nullptr};
    float *stage[2] = {nullptr, nullptr};
    size_t stage_floats[2] = {0, 0};
    // gs_ctx_set_pass_timing: per timed pass, events around the halo stream's work (boundary-band kernel +
    // ghost-row exchange: th0, th1) and around the interior kernel on the compute stream (tc0, tc1)
    std::vector<hipEvent_t> th0, th1, tc0, tc1;
    int timed = 0; // passes recorded since the timing was switched on
};

// The form of difference sharing (share_mode, gs_tuner.cpp) of runs that have not been tuned: across lanes too -- never
// slower than sharing within a lane by more than 1 % in any measurement, 3-5 % faster on developed patterns
// (profiles/r05_cross_lane.md).
constexpr int kShareDefault = 2;

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
    uint64_t downloads = 0;                                   // images enqueued with gs_field_download_async so far
    uint64_t place_probes = 0, place_drawn = 0;               // gs_fields_place: pair probes timed, extra blocks drawn (gs_debug_place_stats)
    int pass_timing = 0;                                      // passes per slab still to be timed (0 = off)
    // Configuration of the temporally blocked kernel in force (tuned_rpu > 0): unit height, fused steps
    // per pass and columns per lane for slabs of tuned_rows x tuned_cols -- chosen by gs_run's on-line
    // tuner (single-slab contexts) or handed in through gs_ctx_set_tuned (slab chains).
    uint64_t tuned_rows = 0, tuned_cols = 0;
    int tuned_fuse = 0, tuned_rpu = 0, tuned_split = 0, tuned_k = 0; // tuned_k: fused steps per pass chosen
    int tuned_cpl = 0;                                               // columns per lane chosen
    int tuned_share = kShareDefault;                                 // difference sharing chosen (share_mode: 0 / 1 / 2)
    // every finished choice (a context that alternates between grids does not re-tune)
    struct Tuned { uint64_t rows, cols; int fuse, rpu, split, k, cpl, share; };
    std::vector<Tuned> tuned_cache;
    // Tunings in progress, one per shape (each may span several gs_run calls; two grids driven
    // alternately advance independently).  `batch` / `nb`: timing windows enqueued but not read yet.
    struct Trial { int rpu, V, k, cpl, reps, share; };
    struct Tuning {
        uint64_t rows = 0, cols = 0;
        int fuse = 0, next = 0, best_rpu = 0, best_split = 0, best_k = 0, best_cpl = 0, best_share = kShareDefault;
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
        size_t plane_bytes = 0;
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
        // Images requested with gs_field_download_async while launches were pending: the copy is enqueued behind the
        // launch (nothing waits), and whether the launch gave up is only known when the image is WAITED for -- the abort
        // word travels to `seen` (pinned host memory) behind the image.  If launch `after_seq` or an earlier one gave up,
        // resolve_window fetches the image again right after it has run that launch again.
        struct Image { gs_field *f; float *host; int32_t after_seq; };
        std::vector<Image> images;
        int32_t *seen = nullptr; // pinned, two words: the abort word as either image stream last saw it
    } win;
    int share_now = kShareDefault; // form of difference sharing in force when gs_options.share_taps leaves the choice open (share_mode)
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

namespace gsi {

struct Run; // gs_run: state of one call (below)

// gs_api.cpp
int32_t same_shape(const gs_field *a, const gs_field *b);
int32_t sync_all(gs_ctx *ctx);
int32_t refresh_ghosts(gs_ctx *ctx, gs_field *f);
int min_slab_rows(const gs_ctx *ctx, const gs_field *f);
GsStepArgs make_args(const gs_ctx *ctx, const gs_field *in_u, const gs_field *in_v,
                     const gs_field *out_u, const gs_field *out_v, int i, int fuse);
int32_t join_bands(gs_ctx *ctx, hipStream_t stream);
int bands_for(const gs_ctx *ctx, const gs_field *f, int fuse);
int32_t step_bands(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v, int fuse, int V);
int32_t step_impl(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v,
                  int fuse = 1);
int32_t check_step_fields(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v);
int32_t run_steps(gs_ctx *ctx, gs_field *u0, gs_field *v0, gs_field *u1, gs_field *v1, uint64_t steps, int32_t *result_slot,
                  bool allow_window);
// gs_tuner.cpp
long tb_strips(int32_t cols, int fuse, int cpl);
int share_mode(const gs_ctx *ctx);
int fast_of(const gs_ctx *ctx);
int fit_heights(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl, int fast, int *out, int max, bool partial = false);
bool tuned_for(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse);
int32_t pick_cols_per_lane(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse);
int32_t pick_rows_per_unit(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse);
int32_t model_rows_per_unit(const gs_ctx *ctx, int32_t rows, int32_t cols, int fuse, int cpl);
void pick_tile_config(long rows, long cols, int *shape, int *k);
uint64_t slab_rows_of(const gs_field *f);
bool same_slab_shape(const gs_ctx *ctx, uint64_t rows_a, uint64_t cols_a, uint64_t rows_b, uint64_t cols_b);
bool tuned_shape(const gs_ctx *ctx, const gs_field *f, int fuse);
void recall_tuned(gs_ctx *ctx, const gs_field *f, int fuse);
void remember_tuned(gs_ctx *ctx, const gs_ctx::Tuned &t);
int32_t tune_online(Run &r, int fuse);
// kernel = auto runs the LDS-window kernel (gs_run_tile_k) below this many cells:
constexpr uint64_t kTileAutoCells = 1536 * 1024; // above, the marching kernel is ahead (1080 x 1920: 380-420 k vs 350 k)
// ... and the persistent window kernel (gs_run_window_k) from this many cells on, in calls of >= kWindowAutoSteps steps, where one
// round of windows covers the grid: 331 k against 236 k at 720 x 1280, 374 k against 297 k at 1024 x 1024, 509 k against 316 k at
// 900 x 1600 in 1000-step calls (273 / 220, 305 / 269, 407 / 290 in 64-step calls); at 512 x 1024 the LDS-window kernel is
// ahead, 242 k against 188 k (profiles/r06_logs/window_small_grids_k.log)
constexpr uint64_t kWindowAutoCells = 800 * 1024;
constexpr uint64_t kWindowAutoSteps = 32; // (the reference's steps per image: profiles/r06_window_kernel.md)
// gs_window.cpp
std::vector<GsWindowDesc> plan_windows(const gs_ctx *ctx, uint64_t rows, uint64_t cols, int want_rpw, int want_k, int *rpw_out, int *k_out);
std::vector<GsWindowDesc> plan_windows(int cu_count, bool zero_halo, bool cheap, uint64_t rows, uint64_t cols, int want_rpw, int want_k,
                                       int *rpw_out, int *k_out);
int32_t ensure_window_rt(gs_ctx *ctx, const gs_field *f);
int32_t resolve_window(gs_ctx *ctx);
int32_t run_window(gs_ctx *ctx, Run &r, uint64_t steps, bool forced, int32_t *launched, int32_t *result_slot);

// ---- gs_run: state of one call -------------------------------------------------------------------------------------
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

} // namespace gsi
