/*
 * gs_hip.h -- C ABI of the MI355X-native Gray-Scott compute backend (libgs_hip.so).
 *
 * This is the drop-in boundary for ONE path of HadrienG2/grayscott: the per-timestep
 * 9-point Laplacian + u*v^2 reaction update that every compute backend of the reference
 * implements behind
 *     SimulateBase / SimulateCreate / Simulate      compute/shared/src/lib.rs:19-58
 *     Concentration / Species                       data/src/concentration/mod.rs:17-296
 * Results follow the reference's *naive* backend (compute/naive/src/lib.rs:42-83) bit for
 * bit, including its clipped-window boundary rule and FTZ-without-DAZ denormal handling.
 *
 * The reference is Rust; a backend crate binds these entry points through `extern "C"`
 * (rust/compute_hip/src/ffi.rs, shown in INTEGRATION.md).  Plain pointers, sizes and
 * integer status codes only: no C++ / torch / HIP types cross this boundary.
 *
 * Conventions
 *   - Every function returns GS_OK (0) or a negative gs_status; gs_last_error() returns a
 *     thread-local, human-readable message for the last failure on the calling thread.
 *     No exceptions, aborts or panics cross the ABI.  (The reference's error contract:
 *     `type Error: Error + From<C::Error> + Send + Sync`, compute/shared/src/lib.rs:31.)
 *   - Handles are created and destroyed by the caller.  The library never keeps a host
 *     pointer past the call that received it.
 *   - Calls on one gs_ctx must be externally serialised (the reference takes
 *     `&mut Species` in perform_steps); a context may be moved between threads: every
 *     entry point selects its own device(s).
 *   - gs_step / gs_run only enqueue work; gs_sync (or a download) waits for it.  A host-side
 *     Simulate::perform_steps is gs_run + gs_sync (every backend of the reference returns from
 *     perform_steps with the steps done: compute/shared/src/gpu/mod.rs:77-91); gs_run alone is
 *     the asynchronous SimulateGpu::prepare_steps (:70-75).
 *   - Shapes are [rows, cols] in scalar units, as in Concentration::shape()
 *     (data/src/concentration/mod.rs:191-221).  Storage is row-major f32 (`Precision`,
 *     data/src/lib.rs:11).
 *
 * Domain decomposition
 *   A context owns `n_local` row slabs (one per entry of device_ids; ids may repeat) which
 *   together cover the rows of this process; `world` processes (one per GPU when launched
 *   under torchrun) cover the global grid in rank order.  Slabs keep 4 ghost rows above and
 *   below; a pass that fuses K <= 4 time steps updates the K boundary rows of each side
 *   first and pushes them to the neighbouring slab's ghost rows -- by a device-to-device copy
 *   inside a process, by RCCL ncclSend/ncclRecv between processes -- on a side stream,
 *   overlapped with the interior update.  The reference has no multi-device code; its
 *   in-process precedent is SimulateCpu::split_grid (compute/shared/src/cpu.rs:111-154).
 *
 * Environment
 *   The library itself reads these variables (the host mirrors add one per gs_options field, GS_HIP_<FIELD>, as the
 *   reference's CliArgs do with clap's `env`).  None of them changes results: every combination is bit-identical.
 *     GS_RCCL_LIBRARY       library to bind instead of librccl (a custom RCCL build; the tests' shared-memory
 *                           transport double).  An explicit choice never falls back to the system's librccl.
 *     GS_HIP_TRACE_LAUNCH   1 = print the first 64 kernel launches (label, row ranges, layout) on stderr
 *     GS_HIP_TRACE_TUNER    1 = print every timing window of gs_run's on-line tuner and what it chose, and every probe
 *                           of gs_fields_place
 *   Launch-policy switches for A/B timing (defaults are the measured best; grayscott_amd/csrc/gs_experiments.h):
 *     GS_HIP_PLACE_ALL      1 = gs_fields_place draws all its candidates even when two fast pairs are found before
 *     GS_HIP_PLACE_DEEP     0 = gs_fields_place never draws more than `candidates` blocks (default: up to 4 x as many while more
 *                           than half of the device's memory is free)
 *     GS_HIP_PLACE_FORCE    "a,b,c,d" = test hook: gs_fields_place draws its candidates and moves the planes to blocks a, b, c, d
 *                           of those it holds (0-3 the planes' own, 4 and up drawn), without probes
 *     GS_HIP_EDGE_KINDS     0 = edge units of the marching kernel all take the general path
 *     GS_HIP_EDGE_SPLIT     0 / 1 = never / always dispatch edge units as two half-height units
 *     GS_HIP_FAIR           0 / 1 = never / always run one-round launches as in-step 16-wave workgroups
 *     GS_HIP_FAIR_FROM      progress (0 ... 256) from which the in-step form steers wave priorities
 *     GS_HIP_XCD_M          0 = plain workgroup order, n = XCD-aware renumbering in groups of 8 n workgroups
 *     GS_HIP_XCD_M_STREAM   the same for the single-step kernel
 *     GS_HIP_TILE_LDS_FLOOR least dynamic LDS (bytes) of the LDS-window kernel: limits its workgroups per CU
 *     GS_HIP_WINDOW_PATIENCE polls (2-3 us each) a wave of the persistent window kernel waits for its neighbours' cells
 *                           before the launch gives up (default 2^20: 2-3 s)
 *     GS_HIP_WINDOW_WAVES   "left,interior,right": waves in use (of 16) in the windows of the grid's left-most, inner
 *                           and right-most tile column (default 12,16,12 under the clipped rule, 16,16,16 otherwise)
 */
#ifndef GS_HIP_H
#define GS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 4

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID = -1,     /* bad argument / shape mismatch (the reference panics: mod.rs:291-295) */
    GS_ERR_HIP = -2,         /* a HIP runtime call failed                                  */
    GS_ERR_RCCL = -3,        /* librccl could not be loaded or an RCCL call failed         */
    GS_ERR_NO_DEVICE = -4,   /* no usable gfx950 device                                    */
    GS_ERR_UNSUPPORTED = -5, /* request outside what this build implements                 */
    GS_ERR_NOMEM = -6
} gs_status;

/* Parameters (data/src/parameters.rs:13-33): stencil weights row-major + the five rates. */
typedef struct gs_params {
    float w[3][3];
    float du;   /* diffusion_rate_u */
    float dv;   /* diffusion_rate_v */
    float feed; /* feed_rate        */
    float kill; /* kill_rate        */
    float dt;   /* time_step        */
} gs_params;

/* Arithmetic flavour of the step kernels.
 *   GS_MATH_STRICT  every reference operation is a separately rounded f32 op and the
 *                   kernels run with f32 denormal mode "flush results, keep inputs", which
 *                   is what MXCSR.FTZ (DenormalsFlusher, compute/shared/src/lib.rs:161-180)
 *                   does on the CPU: bit-identical to naive under FTZ, sub-normals included.
 *   GS_MATH_FUSED   the eight tap updates per species use one FMA each (exact for the
 *                   power-of-two Oono-Puri weights as long as the product is a normal
 *                   number) and denormals are kept: bit-identical to naive wherever no
 *                   intermediate is sub-normal, |diff| <= 1e-37 elsewhere.  Refused
 *                   (GS_ERR_UNSUPPORTED) for weights that are not 0 or a power of two.    */
typedef enum gs_math { GS_MATH_STRICT = 0, GS_MATH_FUSED = 1 } gs_math;

/* Which step kernel runs.  AUTO = STREAM for a single gs_step; inside gs_run, by grid size when nothing
 * is pinned: the LDS-resident whole-run kernel up to 1536 cells; WINDOW for calls of >= 32 steps on grids
 * from 0.8 M cells that one round of 80-row windows covers (one per compute unit; 8, 6, 4 or 2 steps per
 * exchange, as many as fit); TILE otherwise up to 1.5 M cells; TB with fuse_steps (default 4) otherwise, for
 * slab chains and whenever fuse_steps, rows_per_block, cols_per_lane, split or use_graph pin a schedule. */
typedef enum gs_kernel {
    GS_KERNEL_AUTO = 0,    /* best measured variant for the shape                          */
    GS_KERNEL_SIMPLE = 1,  /* one thread per cell, global loads only (cross-check kernel)  */
    GS_KERNEL_STREAM = 2,  /* register sliding window, 16-B loads, DPP halo exchange       */
    GS_KERNEL_TB = 3,      /* temporally blocked streaming kernel: fuse_steps steps / launch */
    GS_KERNEL_LDS = 4,     /* LDS-staged (tile + halo) window, one step per launch (measured
                              alternative to STREAM; never chosen by AUTO)                   */
    GS_KERNEL_TILE = 5,    /* gs_run only, single slab: up to 8 steps per launch on LDS-resident windows
                              with a K-cell apron, one cell per lane and 16 waves per window (gs_step and
                              slab chains fall back to STREAM / TB); what AUTO runs on mid-size grids  */
    GS_KERNEL_WINDOW = 6   /* gs_run only, single slab, grids of at most one register-resident window per compute unit
                              (1.5 - 2.3 M cells on 256 CUs: the reference's default 1080 x 1920 is 252 windows): the
                              whole call is ONE persistent launch; every workgroup keeps its window (72 x 120 owned
                              cells + a k-cell apron; lower windows on the grid's left and right edge, whose cells cost
                              more) in registers and trades its apron with its neighbours every k steps through
                              exchange planes, flags and sc1 accesses (fuse_steps = k: 2, 4, 6 or 8; rows_per_block =
                              full window rows: 80).  What AUTO runs for calls of >= 32 steps where such windows cover the
                              grid: 496 k against TB's 395 k Mcells x steps / s at 1080 x 1920 in 1000-step calls; in
                              32-step calls the two tie (profiles/r05_window_kernel.md).
                              A launch whose workgroups are not all resident (another long-running kernel holds CUs)
                              gives up after a bounded wait: the launches before it stand, it and the later ones are run
                              again with TB by the next call that waits for or reads results (each from its own input
                              planes, which no launch writes), and the context stays with TB  */
} gs_kernel;

/* Rule on the edges of the global grid.  The reference has two:
 * CLIPPED   -- compute_naive's, the parity target: the 3x3 window is clipped to the grid and the
 *              weights are indexed from the clipped window's top-left corner
 *              (compute/naive/src/lib.rs:57-71);
 * ZERO_HALO -- the Vulkan and SIMD backends': full window, weights centred, cells outside the grid
 *              read as 0 (compute/gpu/naive/src/pipeline.rs:105-113, main.comp:37-44;
 *              data/src/concentration/simd/mod.rs:281-326), here with naive's operation order. */
enum gs_boundary { GS_BOUNDARY_CLIPPED = 0, GS_BOUNDARY_ZERO_HALO = 1 };

/* Backend options: the C view of the Rust `CliArgs` (compute/shared/src/lib.rs:20-25 --
 * every field has a default; zero-initialise and override). */
typedef struct gs_options {
    int32_t math;            /* gs_math; default STRICT                                    */
    int32_t kernel;          /* gs_kernel; default AUTO                                    */
    int32_t rows_per_block;  /* rows each wave marches over (0 = auto)                     */
    int32_t fuse_steps;      /* steps fused per launch in gs_run (1..4; 0 = auto); single slab */
    int32_t use_graph;       /* 1 = gs_run replays batches of 16 passes through a hipGraph: one host-side *
                              * launch per batch (single slab, no row bands; 0 = off)                   */
    int32_t pitch_pad;       /* extra f32 of row pitch beyond the 64-float round-up        */
    int32_t split;           /* row bands a single slab is scheduled as (0 or 1 = off): adjacent  *
                              * bands only depend on each other's K boundary rows, so the tail *
                              * of one pass overlaps the start of the next.  Opt-in: the gain  *
                              * (up to +3 % at 16384^2) is not stable from box to box          */
    int32_t general_kernels; /* 1 = never use the kernel variants specialised for the default  *
                              * side weights (0.5) / dt == 1; results are bit-identical either *
                              * way, the switch exists for A/B timing and tests                */
    int32_t cols_per_lane;   /* columns per lane of the temporally blocked kernel: 4 (wide, for  *
                              * large grids), 2 or 1 (more, narrower waves for small grids);   *
                              * 0 = chosen on line by gs_run                                   */
    int32_t boundary;        /* gs_boundary; default CLIPPED                                     */
    int32_t no_tune;         /* 1 = gs_run never times candidate configurations: it runs the pinned *
                              * values above, a configuration set with gs_ctx_set_tuned, or the     *
                              * untuned defaults                                                   */
    int32_t tile_shape;      /* GS_KERNEL_TILE: window of a workgroup, 1 = 32 rows x 64 columns, 2 = 16 x 64,       *
                              * 3 = 64 x 64; 0 = 32 x 64.  With kernel = TILE, fuse_steps (1..8, and less than  *
                              * half the window's rows) sets the steps per launch                          */
    int32_t share_taps;      /* full difference sharing in the temporally blocked kernel (the S / SE / SW taps of a  *
                              * row are, negated, the N / NW / NE taps of the next: 46 instead of 52 arithmetic    *
                              * instructions per cell-step, bit-identical; needs side weights 0.5, dt == 1 and      *
                              * w[0][0] == w[2][2], w[0][2] == w[2][0] -- true of every stencil of the reference):  *
                              * 0 = on (form 3) unless gs_run's on-line tuner measures the chosen configuration   *
                              * faster without, 1 = on, WITHIN a lane only (46 instructions per cell-step; the halo  *
                              * columns of a lane's two go through an LDS board), 2 = off, 3 = on and ACROSS lanes   *
                              * too (the three differences that cross a lane boundary are formed by one of the two  *
                              * lanes and read by the other as DPP operands: 41 instructions per cell-step, half the *
                              * LDS traffic; 3-5 % less energy per cell-step than form 1 on every input)             */
    int32_t reserved[3];
} gs_options;

typedef struct gs_ctx gs_ctx;     /* devices, streams, row partition, RCCL communicator    */
typedef struct gs_field gs_field; /* one f32 plane [rows, cols], slab-distributed           */

/* Parameters::default() (parameters.rs:72-83) with the Oono-Puri weights (:116-122). */
void gs_default_params(gs_params *out);
void gs_default_options(gs_options *out);

int32_t gs_abi_version(void);
const char *gs_last_error(void);
int32_t gs_device_count(int32_t *out);

/* 128-byte RCCL unique id, created on rank 0 and handed to every rank out of band. */
#define GS_UNIQUE_ID_BYTES 128
int32_t gs_get_unique_id(void *out128);

/* Is RCCL usable from this process?  Loads the library exactly as a multi-process context does (GS_RCCL_LIBRARY,
 * else librccl), creates a ONE-rank communicator on `device` and moves a message of `floats` f32 to itself with the
 * call pattern of the ghost-row exchange (one group: ncclSend + ncclRecv, on a high-priority stream), then compares
 * it.  For a maintainer's first multi-GPU run; no context is needed. */
int32_t gs_rccl_selftest(int32_t device, uint64_t floats);

/* Which libraries this process's libgs_hip.so is bound to, as one JSON object: {"hip": path of the HIP runtime,
 * "hip_runtime_version", "rccl": path or null, "rccl_version", "rccl_named_by_GS_RCCL_LIBRARY"}.  libgs_hip.so links
 * the HIP runtime by SONAME (libamdhip64.so.7) and dlopens RCCL by SONAME (librccl.so.1) on first use, so it binds
 * WHATEVER COPY THE PROCESS HAS MAPPED FIRST: in a process that imported torch before creating a context -- bench.py,
 * the tests -- both are the copies the torch wheel bundles (one HIP runtime in the process, the one that owns
 * the planes' device pointers, and the RCCL built against it, which the library's communicator then shares with
 * torch's ProcessGroupNCCL if that exists); in a torch-free process -- the Rust binary -- they are /opt/rocm's.  Both
 * pairings run the one-rank exchange of gs_rccl_selftest in the GPU suite (tests/test_gpu_multiprocess.py).
 * load_rccl = 0 reports RCCL only if this process has loaded it already (nothing is loaded for the answer). */
int32_t gs_runtime_info(int32_t load_rccl, char *out, size_t cap);

/* SimulateCreate::new(params, args) (compute/shared/src/lib.rs:42-45).
 *   device_ids / n_local : local slabs, top to bottom (NULL / 0 = one slab on device 0)
 *   rank, world          : this process's place in the row-wise chain (0, 1 = single process)
 *   unique_id            : gs_get_unique_id() bytes of rank 0; required iff world > 1      */
int32_t gs_ctx_create(gs_ctx **out, const gs_params *params, const gs_options *opts,
                      const int32_t *device_ids, int32_t n_local, int32_t rank, int32_t world,
                      const void *unique_id);
int32_t gs_ctx_destroy(gs_ctx *ctx);
int32_t gs_ctx_set_params(gs_ctx *ctx, const gs_params *params);

/* Concentration::default / zeros / ones (concentration/mod.rs:205-218) = create (+ fill).
 * `rows`, `cols` are the GLOBAL shape; every process passes the same values. */
int32_t gs_field_create(gs_ctx *ctx, gs_field **out, uint64_t rows, uint64_t cols);
int32_t gs_field_destroy(gs_ctx *ctx, gs_field *f);
int32_t gs_field_shape(const gs_field *f, uint64_t *rows, uint64_t *cols);
/* Rows [row0, row1) of the global grid that this process stores. */
int32_t gs_field_local_rows(const gs_field *f, uint64_t *row0, uint64_t *row1);
/* Concentration::raw_shape (mod.rs:223-228): local rows incl. 2 x 4 ghost rows per slab, row
 * pitch in f32. */
int32_t gs_field_raw_shape(const gs_field *f, uint64_t *raw_rows, uint64_t *pitch);

int32_t gs_field_fill(gs_ctx *ctx, gs_field *f, float value);
/* Concentration::fill_slice (mod.rs:230-243): global half-open ranges; rows outside this
 * process's slabs are skipped. */
int32_t gs_field_fill_slice(gs_ctx *ctx, gs_field *f, uint64_t r0, uint64_t r1, uint64_t c0,
                            uint64_t c1, float value);
/* Concentration::finalize (mod.rs:245-253): make the plane usable as a step input, i.e.
 * refresh ghost rows after fill / fill_slice / upload.  gs_step does it on demand. */
int32_t gs_field_finalize(gs_ctx *ctx, gs_field *f);
/* Dense row-major host <-> device copies of this process's rows (blocking).  `host` points
 * at local row 0, i.e. global row `row0` of gs_field_local_rows.  download =
 * Concentration::write_scalar_view (mod.rs:277-288). */
int32_t gs_field_upload(gs_ctx *ctx, gs_field *f, const float *host);
int32_t gs_field_download(gs_ctx *ctx, gs_field *f, float *host);
/* Device address of local slab `slab`'s row 0 (for zero-copy consumers); *pitch in f32. */
int32_t gs_field_device_ptr(const gs_field *f, int32_t slab, void **ptr, uint64_t *pitch,
                            uint64_t *slab_row0, uint64_t *slab_rows, int32_t *device);

/* Tell the library that the caller has written cells of `f` through gs_field_device_ptr (a zero-copy
 * producer): the copies of its boundary rows in the neighbouring slabs' ghost rows are stale, exactly as
 * after gs_field_upload.  The caller orders its writes before the next library call itself (the library's
 * streams do not know about them). */
int32_t gs_field_mark_written(gs_ctx *ctx, gs_field *f);

/* One time step: reads (in_u, in_v), writes (out_u, out_v).  Asynchronous.  The caller
 * flips its handles afterwards, as Species::flip does (concentration/mod.rs:88-92). */
int32_t gs_step(gs_ctx *ctx, gs_field *in_u, gs_field *in_v, gs_field *out_u, gs_field *out_v);

/* Simulate::perform_steps (compute/shared/src/lib.rs:48-58): `steps` steps ping-ponging
 * between slot 0 (u0, v0: input on entry) and slot 1.  *result_slot receives the slot that
 * holds the newest state (steps odd -> 1); the caller swaps its handles accordingly so that
 * "the input concentrations contain the final results" (:51-52).  Asynchronous.  The first runs on
 * a shape time a few candidate configurations (unit height, steps per pass, columns per lane) on
 * passes of the simulation itself -- nothing is recomputed.  A call with fewer than 64 steps still
 * to go never waits for those timings (it reads them in a later call); a longer one waits for each
 * phase, so that a long first run is tuned when it returns.  gs_options.no_tune (or pinning
 * rows_per_block) switches the tuning off. */
int32_t gs_run(gs_ctx *ctx, gs_field *u0, gs_field *v0, gs_field *u1, gs_field *v1,
               uint64_t steps, int32_t *result_slot);

/* Placement by measurement, for the planes of a large Species on a context with one slab per process (the host mirrors'
 * make_species call it for every Species of >= 2^26 cells unless told not to).  Where a hipMalloc lands in HBM is below
 * what a process controls, and it matters: the blocks lie in a few physical regions ("groups", runs of 2-30 consecutive
 * 1 GiB allocations; one large allocation is always inside one), and two planes of ONE group that a pass writes (and
 * reads) together are slow -- a pass that reads two 1 GiB blocks and writes them back takes 0.86-0.96 ms within a group,
 * 0.72-0.79 ms across groups, whatever the offsets, while every block alone reads and writes at the same rate from
 * every XCD (tools/ubench/hbm_kinds.hip; profiles/r06_placement.md).  Four planes of one group run the HBM-bound
 * single-step kernel at 0.58-0.65 of 8 TB/s, U's planes in one group and V's in another at 0.73-0.76, and the marching
 * kernel of gs_run gains 8-12 % at 16384^2.  This call times that pass (which leaves the blocks' contents alone) over
 * the pairs among the planes' four blocks (6 probes of 3 passes, 20 ms at 16384^2).  If each slot's (U, V) pair is as
 * fast as the fastest pair seen, and a slower pair has been seen, nothing moves and nothing is allocated.  Else it
 * draws blocks of the planes' size ONE AT A TIME -- at most `candidates` (1..124; the hosts' default is 12: at most
 * 12 GiB held for a moment at 16384^2) --, times each against every block held, stops as soon as two disjoint fast pairs
 * exist, moves the planes that have to move (device copies, one at a time: the planes KEEP THEIR CONTENTS, also when a
 * copy fails) and frees the rest.  A fresh box can hand out 16 and more consecutive blocks of one region: if `candidates`
 * draws do not settle it and MORE THAN HALF of the device's memory is free, the search goes on to 4 x `candidates` blocks
 * with one probe per block (planes of >= 512 MiB; GS_HIP_PLACE_DEEP=0 switches it off).  It waits for the context's work first and can come at any time.  first_ms / best_ms
 * (optional): mean time of the probe pass over the two slots' (U, V) pairs, before and after (0 when nothing was done). */
int32_t gs_fields_place(gs_ctx *ctx, gs_field *const planes[4], int32_t candidates, float *first_ms, float *best_ms);

/* Wait for everything enqueued on this context (all local devices and streams). */
int32_t gs_sync(gs_ctx *ctx);

/* Overlapped result download -- the analogue of ImageConcentration::write_scalar_view_after /
 * make_scalar_view_after (data/src/concentration/gpu/image/mod.rs:183-206), which `simulate`
 * uses so that the N steps and the download of the result are one asynchronous submission
 * (simulate/src/main.rs:99-106).
 *   gs_host_alloc / gs_host_free   page-locked host memory for the images
 *   gs_field_download_async        enqueue "copy this process's rows of `f` to `host`" behind
 *                                  the work already enqueued and return at once.  The plane is
 *                                  first densified into a device staging buffer, so steps
 *                                  enqueued afterwards are not held back by PCIe; `host` must
 *                                  stay valid until gs_download_wait / gs_sync.  It never waits: behind
 *                                  a persistent window launch (1080 x 1920 in long calls), which may
 *                                  still give up, the image is validated when it is waited for -- a
 *                                  launch that gave up is then run again and the image fetched again.
 *   gs_download_wait               wait for the downloads enqueued so far (not for later steps)
 *   gs_download_wait_but           ... for all but the newest `in_flight` (0 or 1) of them: with two images in flight
 *                                  (two staging buffers are used in turn) the host copy of one image overlaps the
 *                                  staging of the next and the hand-over of the one before: the PCIe link stays busy */
int32_t gs_host_alloc(void **out, uint64_t bytes);
int32_t gs_host_free(void *p);
int32_t gs_field_download_async(gs_ctx *ctx, gs_field *f, float *host);
int32_t gs_download_wait(gs_ctx *ctx);
int32_t gs_download_wait_but(gs_ctx *ctx, int32_t in_flight);

/* The per-pixel work of data-to-pics (data-to-pics/src/main.rs:139-144, ui/src/lib.rs:113-123): paint this
 * process's rows of `f` (the reference paints the V plane) into dense RGB8 [rows, cols, 3] through a
 * palette of n_colors RGB triples: pixel = palette[clamp(floor((double)(scale * value) * n_colors), 0,
 * n_colors - 1)], NaN -> entry 0 -- the rule of colorous' sequential gradients (the reference uses
 * colorous::INFERNO with scale = AMPLITUDE_SCALE = 1 / 0.5).  The palette is data: a binding passes the
 * 256 entries of the gradient it wants.  Blocking, like gs_field_download. */
int32_t gs_field_colormap(gs_ctx *ctx, gs_field *f, float scale, const uint8_t *palette_rgb, int32_t n_colors,
                          uint8_t *host_rgb);

/* Device-side stopwatch on the context's compute stream(s) (HIP events): start/stop
 * bracket enqueued work; elapsed is the maximum over local slabs, in milliseconds. */
int32_t gs_timer_start(gs_ctx *ctx);
int32_t gs_timer_stop(gs_ctx *ctx, float *elapsed_ms);

/* The configuration gs_run uses for slabs of `slab_rows` x `cols` cells: unit height, steps fused per
 * pass, columns per lane, full difference sharing (1 = on, 2 = off, 3 = across lanes too, as gs_options.share_taps; _set_ takes 0 as 3)
 * (zeros from _get_ when nothing was chosen yet).  Single-slab contexts find it
 * themselves (on-line tuning inside gs_run); a slab chain takes what it is given: one process tunes on a
 * single slab of the slab's shape, reads the result with _get_ and every process of the chain sets it
 * with _set_ -- all of them the same values, since the ghost-row exchange is fuse_steps rows deep
 * (grayscott_amd/dist.py: share_tuning). */
int32_t gs_ctx_get_tuned(const gs_ctx *ctx, uint64_t slab_rows, uint64_t cols, int32_t *rows_per_block,
                         int32_t *fuse_steps, int32_t *cols_per_lane, int32_t *share_taps);
int32_t gs_ctx_set_tuned(gs_ctx *ctx, uint64_t slab_rows, uint64_t cols, int32_t rows_per_block,
                         int32_t fuse_steps, int32_t cols_per_lane, int32_t share_taps);

/* What RCCL itself reports for this context's communicator (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice): the number of ranks, this rank and its device; 0, -1, -1 for a single process. */
int32_t gs_ctx_comm_info(const gs_ctx *ctx, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *rccl_device);

/* Counters of a context since its creation, and -- on slab chains -- where the time of the passes timed
 * with gs_ctx_set_pass_timing went.  Not part of the reference's interface: what bench.py and the tests
 * read instead of inferring it from launch counts. */
typedef struct gs_stats {
    uint64_t passes;          /* passes over the planes enqueued (one pass advances 1..8 time steps)        */
    uint64_t steps;           /* time steps enqueued                                                       */
    uint64_t launches;        /* step-kernel launches (slab chains: boundary band + interior per slab)     */
    uint64_t ghost_refreshes; /* blocking ghost-row refreshes (slab chains: after fill / upload, or when a  *
                               * pass needs deeper ghost rows than the previous one left)                  */
    uint64_t timed_passes;    /* passes covered by the three sums below (the slowest local slab's)         */
    float halo_ms;            /* halo stream: boundary-band kernel + ghost-row exchange, summed             */
    float interior_ms;        /* compute stream: interior kernel, summed                                   */
    float halo_exposed_ms;    /* sum over passes of max(0, end of the halo stream's work - end of the      *
                               * interior kernel): what the exchange did NOT hide behind the interior      */
    float reserved;
    uint64_t window_fallbacks; /* persistent window launches that gave up (another kernel held compute units) and were  *
                                * run again by the marching kernel: 0 or 1, the context stays with the marching kernel  *
                                * afterwards; a timing that contains one measured the stall, not a rate               */
} gs_stats;
int32_t gs_ctx_stats(gs_ctx *ctx, gs_stats *out);
/* Time the next `passes` passes (0..4096; 0 = off) of every local slab of a slab chain with HIP events on
 * the halo and compute streams; gs_ctx_stats waits for them and reports the sums.  Waits for enqueued work. */
int32_t gs_ctx_set_pass_timing(gs_ctx *ctx, int32_t passes);

/* Introspection for tests and the bench: name of the kernel variant last launched
 * ("tb-k4/strict@32x2" = 4 fused steps, strict math, tuned: 32-row units, 2 row bands) and the number of
 * kernel launches so far. */
int32_t gs_ctx_info(const gs_ctx *ctx, char *kernel_name, size_t cap, uint64_t *launches);

/* Measurement hook, not for bindings (tools/rccl_under_load.py): the ghost-row exchange's transport on ONE GPU while the
 * caller keeps the chip busy or idle.  mode 0: a one-rank RCCL communicator, `messages` ncclSend / ncclRecv pairs of
 * `floats` f32 to itself in one group; mode 1: the same bytes as device-to-device copies (the in-process chain's route);
 * both on a high-priority stream created like a slab's halo stream.  _run enqueues one exchange and waits for it:
 * host_ms from the first enqueue to the end of the wait, device_ms between events around it on its stream. */
typedef struct gs_exchange_probe gs_exchange_probe;
int32_t gs_debug_exchange_probe_create(int32_t device, int32_t mode, int32_t messages, uint64_t floats, gs_exchange_probe **out);
int32_t gs_debug_exchange_probe_run(gs_exchange_probe *p, float *host_ms, float *device_ms);
int32_t gs_debug_exchange_probe_destroy(gs_exchange_probe *p);
/* Introspection for the bench and the tests, not for bindings: what gs_fields_place has done on this context so far --
 * pair probes timed and extra blocks drawn (each of the planes' size; all freed or handed to planes by now). */
int32_t gs_debug_place_stats(const gs_ctx *ctx, uint64_t *probes, uint64_t *blocks_drawn);
/* Test hook, not for bindings: the key of the table that remembers on which (device, kernel entry) more than 64 KB
 * of dynamic LDS were opted into -- 1 when (device, slot, bytes) is new (and is recorded), 0 when a launch on that
 * device would skip the opt-in, -1 for a bad slot (tests/test_capi_cpu.py). */
int32_t gs_debug_dyn_lds_key(int32_t device, int32_t slot, int32_t bytes);
/* Test hook, not for bindings: the tiling GS_KERNEL_WINDOW would use for a grid on a device of `compute_units` CUs (no
 * device needed).  Returns the number of windows (0: the grid is not one round of windows) and writes up to cap_windows
 * descriptors of 6 + 14 int32 each: first owned row, first owned column, owned rows, owned columns, window rows in use,
 * number of neighbours, neighbour indices. */
int32_t gs_debug_window_plan(uint64_t rows, uint64_t cols, int32_t compute_units, int32_t boundary, int32_t cheap_edge_kinds,
                             int32_t window_rows, int32_t k, int32_t *out, int32_t cap_windows, int32_t *rows_per_wave, int32_t *k_out);

#ifdef __cplusplus
}
#endif
#endif /* GS_HIP_H */
