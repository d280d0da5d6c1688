// gs_kernels.h -- host/device contract between the host side (gs_api.cpp, gs_tuner.cpp, gs_window.cpp, gs_fields.cpp) and the gfx950 kernels.
//
// One "plane" is a row-major f32 array of one species in one slot for one row slab:
//   element (r, c) of the slab, r in [-ghost, rows + ghost) (rows outside [0, rows) = ghost rows),
//   c in [0, pitch), lives at  base + r * pitch + c ;  pitch % 64 == 0 (256-B rows).
// Columns [cols, pitch) are padding: readable, writable, never used as neighbours.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Largest grid (cells) gs_launch_resident_* takes (4 planes of (rows + 2) x (cols + 2) floats in LDS: at most
// 74 KB).  Above, the LDS-window kernel with its many workgroups is faster (profiles/archive/r02_sweeps.md, section 10).
constexpr int kGsResidentCells = 1536;
// gs_launch_tile_*: the most time steps one launch advances its tiles by.
constexpr int kGsTileMaxSteps = 8;

struct GsStepArgs {
    const float *in_u, *in_v; // local row 0, col 0 of the input planes
    float *out_u, *out_v;     // same for the output planes
    int32_t rows;             // rows owned by this slab
    int32_t cols;             // valid columns
    int32_t pitch;            // row pitch in floats
    // Up to two half-open local row ranges to update: [ra0, ra1) and [rb0, rb1).
    int32_t ra0, ra1, rb0, rb1;
    // 1 when the ghost row above / below holds a neighbouring slab's row (slab seam),
    // 0 when that side is a global edge (naive's clipped window applies there).
    int32_t top_present, bottom_present;
    int32_t ghost;         // ghost rows stored above / below the slab (>= fused steps on seams)
    int32_t rows_per_unit; // rows one wave marches over
    // Tapered tail of range a (temporal-blocking kernel, filled in by its launcher): the first
    // `big_chunks` chunks have rows_per_unit rows, the rest `small_rpu` rows, so that the units
    // dispatched last are short and the launch drains quickly.
    int32_t big_chunks, small_rpu;
    // second taper level: after `mid_chunks` chunks of small_rpu rows the rest have tiny_rpu rows
    // (mid_chunks < 0: one level only)
    int32_t mid_chunks, tiny_rpu;
    // how many of range a's LAST chunks are dispatched first (the chunks a bottom grid edge can touch;
    // at least 1), filled in by the launcher
    int32_t bot_first;
    // 2 = units on a global edge are dispatched as two half-height units (1 = whole): the outer strips of
    // every chunk and all strips of the first `edge_chunks` chunks in dispatch order (the bottom and top
    // chunk rows).  Edge units run the general path, ~1.6x as slow: in a launch of one or two rounds of wave
    // slots, where every unit starts at once, whole ones would finish last.  Filled in by the launcher.
    int32_t edge_split, edge_chunks;
    // Parameter-specialised variants of the temporal-blocking kernel (bit-identical results, fewer
    // instructions): bit 0 = the four side weights w[0][1], w[1][0], w[1][2], w[2][1] are exactly
    // 0.5f, bit 1 = dt is exactly 1.0f.  Both hold for Parameters::default().
    int32_t fast;
    // Columns per lane of the temporal-blocking kernel: 4 (0 means 4), 2 or 1.
    int32_t cpl;
    // 1 = this launch has the GPU to itself (a single slab, no row bands): a launch of one round may then run
    // as 16-wave workgroups that keep step (gs_launch_tb).  0 on slab chains and row bands: workgroups that
    // own whole CUs until all their waves end would keep the boundary-band kernel, the ghost-row copies and
    // RCCL's kernels out until the end of the launch (8 slabs on one GPU: 0.79 instead of 0.88 of one slab).
    int32_t allow_fair;
    // 1 = edge units of the temporal-blocking kernel take the cheap kinds of edge path where one applies
    // (gs_step_tb_k); 0 = the general path for all of them (GS_HIP_EDGE_KINDS=0, A/B timing).  Same results.
    int32_t edge_kinds;
    // In-step form: the progress (0 ... 256) from which a wave's priority is steered; before, the waves run as
    // the arbitration leaves them (filled in by the launcher).
    int32_t fair_from;
    // XCD-aware unit order (filled in by the launcher; 0 = off).  The dispatcher deals workgroups over the 8 XCDs
    // round-robin, so workgroup b runs on XCD b % 8.  From workgroup xcd_first on, every group of 8 * xcd_m
    // workgroups is renumbered so that the xcd_m workgroups an XCD gets are consecutive ones -- neighbours in the
    // grid, whose overlapping rows and columns then meet in that XCD's L2.
    int32_t xcd_m, xcd_first;
    // Boundary rule on global edges: 0 = naive's clipped window (weights anchored at the window's
    // top-left corner), 1 = full window with zeros outside the grid (gs_boundary in gs_hip.h).
    int32_t zero_halo;
    float w[3][3];         // stencil weights, row-major (parameters.rs:87-88)
    float du, dv, feed, feed_plus_kill, dt;
};

// gs_launch_window_*: one persistent launch for a whole gs_run on grids of ONE round of register-resident windows.
// Every workgroup owns a rectangle of the grid (GsWindowDesc; the rectangles tile the grid) and keeps it plus a k-cell
// apron in registers: `active` window rows (waves beyond them idle) x 128 columns.  Every k steps it stores the k-cell
// ring of its owned cells into the exchange planes, raises its flag, waits for the flags of the workgroups whose
// cells its apron covers (`nbr`) and loads its apron from their rings.  Exchange e uses the planes of parity e & 1.
// The input planes are never written: a launch that gives up (abort set) has destroyed nothing.
// Windows of one tile column share their height, and columns whose cells cost more (the grid's left and right edge
// under the clipped rule) get lower windows, so that every workgroup's step takes the same time: they all wait for
// their neighbours at every exchange, the slowest sets the pace (gs_window.cpp: plan_windows).
constexpr int kGsWindowMaxNbr = 14;
struct GsWindowDesc {
    int32_t r0, c0;    // first owned row / column (global)
    int32_t oh, ow;    // owned rows / columns (the last window of a tile column / row may reach beyond the grid)
    int32_t active;    // window rows in use: oh + 2 k rounded up to whole waves (a multiple of rpw, <= 16 rpw)
    int32_t n_nbr;     // workgroups whose owned cells lie in this window's apron
    int32_t nbr[kGsWindowMaxNbr];
};
struct GsWindowArgs {
    float *xu[2], *xv[2];      // exchange planes: local row 0, column 0; the field planes' pitch, at least the grid's rows
    int32_t *flags;            // one per workgroup; a workgroup that has finished exchange e holds epoch + e + 1
    int32_t *abort;            // sticky, 0 or the `seq` of the launch in which a poll ran out of patience (workgroups not
                               // co-resident): every workgroup of that launch and of every later one then leaves
    const GsWindowDesc *desc;  // one per workgroup (device memory)
    int32_t n_windows;
    int32_t steps;             // time steps of this launch (>= 1)
    int32_t k;                 // steps per exchange: even, 2 ... 8
    int32_t epoch;             // flags left by earlier launches are <= epoch
    int32_t patience;          // polls before a workgroup gives up
    int32_t seq;               // this launch's number on its context (>= 1): what a workgroup that gives up leaves in *abort
};

// Launchers, one set per arithmetic flavour (see gs_math in include/gs_hip.h).  Each
// returns the hipError_t of the launch.  `name` receives a static kernel-variant label.
#define GS_DECLARE_LAUNCHERS(SUFFIX)                                                           \
    hipError_t gs_launch_simple_##SUFFIX(const GsStepArgs &a, hipStream_t s, const char **name); \
    hipError_t gs_launch_stream_##SUFFIX(const GsStepArgs &a, hipStream_t s, const char **name); \
    hipError_t gs_launch_resident_##SUFFIX(const GsStepArgs &a, int steps, hipStream_t s, const char **name); \
    hipError_t gs_launch_tb_##SUFFIX(const GsStepArgs &a, int k, hipStream_t s, const char **name); \
    hipError_t gs_launch_tile_##SUFFIX(const GsStepArgs &a, int k, int shape, hipStream_t s, const char **name); \
    hipError_t gs_launch_lds_##SUFFIX(const GsStepArgs &a, hipStream_t s, const char **name);  \
    hipError_t gs_launch_window_##SUFFIX(const GsStepArgs &a, const GsWindowArgs &x, int rpw, hipStream_t s, const char **name); \
    int gs_tb_wave_slots_##SUFFIX(int k, int fast, int cpl);

GS_DECLARE_LAUNCHERS(strict)
GS_DECLARE_LAUNCHERS(fused)

// Entry points of the parameter-specialised temporal-blocking kernels (strict flavour only; their
// own translation unit, see gs_step_kernels.hip: GS_TB_OP_ONLY).  nullptr for an unknown variant.
// wg: waves per workgroup, 4 or 16 (the fair-progress form of one-round launches: K = 4, cpl 1 or 2 only).
const void *gs_tb_op_kernel_strict(int k, int fast, int cpl, int wg);

// Plane utilities (math-agnostic, defined once in gs_util_kernels.hip).
hipError_t gs_launch_colormap(const float *row0, int32_t pitch, int32_t rows, int32_t cols, float scale,
                              const uint8_t *palette, int32_t n, uint8_t *rgb, hipStream_t s);
hipError_t gs_launch_fill_rect(float *row0, int32_t pitch, int32_t r0, int32_t r1, int32_t c0,
                               int32_t c1, float value, hipStream_t s);
// rows x cols of a plane (row pitch `pitch` floats) to a dense array: gs_field_download_async's staging copy.
hipError_t gs_launch_pack_rows(const float *row0, int32_t pitch, int32_t rows, int32_t cols, float *dst, hipStream_t s);
// gs_fields_place's probe: reads `bytes` (a multiple of 16, 16-byte aligned) of x and of y and writes them back unchanged.
hipError_t gs_launch_pair_probe(void *x, void *y, size_t bytes, hipStream_t s);
