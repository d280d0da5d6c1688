// gs_step_kernels.hip -- gfx950 (CDNA4, wave64) kernels for the Gray-Scott time step.
//
// Arithmetic spec: compute_naive::Simulation::perform_step,
// /root/reference/compute/naive/src/lib.rs:42-83 -- for every cell, a row-major fold
//     acc = acc + w[i][j] * (elem - centre)
// over the 3x3 window CLIPPED to the grid, weights indexed from the window's top-left
// corner (:57-71), then the reaction update (:74-79).  Each reference operation is one
// rounded f32 operation here, in the same order; this file is compiled with
// -ffp-contract=off so the compiler never fuses.  It is compiled twice:
//   GS_MATH_FUSED=0 ("strict"): taps are sub, mul, add; f32 denormal mode = flush results,
//       keep inputs (-fdenormal-fp-math-f32=preserve-sign,ieee -> FP_DENORM 1), which is
//       MXCSR.FTZ without DAZ, i.e. the reference's DenormalsFlusher
//       (compute/shared/src/lib.rs:161-180).
//   GS_MATH_FUSED=1 ("fused"):  taps are sub, fma; denormals kept.  Exact for weights that
//       are 0 or a power of two whenever the product is a normal number.
//
// Facts the kernels rely on (derived from the spec, checked by tests/test_oracle_kat.py):
//   * the centre tap contributes w * (u - u) = +0 and acc is never -0, so it is skipped;
//   * a clipped (absent) neighbour equals "neighbour value := centre value" (adds +0);
//   * with the row above absent (global top edge) the centre row takes weight row 0 and the
//     row below weight row 1; same shift for the columns on the global left edge;
//   * "0.0f + first tap" is kept: it turns a -0 product into +0 exactly as the fold does.
//
// Kernels
//   gs_step_simple_k   one thread per cell, literal window loop (cross-check kernel).
//   gs_step_stream_k   one step per launch: a wave owns a 256-column strip (one 16-B load
//       per lane per row and species), marches down `rows_per_unit` rows keeping a 3-row
//       window in registers, and gets its left/right neighbours from the adjacent lanes
//       with DPP wave shifts (no LDS, no extra memory traffic); only lanes 0 and 63 fetch
//       one halo column each.  Each input element is read from HBM once per step except
//       the 2 rows shared by vertically adjacent units.  HBM-bound: 16 B per cell-step.
//   gs_step_tb_k<K,FAST> the production kernel of gs_run: K <= 4 time steps per launch
//       (temporal blocking), K register-resident time levels per wave, sacrificial edge
//       lanes instead of halo loads.  ~16 B of HBM traffic per cell for K steps;
//       VALU-issue bound for K >= 3.  Bit-identical to K single steps.
//
// This file sets the flavour macros, includes the kernels -- gs_cell.h (per-cell arithmetic), gs_march.h (gs_step_tb_k and
// its variant with full difference sharing), gs_single_step.h (simple / stream / LDS-staged), gs_lds_resident.h (resident
// and LDS-window kernels), gs_window_kernel.h (the persistent window kernel) -- and holds their launchers.
// Build-time switches of A/B and diagnostic builds (none is set in the shipped build) and the run-time
// GS_HIP_* switches of the launchers live in gs_experiments.h.
#include "gs_kernels.h"
#include "gs_experiments.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>

#ifndef GS_MATH_FUSED
#error "compile with -DGS_MATH_FUSED=0 or 1"
#endif
// GS_TB_OP_ONLY=1: this translation unit provides nothing but the parameter-specialised (".op")
// instances of gs_step_tb_k, through gs_tb_op_kernel_strict() (24 kernels: built apart from the
// rest so that the translation units compile in parallel; grayscott_amd/_build.py).
#ifndef GS_TB_OP_ONLY
#define GS_TB_OP_ONLY 0
#endif
#if GS_TB_OP_ONLY && GS_MATH_FUSED
#error "the fused build has no specialised variants"
#endif

#if GS_MATH_FUSED
#define GS_SUFFIX(x) x##_fused
#define GS_TAP(acc, w, s, c) (acc) = __builtin_fmaf((w), (s) - (c), (acc))
#define GS_MATH_NAME "fused"
#else
#define GS_SUFFIX(x) x##_strict
#define GS_TAP(acc, w, s, c) (acc) = (acc) + (w) * ((s) - (c))
#define GS_MATH_NAME "strict"
#endif

#include "gs_cell.h"
#include "gs_march.h"
#if !GS_TB_OP_ONLY
#include "gs_single_step.h"
#include "gs_lds_resident.h"
#include "gs_window_kernel.h"
#endif

#if !GS_TB_OP_ONLY
// Opt-in for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute
// belongs to the device function ON THE CURRENT DEVICE, so what has been set is remembered per (device,
// function): a process that drives several GPUs (device_ids = 0, 1, ...; two contexts) opts in on each.
// Contexts on different threads launch through here: the table has a lock, like tb_waves_of's.
static bool dyn_lds_seen(int device, const void *fn, int bytes, bool record)
{
    struct Entry { int device; const void *fn; int bytes; };
    static Entry table[256];
    static int n = 0;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    for (int i = 0; i < n; ++i)
        if (table[i].device == device && table[i].fn == fn) {
            if (table[i].bytes >= bytes) return true;
            if (record) table[i].bytes = bytes;
            return false;
        }
    if (record && n < 256) table[n++] = Entry{device, fn, bytes};
    return false;
}
static hipError_t ensure_dyn_lds(const void *fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (dyn_lds_seen(device, fn, (int)bytes, false)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) (void)dyn_lds_seen(device, fn, (int)bytes, true);
    return e;
}
// What the table above keys on, for the unit test of the key (tests/test_capi_cpu.py): 1 when (device, fn
// slot, bytes) is new, and it is recorded; 0 when a launch on that device would skip the call.
#if !GS_MATH_FUSED
extern "C" int32_t gs_debug_dyn_lds_key(int32_t device, int32_t slot, int32_t bytes)
{
    static const char slots[16] = {0};
    if (slot < 0 || slot >= 16) return -1;
    if (dyn_lds_seen(-1000 - device, &slots[slot], bytes, false)) return 0;
    (void)dyn_lds_seen(-1000 - device, &slots[slot], bytes, true);
    return 1;
}
#endif

hipError_t GS_SUFFIX(gs_launch_simple)(const GsStepArgs &a, hipStream_t s, const char **name)
{
    if (name) *name = "simple/" GS_MATH_NAME;
    const long nrows = (long)(a.ra1 - a.ra0) + (a.rb1 - a.rb0);
    if (nrows <= 0 || a.cols <= 0) return hipSuccess;
    const long bpr = (a.cols + 255) >> 8;
    const long blocks = nrows * bpr;
    if (blocks > 0x7fffffffL) return hipErrorInvalidConfiguration;
    GsStepArgs args = a;
    void *kargs[] = {&args};
    return hipLaunchKernel(reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_simple_k)),
                           dim3((unsigned)blocks), dim3(256), kargs, 0, s);
}

// `steps` time steps of a grid of at most kResidentCells cells in one launch (gs_run_resident_k);
// the result is stored in the out-planes when steps is odd, else back in the in-planes.
hipError_t GS_SUFFIX(gs_launch_resident)(const GsStepArgs &a, int steps, hipStream_t s, const char **name)
{
    static const char *const names[2] = {"resident-lds/" GS_MATH_NAME, "resident-lds/" GS_MATH_NAME ".op"};
    const long cells = (long)a.rows * a.cols;
    if (a.rows <= 0 || a.cols <= 0 || cells > kResidentCells || steps < 0 || a.top_present || a.bottom_present)
        return hipErrorInvalidValue;
    int fast = a.fast & (GS_MATH_FUSED ? 0 : 3);
    if (fast != 3) fast = 0; // only the variant for the default parameters is built besides the general one
    if (name) *name = names[fast ? 1 : 0];
    const void *fn = nullptr;
    const int zh = a.zero_halo ? 1 : 0;
#define GS_RES_FN(F, Z) reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_resident_k)<F, Z>)
    if (fast) fn = zh ? GS_RES_FN(GS_MATH_FUSED ? 0 : 3, 1) : GS_RES_FN(GS_MATH_FUSED ? 0 : 3, 0);
    else fn = zh ? GS_RES_FN(0, 1) : GS_RES_FN(0, 0);
#undef GS_RES_FN
    const size_t lds = (size_t)4 * (a.rows + 2) * (a.cols + 2) * sizeof(float); // <= 74 KB (1 x 1536 cells)
    { // more than 64 KB of dynamic LDS needs the opt-in, per device and device function
        const hipError_t e = ensure_dyn_lds(fn, lds > 64 * 1024 ? (size_t)80 * 1024 : lds);
        if (e != hipSuccess) return e;
    }
    GsStepArgs args = a;
    int to_out = steps & 1;
    void *kargs[] = {&args, &steps, &to_out};
    // only the waves that hold cells take part (and in the barrier of every step)
    const long threads = cells >= kResidentThreads ? kResidentThreads : ((cells + 63) / 64) * 64;
    return hipLaunchKernel(fn, dim3(1), dim3((unsigned)threads), kargs, lds, s);
}

// K <= kGsTileMaxSteps time steps of a single slab in one launch of gs_run_tile_k (in-planes -> out-planes).
// `shape`: 0 = windows of 32 rows x 64 columns (2 rows per wave), 1 = 16 x 64 (1 row), 2 = 64 x 64 (4 rows);
// 2K < window rows.
hipError_t GS_SUFFIX(gs_launch_tile)(const GsStepArgs &a, int k, int shape, hipStream_t s, const char **name)
{
    static const char *const names[3][2] = {{"tile32x64/" GS_MATH_NAME, "tile32x64/" GS_MATH_NAME ".op"},
                                            {"tile16x64/" GS_MATH_NAME, "tile16x64/" GS_MATH_NAME ".op"},
                                            {"tile64x64/" GS_MATH_NAME, "tile64x64/" GS_MATH_NAME ".op"}};
    static const int rpw[3] = {2, 1, 4};
    if (a.rows <= 0 || a.cols <= 0 || k < 1 || k > kTileMaxK || shape < 0 || shape > 2 || a.top_present || a.bottom_present ||
        2 * k >= tile_rows(rpw[shape]))
        return hipErrorInvalidValue;
    int fast = a.fast & (GS_MATH_FUSED ? 0 : 3);
    if (fast != 3) fast = 0; // only the variant for the default parameters is built besides the general one
    if (name) *name = names[shape][fast ? 1 : 0];
    const long ho = tile_rows(rpw[shape]) - 2 * k, wo = kTileCols - 2 * k;
    const long tiles = ((a.rows + ho - 1) / ho) * ((a.cols + wo - 1) / wo);
    if (tiles > 0x7fffffffL) return hipErrorInvalidConfiguration;
    const void *fn = nullptr;
#define GS_TILE_FN(S, RPW_)                                                                                   \
    case S: fn = fast ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_tile_k)<RPW_, GS_MATH_FUSED ? 0 : 3>)  \
                      : reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_tile_k)<RPW_, 0>); break;
    switch (shape) { GS_TILE_FN(0, 2) GS_TILE_FN(1, 1) GS_TILE_FN(2, 4) }
#undef GS_TILE_FN
    size_t lds = tile_lds_bytes(rpw[shape]);
    static const int lds_floor = gs_env_int("GS_HIP_TILE_LDS_FLOOR", 0, 0, 160 * 1024);
    if (lds < (size_t)lds_floor) lds = (size_t)lds_floor; // experiment: limit the workgroups per CU
    { // more than 64 KB of dynamic LDS needs the opt-in, per device and device function
        const hipError_t e = ensure_dyn_lds(fn, lds);
        if (e != hipSuccess) return e;
    }
    GsStepArgs args = a;
    void *kargs[] = {&args, &k};
    return hipLaunchKernel(fn, dim3((unsigned)tiles), dim3(kTileWaves * 64), kargs, lds, s);
}

// One persistent launch of gs_run_window_k: `x.steps` time steps of a single slab, in-planes -> out-planes.
// rpw: rows per wave (a window is at most 16 rpw rows x 128 columns); x.desc holds the caller's tiling of the grid.
hipError_t GS_SUFFIX(gs_launch_window)(const GsStepArgs &a, const GsWindowArgs &x, int rpw, hipStream_t s, const char **name)
{
    // (5 rows per wave: 80-row windows.  The 96-row form of round 4 -- 6 rows per wave, 12 cells per lane -- spilled 43
    // registers in the strict build and tied with the marching kernel where it applied; it is gone.)
    static const char *const names[3] = {"window-r5/" GS_MATH_NAME, "window-r5/" GS_MATH_NAME ".op", "window-r5/" GS_MATH_NAME ".op.ds"};
    if (a.rows <= 0 || a.cols <= 0 || a.top_present || a.bottom_present || rpw != 5 || x.steps < 1 || x.k < 2 ||
        x.k > 8 || (x.k & 1) || 2 * x.k >= win_rows(rpw) || 2 * x.k + 2 > kWinCols || !x.flags || !x.abort || !x.xu[0] || !x.xu[1] || !x.xv[0] || !x.xv[1])
        return hipErrorInvalidValue;
    if (!x.desc || x.n_windows < 1 || x.seq < 1) return hipErrorInvalidValue;
    // byte offsets inside a plane are 32-bit in the kernel
    if ((long)(a.rows + 8) * a.pitch * 4 > 0x7fffffffL) return hipErrorInvalidValue;
    int fast = a.fast & (GS_MATH_FUSED ? 0 : 7);
    // only the variants for the default parameters are built besides the general one: .op, and .op.ds with full
    // difference sharing inside a wave's band (gs_options.share_taps = 2 switches it off)
    if ((fast & 3) != 3) fast = 0;
    if (name) *name = names[fast == 7 ? 2 : (fast ? 1 : 0)];
    const void *fn = nullptr;
#define GS_WIN_FN(R) (fast == 7 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_window_k)<R, GS_MATH_FUSED ? 0 : 7>) \
                      : fast    ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_window_k)<R, GS_MATH_FUSED ? 0 : 3>) \
                                : reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_window_k)<R, 0>))
    fn = GS_WIN_FN(5);
#undef GS_WIN_FN
    const size_t lds = win_lds_bytes();
    { // more than 64 KB of dynamic LDS needs the opt-in, per device and device function
        const hipError_t e = ensure_dyn_lds(fn, lds);
        if (e != hipSuccess) return e;
    }
    GsStepArgs args = a;
    GsWindowArgs xa = x;
    void *kargs[] = {&args, &xa};
    return hipLaunchKernel(fn, dim3((unsigned)x.n_windows), dim3(kWinWaves * 64), kargs, lds, s);
}

hipError_t GS_SUFFIX(gs_launch_stream)(const GsStepArgs &a, hipStream_t s, const char **name)
{
    if (name) *name = "stream-g2/" GS_MATH_NAME;
    if (a.cols <= 0 || a.rows_per_unit <= 0) return hipErrorInvalidValue;
    const long rpu = a.rows_per_unit;
    const long chunks = ((long)(a.ra1 - a.ra0) + rpu - 1) / rpu + ((long)(a.rb1 - a.rb0) + rpu - 1) / rpu;
    if (chunks <= 0) return hipSuccess;
    const long strips = (a.cols + 255) >> 8;
    const long blocks = (chunks * strips + 3) / 4;
    if (blocks > 0x7fffffffL) return hipErrorInvalidConfiguration;
    GsStepArgs args = a;
    // XCD-aware order, as in gs_launch_tb: this kernel IS bound by HBM, so the re-reads the XCDs' L2s absorb are
    // time -- 16384^2 365 k -> 376 k (6.0 TB/s algorithmic), 4096^2 323 k -> 347 k, 1080 x 1920 171 k -> 186 k; 8192^2
    // unchanged on average (265-333 k from one context to the next either way: the four planes' placement decides).
    // Groups of 8 x 64 workgroups lose 6 % (profiles/archive/r03_sweeps.md, section 12).  GS_HIP_XCD_M_STREAM = 0 / n: off / 8 n.
    static const int xcd_env = gs_env_int("GS_HIP_XCD_M_STREAM", -1, 0, kGsXcdGroupMax);
    args.xcd_m = xcd_env >= 0 ? xcd_env : 16;
    void *kargs[] = {&args};
    return hipLaunchKernel(reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_stream_k)<2>),
                           dim3((unsigned)blocks), dim3(256), kargs, 0, s);
}

// K fused steps over the row ranges of GsStepArgs; on slab seams the ghost rows must be K deep.
// Kernel entry for k fused steps, specialisation `fast` (already reduced to {0, 1, 3}) and cpl columns per lane.
static const void *tb_entry(int k, int fast, int cpl, int wg = 4)
{
    const void *fn = nullptr;
#define GS_TB_CASE(KK, CC)                                                                      \
    case (KK) * 8 + (CC): fn = reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<KK, 0, CC>); break;
    if (wg == 16) { // the fair-progress form: 4 fused steps, 1 or 2 columns per lane
        if (k != 4 || (cpl != 1 && cpl != 2)) return nullptr;
        if (fast) {
#if !GS_MATH_FUSED
            return gs_tb_op_kernel_strict(k, fast, cpl, 16);
#endif
        }
#if GS_MATH_FUSED
        // (2 columns per lane need 129 registers in the fused flavour, one more than a wave of a 16-wave workgroup may have:
        // the variant spilled a register to scratch; one-round launches of the fused flavour run as 4-wave workgroups)
        if (cpl == 2) return nullptr;
        return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 0, 1, 16>);
#else
        return cpl == 1 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 0, 1, 16>)
                        : reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 0, 2, 16>);
#endif
    }
    if (fast) {
#if !GS_MATH_FUSED
        fn = gs_tb_op_kernel_strict(k, fast, cpl, 4);
#endif
    } else {
        switch (k * 8 + cpl) {
            GS_TB_CASE(1, 4) GS_TB_CASE(2, 4) GS_TB_CASE(3, 4) GS_TB_CASE(4, 4)
            GS_TB_CASE(1, 2) GS_TB_CASE(2, 2) GS_TB_CASE(3, 2) GS_TB_CASE(4, 2)
            GS_TB_CASE(1, 1) GS_TB_CASE(2, 1) GS_TB_CASE(3, 1) GS_TB_CASE(4, 1)
        }
    }
#undef GS_TB_CASE
    return fn;
}

// Waves per SIMD the register file allows a kernel entry: 512 registers per lane, allocated in steps
// of 8 (MI355X_MICROARCH.md, register files); the kernels use no LDS memory.
static int tb_waves_of(const void *f)
{
    static const void *occ_fn[64];
    static int occ_waves[64], occ_n = 0;
    static std::mutex occ_lock; // contexts on different threads launch through here
    std::lock_guard<std::mutex> occ_guard(occ_lock);
    for (int i = 0; i < occ_n; ++i)
        if (occ_fn[i] == f) return occ_waves[i];
    hipFuncAttributes attr;
    attr.numRegs = 0;
    int w = 2;
    if (hipFuncGetAttributes(&attr, f) == hipSuccess && attr.numRegs > 0) {
        const int alloc = ((attr.numRegs + 7) / 8) * 8;
        w = 512 / alloc > 8 ? 8 : (512 / alloc < 1 ? 1 : 512 / alloc);
    } else {
        (void)hipGetLastError();
    }
    if (gs_env_int("GS_HIP_TRACE_TUNER", 0, 0, 1))
        std::fprintf(stderr, "gs_hip: kernel entry %p: %d registers -> %d waves per SIMD\n", f, attr.numRegs, w);
    if (occ_n < 64) { occ_fn[occ_n] = f; occ_waves[occ_n++] = w; }
    return w;
}

// The variant that runs for GsStepArgs::fast = `fast` with k fused steps, cpl columns per lane and wg waves per
// workgroup: 0 (general), 1 (side weights 0.5), 3 (and dt == 1), 7 (and full difference sharing: built for 2 columns
// per lane, 2 to 4 fused steps) or 15 (and across lanes).
static int tb_reduce_fast(int fast, int k = 0, int cpl = 0, int wg = 4)
{
    // The fused build has no use for bit 0 (its taps are sub + fma already) and measured slower
    // with bit 1 (profiles/archive/r01_sweeps.md, runs 48/49): it always runs the general variant.  dt == 1
    // alone (fast == 2) is not worth a variant either, and bit 2 means nothing without the other two.
    fast &= GS_MATH_FUSED ? 0 : 15;
    if (!(fast & 1)) return 0;
    if ((fast & 7) == 7) {
#if !GS_MATH_FUSED
        if ((fast & 8) && gs_tb_op_kernel_strict(k, 15, cpl, wg)) return 15;
        if (gs_tb_op_kernel_strict(k, 7, cpl, wg)) return 7;
#endif
        return 3;
    }
    return fast & 3;
}

// Wave slots of the chip for the kernel entry a launch with these parameters would use (the tuner's
// "a launch of exactly r rounds" candidates, gs_tuner.cpp); 0 = no such entry.
int GS_SUFFIX(gs_tb_wave_slots)(int k, int fast, int cpl)
{
    if (k < 1 || k > 4 || (cpl != 1 && cpl != 2 && cpl != 4)) return 0;
    const void *fn = tb_entry(k, tb_reduce_fast(fast, k, cpl), cpl);
    return fn ? 1024 * tb_waves_of(fn) : 0;
}

hipError_t GS_SUFFIX(gs_launch_tb)(const GsStepArgs &a, int k, hipStream_t s, const char **name)
{
    // "cN": N columns per lane (4 = the wide layout); ".op": the variant specialised for the
    // default (Oono-Puri) side weights, with or without dt == 1
    // ".op.ds": ... and with full difference sharing (cells_vshare)
#define GS_TB_NAMES(C)                                                                          \
    {{"tb-k1" C "/" GS_MATH_NAME, "tb-k2" C "/" GS_MATH_NAME, "tb-k3" C "/" GS_MATH_NAME, "tb-k4" C "/" GS_MATH_NAME}, \
     {"tb-k1" C "/" GS_MATH_NAME ".op", "tb-k2" C "/" GS_MATH_NAME ".op", "tb-k3" C "/" GS_MATH_NAME ".op",            \
      "tb-k4" C "/" GS_MATH_NAME ".op"},                                                        \
     {"tb-k1" C "/" GS_MATH_NAME ".op.ds", "tb-k2" C "/" GS_MATH_NAME ".op.ds", "tb-k3" C "/" GS_MATH_NAME ".op.ds",   \
      "tb-k4" C "/" GS_MATH_NAME ".op.ds"},                                                     \
     {"tb-k1" C "/" GS_MATH_NAME ".op.dx", "tb-k2" C "/" GS_MATH_NAME ".op.dx", "tb-k3" C "/" GS_MATH_NAME ".op.dx",   \
      "tb-k4" C "/" GS_MATH_NAME ".op.dx"}}
    // ".op.dx": ... and across lanes (cells_xshare)
    static const char *const names[3][4][4] = {GS_TB_NAMES("c1"), GS_TB_NAMES("c2"), GS_TB_NAMES("")};
#undef GS_TB_NAMES
    // "f": the fair-progress form (16-wave workgroups) of one-round launches
    static const char *const names16[2][4] = {
        {"tb-k4c1f/" GS_MATH_NAME, "tb-k4c1f/" GS_MATH_NAME ".op", "tb-k4c1f/" GS_MATH_NAME ".op.ds", "tb-k4c1f/" GS_MATH_NAME ".op.dx"},
        {"tb-k4c2f/" GS_MATH_NAME, "tb-k4c2f/" GS_MATH_NAME ".op", "tb-k4c2f/" GS_MATH_NAME ".op.ds", "tb-k4c2f/" GS_MATH_NAME ".op.dx"}};
    auto name_of = [](int f) { return f == 15 ? 3 : (f == 7 ? 2 : (f ? 1 : 0)); };
    if (k < 1 || k > 4 || a.cols <= 0 || a.rows_per_unit <= 0) return hipErrorInvalidValue;
    const int cpl = a.cpl == 0 ? 4 : a.cpl;
    if (cpl != 1 && cpl != 2 && cpl != 4) return hipErrorInvalidValue;
    const int fast = tb_reduce_fast(a.fast, k, cpl);
    if (name) *name = names[cpl == 1 ? 0 : (cpl == 2 ? 1 : 2)][name_of(fast)][k - 1];
    const long rpu = a.rows_per_unit;
    const long rows_a = (long)a.ra1 - a.ra0;
    const long W = tb_cols_per_wave(k, cpl);
    const long strips = (a.cols + W - 1) / W;
    // Kernel entry first: the taper below needs its occupancy.
    const void *fn = tb_entry(k, fast, cpl);
    if (!fn) return hipErrorInvalidValue;
    const int waves = tb_waves_of(fn);
    // Tapered tail (consecutive passes are dependent launches that cannot overlap, so the drain phase
    // of a launch is idle time): when the launch is at least two rounds of the chip's wave slots, the
    // last round of units is an eighth as tall as the others and the round before it half as tall.
    // Measured at 16384^2 (profiles/archive/r02_sweeps.md, section 7): +1...2 % over round 1's single level
    // (the last two rounds at a quarter), and unit heights of 128-192 rows become usable.
    const long slots = 1024L * waves; // 256 CUs x 4 SIMDs x waves per SIMD
    long big_chunks = rows_a > 0 ? rows_a / rpu : 0, small = rpu, mid_chunks = -1, tiny = rpu;
    if ((rows_a / rpu) * strips >= 2 * slots) {
        const long h1 = rpu / 2 >= 2L * k ? rpu / 2 : 2L * k;
        const long h2 = rpu / 8 >= 2L * k ? rpu / 8 : 2L * k;
        const long c1 = (slots + strips - 1) / strips, c2 = (slots + strips - 1) / strips;
        const long rows12 = c1 * h1 + c2 * h2;
        if (h1 < rpu && rows12 <= rows_a / 3) {
            big_chunks = (rows_a - rows12) / rpu;
            small = h1;
            if (h2 < h1) {
                tiny = h2;
                mid_chunks = (rows_a - big_chunks * rpu - c2 * h2 + h1 - 1) / h1;
                while (mid_chunks > 0 && rows_a - big_chunks * rpu - mid_chunks * h1 < 0) --mid_chunks;
                if (mid_chunks < 0) mid_chunks = 0;
            }
        }
    }
    const long rest = rows_a - big_chunks * rpu;
    const long chunks_a = rows_a <= 0 ? 0
                        : mid_chunks < 0 ? big_chunks + (rest + small - 1) / small
                                         : big_chunks + mid_chunks + (rest - mid_chunks * small + tiny - 1) / tiny;
    const long chunks = chunks_a + ((long)(a.rb1 - a.rb0) + rpu - 1) / rpu;
    if (chunks <= 0) return hipSuccess;
    if (k > a.ghost && (a.top_present || a.bottom_present)) return hipErrorInvalidValue;
    GsStepArgs args = a;
    args.big_chunks = (int32_t)big_chunks;
    args.small_rpu = (int32_t)small;
    args.mid_chunks = (int32_t)mid_chunks;
    args.tiny_rpu = (int32_t)tiny;
    // Row range of chunk cc of range a (the kernel's formulas); the last `bot` chunks -- the ones the
    // grid's bottom edge can touch, at least one -- are dispatched first, then the chunks from the top.
    auto chunk_rows = [&](long cc, long &r0, long &r1) {
        if (cc < big_chunks) { r0 = a.ra0 + cc * rpu; r1 = r0 + rpu; }
        else if (mid_chunks < 0 || cc < big_chunks + mid_chunks) { r0 = a.ra0 + big_chunks * rpu + (cc - big_chunks) * small; r1 = r0 + small < a.ra1 ? r0 + small : a.ra1; }
        else { r0 = a.ra0 + big_chunks * rpu + mid_chunks * small + (cc - big_chunks - mid_chunks) * tiny; r1 = r0 + tiny < a.ra1 ? r0 + tiny : a.ra1; }
    };
    long bot = 0, r0 = 0, r1 = 0;
    if (!a.bottom_present)
        for (; bot < chunks_a; ++bot) { chunk_rows(chunks_a - 1 - bot, r0, r1); if (!(r1 + k > a.rows)) break; }
    if (bot < 1 && chunks_a > 0) bot = 1; // the launch order of earlier rounds: the last chunk first
    args.bot_first = (int32_t)bot;
    // Edge units as two halves each when the launch is about one round of wave slots (every unit starts at
    // once, so the slow edge units would finish last: 1080 x 1920 +5.7 %, 2048 x 4096 +1.6 %; from two rounds
    // up the edge-first order does the job and halves only add recomputed rows: 8192^2 -1 %;
    // profiles/archive/r02_sweeps.md, section 11).  The kernel's dispatch order: the outer strips of every chunk, then all strips of the
    // bottom `bot` and the top chunk row of range a, then the rest.
    static const int split_env = gs_env_int("GS_HIP_EDGE_SPLIT", -1, 0, 1);
    const long er = ((strips - 1) * W + tb_sacrificial_lanes(k, cpl) * cpl >= a.cols && strips >= 2) ? 2 : 1, ne = 1 + er;
    bool split = 4 * chunks * strips <= 5 * slots && rpu >= 2;
    if (split_env >= 0) split = split_env != 0;
    long units = chunks * strips;
    args.edge_split = 1;
    args.edge_chunks = 0;
    if (split) {
        args.edge_split = 2;
        if (strips <= ne) {
            units = chunks * strips * 2;
        } else {
            const long nec = chunks_a > 0 ? (bot + 1 < chunks ? bot + 1 : chunks) : 0;
            args.edge_chunks = (int32_t)nec;
            units = chunks * ne * 2 + nec * (strips - ne) * 2 + (chunks - nec) * (strips - ne);
        }
    }
    // A launch that fits the chip in ONE round of 16-wave workgroups (one per CU, 4 waves per SIMD) runs the
    // fair-progress form of the kernel (tb_march<FAIR>): every unit starts at once there and, left to the
    // SIMDs' oldest-first arbitration, the waves of a SIMD finish one after the other, the last one alone.
    // Not for short marches of the 1-column layout: there most of a unit's ticks are the memory-bound filling
    // of the level pipeline, and waves left out of phase by the oldest-first arbitration hide each other's
    // waits.  Free-running / in step, same box (profiles/archive/r03_sweeps.md, section 2): 1 column per lane, 10-row
    // units 430 k / 390 k, 12 rows 465 k / 443 k, 16 rows 524 k / 514 k, 20 rows 565 k / 573 k, 40 rows 677 k / 705 k;
    // 2 columns per lane, 10 rows 523 k / 537 k, 15 rows 615 k / 633 k, 19 rows 687 k / 738 k, 38 rows 782 k / 865 k.
    // GS_HIP_FAIR = 0 / 1 forces it off / on.
    static const int fair_env = gs_env_int("GS_HIP_FAIR", -1, 0, 1);
    const bool fair = a.allow_fair && units <= 4096 && units > 1024 && (fair_env < 0 ? (cpl == 2 || rpu >= 20) : fair_env != 0);
    const int fast16 = fair ? tb_reduce_fast(a.fast, k, cpl, 16) : 0;
    const void *fair_fn = fair ? tb_entry(k, fast16, cpl, 16) : nullptr;
    static const int fair_from_env = gs_env_int("GS_HIP_FAIR_FROM", -1, 0, 256);
    args.fair_from = fair_from_env >= 0 ? fair_from_env : 0;
    void *kargs[] = {&args};
    if (fair_fn) {
        if (name) *name = names16[cpl == 1 ? 0 : 1][name_of(fast16)];
        return hipLaunchKernel(fair_fn, dim3((unsigned)((units + 15) / 16)), dim3(1024), kargs, 0, s);
    }
    const long blocks = (units + 3) / 4;
    if (blocks > 0x7fffffffL) return hipErrorInvalidConfiguration;
    // XCD-aware unit order (GsStepArgs::xcd_m): the dispatcher deals workgroups over the 8 XCDs round-robin, so
    // four-strip neighbours in the grid land on eight different L2s and each fetches the columns and rows their
    // windows share for itself.  With every XCD taking 16 consecutive workgroups of each group of 128, the HBM
    // reads of a 16384^2 launch fall from 2.376 to 2.239 GiB (minimum 2.0; FETCH_SIZE, tools/archive/fetch_ab.sh) and the
    // launch gains 0.3-0.5 % (8192^2 +0.8 %, 4 / 8 slabs on one GPU +1.4 / +0.6 %).  Larger groups read no less
    // (68: 2.226 GiB) and run slower (-2 %, 136: -6 %: an XCD's share of the last groups is all tall or all short
    // units).  Launches of about one round keep the plain order: 1080 x 1920 loses 1.2 % with the renumbering
    // (profiles/archive/r03_sweeps.md, section 12).  GS_HIP_XCD_M = 0 / n forces it off / to groups of 8 n.
    static const int xcd_env = gs_env_int("GS_HIP_XCD_M", -1, 0, kGsXcdGroupMax);
    args.xcd_m = xcd_env >= 0 ? xcd_env : (units >= 2 * slots ? 16 : 0);
    // the edge units at the head of the dispatch order stay dealt over all XCDs (they are the slow ones)
    args.xcd_first = (int32_t)(((chunks * ne * (args.edge_split == 2 ? 2 : 1) + 3) / 4 + 7) / 8 * 8);
    return hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(256), kargs, 0, s);
}

hipError_t GS_SUFFIX(gs_launch_lds)(const GsStepArgs &a, hipStream_t s, const char **name)
{
    if (name) *name = "lds-tile16/" GS_MATH_NAME;
    if (a.cols <= 0) return hipErrorInvalidValue;
    const long chunks = ((long)(a.ra1 - a.ra0) + kLdsTileRows - 1) / kLdsTileRows +
                        ((long)(a.rb1 - a.rb0) + kLdsTileRows - 1) / kLdsTileRows;
    if (chunks <= 0) return hipSuccess;
    const long strips = (a.cols + 255) >> 8;
    const long blocks = chunks * strips;
    if (blocks > 0x7fffffffL) return hipErrorInvalidConfiguration;
    GsStepArgs args = a;
    void *kargs[] = {&args};
    return hipLaunchKernel(reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_lds_k)), dim3((unsigned)blocks),
                           dim3(256), kargs, 0, s);
}
#endif // !GS_TB_OP_ONLY

#if defined(GS_WIN_TRACE) && !GS_TB_OP_ONLY
extern "C" int32_t GS_SUFFIX(gs_debug_win_trace_read)(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_win_trace), sizeof(unsigned long long) * 1024 * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif

#if defined(GS_TB_TRACE)
// Copies the trace buffer of THIS translation unit's kernels out (diagnostic builds only).
#if GS_TB_OP_ONLY
extern "C" int32_t gs_debug_trace_read_op(unsigned long long *dst, int32_t units, int32_t clear)
#else
extern "C" int32_t GS_SUFFIX(gs_debug_trace_read)(unsigned long long *dst, int32_t units, int32_t clear)
#endif
{
    if (units > kTraceUnits) units = kTraceUnits;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_trace_buf), (size_t)units * kTraceWords * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(gs_trace_buf)) != hipSuccess) return -1;
        if (hipMemset(p, 0, (size_t)units * kTraceWords * sizeof(unsigned long long)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#if GS_TB_OP_ONLY
// Kernel entry of the specialised variant for K fused steps, `fast` in {1, 3} (GsStepArgs::fast)
// and `cpl` columns per lane.
const void *gs_tb_op_kernel_strict(int k, int fast, int cpl, int wg)
{
    if (fast == 7) { // full difference sharing: 2 columns per lane
        if (cpl != 2) return nullptr;
        if (wg == 16) return k == 4 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_ds_k)<4, 16>) : nullptr;
        switch (k) {
        case 2: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_ds_k)<2>);
        case 3: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_ds_k)<3>);
        case 4: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_ds_k)<4>);
        default: return nullptr;
        }
    }
    if (fast == 15) { // ... and across lanes
        if (cpl != 2) return nullptr;
        if (wg == 16) return k == 4 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_dx_k)<4, 16>) : nullptr;
        switch (k) {
        case 2: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_dx_k)<2>);
        case 3: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_dx_k)<3>);
        case 4: return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_dx_k)<4>);
        default: return nullptr;
        }
    }
    if (wg == 16) {
        if (k != 4 || (cpl != 1 && cpl != 2) || (fast != 1 && fast != 3)) return nullptr;
        if (fast == 1)
            return cpl == 1 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 1, 1, 16>)
                            : reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 1, 2, 16>);
        return cpl == 1 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 3, 1, 16>)
                        : reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 3, 2, 16>);
    }
#define GS_TB_CASE(KK, FF, CC)                                                                  \
    case ((KK) * 4 + (FF)) * 8 + (CC): return reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<KK, FF, CC>);
#define GS_TB_CASES(FF, CC) GS_TB_CASE(1, FF, CC) GS_TB_CASE(2, FF, CC) GS_TB_CASE(3, FF, CC) GS_TB_CASE(4, FF, CC)
    switch ((k * 4 + fast) * 8 + cpl) {
        GS_TB_CASES(1, 4) GS_TB_CASES(3, 4)
        GS_TB_CASES(1, 2) GS_TB_CASES(3, 2)
        GS_TB_CASES(1, 1) GS_TB_CASES(3, 1)
    default: return nullptr;
    }
#undef GS_TB_CASES
#undef GS_TB_CASE
}
#endif
