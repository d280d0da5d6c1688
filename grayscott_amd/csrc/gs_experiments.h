// gs_experiments.h -- every switch of libgs_hip.so that is NOT part of the product's interface, in one place.
//
// (1) Build-time switches of A/B and diagnostic builds (tools/ab_build.py NAME -DGS_...=v builds
//     grayscott_amd/variants/libgs_hip_NAME.so; GS_HIP_LIBRARY makes capi.py load it).  The shipped build sets
//     none of them; the values below are what ships.
// (2) Run-time GS_HIP_* environment switches read by the launchers (A/B timing of launch policies, traces).
//     They never change results -- every combination is bit-identical and covered by the GPU parity suite --
//     and are documented for users in include/gs_hip.h ("Environment").
#pragma once
#include <cstdlib>

// ---- (1) build-time --------------------------------------------------------------------------------------
// GS_TB_XLANE      1 = the temporally blocked kernel fetches its neighbour-lane columns through the LDS crossbar
//                  (ds_bpermute_b32: no VALU issue slot); 0 = DPP wave shifts (the round-1 form, -7 % at 16384^2).
#ifndef GS_TB_XLANE
#define GS_TB_XLANE 1
#endif
// GS_TB_HSHARE     1 = a lane's two (four) cells share the side difference between them (cells_interior).
#ifndef GS_TB_HSHARE
#define GS_TB_HSHARE 1
#endif
// GS_TB_LATE_FETCH 1 = the K = 4 / 2-columns-per-lane march requests its next level-0 row at the END of a tick
//                  (2 rows in flight while the levels are computed, not 3): what keeps the entry at 126 registers.
//                  Other layouts keep the early request: worth 2-6 % where the occupancy does not change.
#ifndef GS_TB_LATE_FETCH
#define GS_TB_LATE_FETCH 1
#endif
// GS_TB_AUX_LOAD / GS_TB_AUX_STORE  cache-policy bits of every plane access of the marching kernel (gfx950:
//                  1 = sc0, 2 = nt, 16 = sc1).  16 / 16 was the timing experiment "what would accesses that other
//                  CUs can observe inside a launch cost" (profiles/archive/r03_sweeps.md, section 3).
#ifndef GS_TB_AUX_LOAD
#define GS_TB_AUX_LOAD 0
#endif
#ifndef GS_TB_AUX_STORE
#define GS_TB_AUX_STORE 0
#endif
// GS_VS_ABLATE_HALO 1 = TIMING ONLY, WRONG RESULTS: the march with full difference sharing without its halo board (no LDS
//                  writes or reads; a lane's halo columns are its own): what the LDS round trip per level costs.
#ifndef GS_VS_ABLATE_HALO
#define GS_VS_ABLATE_HALO 0
#endif
// GS_WIN_LATE_ROW  1 = the persistent window kernel computes the last of a wave's middle rows behind the step's barrier,
//                  after issuing the reads of the neighbouring waves' rows (hides the LDS read burst of 16 waves in
//                  lock-step); 0 = all middle rows before the barrier (round 4's order).
#ifndef GS_WIN_LATE_ROW
#define GS_WIN_LATE_ROW 0
#endif
// GS_WIN_TAGGED    1 = the persistent window kernel's apron exchange hands cells over as data-tagged 8-byte granules
//                  {value, exchange number}: a workgroup stores its ring and at once polls the granules of its own apron --
//                  no drain, no flag, no flag poll, one barrier instead of two per exchange (MI355X_MICROARCH.md, price
//                  list: handoff-1to1 against handoff-flag).  0 = round 4's form: ring, drain, barrier, flag, poll, barrier,
//                  apron loads.  The exchange planes are sized for granules either way.
#ifndef GS_WIN_TAGGED
#define GS_WIN_TAGGED 1
#endif
// GS_WIN_FIRST_POLL_SLEEP / GS_WIN_POLL_SLEEP  (tagged exchange) s_sleep units of 64 clocks before the first poll of the apron's
//                  granules and between polls.
#ifndef GS_WIN_FIRST_POLL_SLEEP
#define GS_WIN_FIRST_POLL_SLEEP 20
#endif
// GS_WIN_EDGE_POLLS_AT_ONCE  1 = only the windows inside the grid wait before their first poll (measured: 497-501 k against
//                  508-522 k with every window waiting, profiles/r06_window_kernel.md)
#ifndef GS_WIN_EDGE_POLLS_AT_ONCE
#define GS_WIN_EDGE_POLLS_AT_ONCE 0
#endif
#ifndef GS_WIN_POLL_SLEEP
#define GS_WIN_POLL_SLEEP 0
#endif
// GS_WIN_PAIR_SYNC 1 = inside a step of the persistent window kernel a wave waits for the waves above and below it only (one LDS
//                  word per wave) instead of the workgroup's barrier: the waves of a SIMD get out of phase, and the LDS read burst
//                  and the lone last wave behind every barrier go.  0 = one workgroup barrier per step.
#ifndef GS_WIN_PAIR_SYNC
#define GS_WIN_PAIR_SYNC 1
#endif
// GS_WIN_PAIR_PRIO (with GS_WIN_PAIR_SYNC) 0 = the SIMD's own arbitration (priority, then age); 1 = s_setprio rotates over a SIMD's
//                  four waves with the step; 2 = by feedback: a wave that had to wait for its neighbours steps back, one whose
//                  neighbours are a step ahead goes first.  GS_WIN_PAIR_POLL_SLEEP: s_sleep units between polls of the words.
#ifndef GS_WIN_PAIR_PRIO
#define GS_WIN_PAIR_PRIO 2
#endif
// GS_WIN_WAVES_LEAVE_ALONE (with GS_WIN_PAIR_SYNC and GS_WIN_TAGGED) 1 = no workgroup barrier behind an exchange either: a wave that has
//                  its apron goes on; 0 = one barrier per exchange (the waves agree whether the launch goes on).
#ifndef GS_WIN_WAVES_LEAVE_ALONE
#define GS_WIN_WAVES_LEAVE_ALONE 1
#endif
#ifndef GS_WIN_PRIO_AHEAD
#define GS_WIN_PRIO_AHEAD 0
#endif
#ifndef GS_WIN_PRIO_LEVEL
#define GS_WIN_PRIO_LEVEL 1
#endif
#ifndef GS_WIN_PRIO_BEHIND
#define GS_WIN_PRIO_BEHIND 3
#endif
#ifndef GS_WIN_PAIR_POLL_SLEEP
#define GS_WIN_PAIR_POLL_SLEEP 0
#endif
// GS_WIN_TRACE     (defined = on; tools/window_timeline.py) wave 0 of every workgroup of the persistent window kernel
//                  stamps the 100 MHz real-time counter at seven points of each of its last 8 super-steps.
// GS_TB_TRACE      (defined = on; tools/wave_timeline.py) every wave of gs_step_tb_k leaves five stamps of the
//                  100 MHz real-time counter -- entry, first level-0 rows used (tick 3), level pipeline full (tick
//                  2K), last level-0 row taken (tick nticks - 2K), exit -- plus the shader-clock counter at entry
//                  and exit, its hardware id and its unit in a device buffer that gs_debug_trace_read*() copy out.
#if defined(GS_TB_TRACE) && defined(__HIPCC__)
constexpr int kTraceWords = 8, kTraceUnits = 1 << 17;
__device__ unsigned long long gs_trace_buf[kTraceWords * kTraceUnits];
__device__ __forceinline__ unsigned long long trace_now()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define GS_TRACE_AT(COND, SLOT) do { if (COND) ts[SLOT] = trace_now(); } while (0)
#define GS_TRACE_PARAM , unsigned long long (&ts)[5] /* tb_march's extra parameter ... */
#define GS_TRACE_ARG , ts                            /* ... and what gs_step_tb_k passes for it */
#else
#define GS_TRACE_AT(COND, SLOT) do { } while (0)
#define GS_TRACE_PARAM
#define GS_TRACE_ARG
#endif

// ---- (2) run-time ----------------------------------------------------------------------------------------
// Integer environment switch, clamped to [lo, hi]; `unset` when the variable is absent or empty.
//   GS_HIP_TRACE_LAUNCH   1 = print the first 64 kernel launches of the process (label, row ranges, layout)
//   GS_HIP_TRACE_TUNER    1 = print every timing window of gs_run's on-line tuner and its choice (and gs_fields_place's probes)
//   GS_HIP_PLACE_ALL      1 = gs_fields_place draws all its candidates even when it has two blocks of each kind (diagnostics)
//   GS_HIP_EDGE_KINDS     0 = every edge unit of the marching kernel takes the general path (default 1: cheap kinds)
//   GS_HIP_EDGE_SPLIT     0 / 1 = never / always dispatch edge units as two half-height units (default: by size)
//   GS_HIP_FAIR           0 / 1 = never / always run one-round launches as in-step 16-wave workgroups
//   GS_HIP_FAIR_FROM      progress (0 ... 256) from which the in-step form steers priorities (default 0)
//   GS_HIP_XCD_M          0 = plain workgroup order; n = XCD-aware renumbering in groups of 8 n (default: 16 on
//                         multi-round launches); GS_HIP_XCD_M_STREAM: the same for the single-step kernel
//   GS_HIP_TILE_LDS_FLOOR bytes of dynamic LDS the LDS-window kernel asks for at least (limits workgroups per CU)
//   GS_HIP_WINDOW_PATIENCE polls (2-3 us each) a wave of the persistent window kernel waits for its neighbours' cells
//                         before the launch gives up (default 2^20, 2-3 s; tests set 1 to provoke it)
//   GS_HIP_WINDOW_WAVES   "left,interior,right": waves in use per window of the left-most / inner / right-most tile column
//                         of the persistent window kernel's tiling (gs_window.cpp: plan_windows)
constexpr int kGsXcdGroupMax = 512; // 8 * 512 workgroups per renumbered group at most
inline int gs_env_int(const char *name, int unset, int lo, int hi)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return unset;
    long x = std::strtol(v, nullptr, 10);
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return (int)x;
}
