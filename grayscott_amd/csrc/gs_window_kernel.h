// gs_window_kernel.h -- grids of one round of register-resident windows (the reference's default 1080 x 1920): the whole
// gs_run in one persistent launch, aprons traded between workgroups inside it (gs_run_window_k).
// Part of the gfx950 step kernels: included by gs_step_kernels.hip (which sets GS_MATH_FUSED / GS_TB_OP_ONLY and the
// GS_SUFFIX / GS_TAP macros) inside one translation unit per arithmetic flavour; not a header to include elsewhere.
#pragma once

namespace {

// ------------------------------------------------------------------------------------
// Grids of ONE round of register-resident windows (1.5-2.3 M cells, what 256 windows of 72 x 120 owned cells cover: the reference's default 1080 x 1920): the whole
// gs_run in one persistent launch, aprons traded between workgroups inside it.
//
// At these sizes a pass of the marching kernel is 19 us for 4 steps of which ~9 are fixed (launch gap, dispatch,
// first-rows burst, level-pipeline fill on memory latency) and its 10-row units recompute 30 % of their rows
// (profiles/archive/r03_sweeps.md, sections 1-4).  Here a workgroup of 16 waves owns a window of 16 * RPW rows x 128
// columns for the whole run: a wave keeps RPW whole rows in registers, two columns per lane (10 cells per lane at
// RPW = 5).  Per step the columns next to a lane's two come from the adjacent lanes (DPP wave shifts), only the
// first and the last row of a wave's band go through LDS for the waves above and below (double-buffered by the
// step's parity), and every cell is updated by the same cell<> code as in every other kernel: bit-identical.
// Inside a step a wave waits for the waves above and below it only (one LDS word per wave says which step's rows it
// has published; round 6 -- a workgroup barrier per step until then), after the RPW - 2 rows that need nothing from
// other waves, and sets its priority by how it stands to them.  The window's outer K cells are an apron: they lose
// their validity one ring per step.  After K steps every wave stores its part of the K-cell ring of the cells the
// workgroup OWNS (the window shrunk by K) into an exchange plane as data-tagged granules {value, exchange number}
// (8 bytes, one sc1 store) and polls the granules of its own apron cells with sc1 loads until they carry the number it
// waits for -- the hand-off MI355X_MICROARCH.md prices as handoff-1to1; no drain, no flag, and no barrier: a wave
// that has its apron goes on.  (Round 4's form -- ring, drain, barrier, flag, poll of the neighbours' flags, barrier,
// apron loads -- is -DGS_WIN_TAGGED=0.)  Exchanges alternate between two sets of exchange planes, so a wave that is
// one exchange ahead never overwrites what a neighbour still has to read.  The input planes are only read and the
// output planes only written at the very end.  Every poll of the exchange is bounded: a wave that runs out of
// patience (its neighbours are not resident: the GPU is shared with another long-running kernel) sets a sticky abort
// word and ends, and every other wave leaves at its next exchange; gs_sync reports it and the host replays the launch.
// Edge windows use the cheap kinds of edge path of the marching kernel (cell<2>, cell<3>, general rows only for the
// grid's first and last row) under the clipped rule and interior code over zeros under the zero-halo rule.
// ------------------------------------------------------------------------------------
constexpr int kWinCols = 128;              // window columns: 64 lanes x 2
// floats per published row: two arrays of 66 -- the lanes' first columns (window column 2 l at element 1 + l), then their
// second columns (2 l + 1 at 66 + 1 + l) -- so that a lane's own columns and the two next to them are two conflict-free
// ds_read2_b32 (first columns of lanes l, l + 1; second columns of lanes l - 1, l).  (Round 4 kept a row in column order
// and read a float2 and two odd-offset scalars: 37 % of the LDS pipe's active cycles were bank conflicts.)
constexpr int kWinHalf = 66;
constexpr int kWinPitch = 2 * kWinHalf;
constexpr int kWinWaves = 16;
__host__ __device__ constexpr int win_rows(int rpw) { return kWinWaves * rpw; }
// 2 buffers x 2 species x 16 waves x (first row, last row) x pitch
__host__ __device__ constexpr size_t win_rows_floats() { return (size_t)2 * 2 * kWinWaves * 2 * kWinPitch; }
// ... + one word per wave: the number of the step whose rows the wave has published (GS_WIN_PAIR_SYNC)
__host__ __device__ constexpr size_t win_lds_bytes() { return (win_rows_floats() + kWinWaves) * sizeof(float); }

// The S / SE / SW taps of the cells of row `z` with respect to the row `p` below it, in the slots cells_vshare (gs_march.h)
// keeps them in: what that function leaves in its carry after the row z -- here for a row that is not updated at this
// point (the row above a wave's band, which another wave owns; the band's first row, whose own update waits for the neighbouring waves' rows).
__device__ __forceinline__ TapCarry<2> win_carry_of(const GsStepArgs &a, const RowT<2> &z, const RowT<2> &p)
{
    TapCarry<2> c;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        c.s_u[i] = half_diff(p.u[i + 1], z.u[i + 1]);          c.s_v[i] = half_diff(p.v[i + 1], z.v[i + 1]);          // S of cell i + 1
        c.se_u[i] = a.w[2][2] * (p.u[i + 1] - z.u[i]);         c.se_v[i] = a.w[2][2] * (p.v[i + 1] - z.v[i]);         // SE of cell i
        c.sw_u[i] = a.w[2][0] * (p.u[i + 1] - z.u[i + 2]);     c.sw_v[i] = a.w[2][0] * (p.v[i + 1] - z.v[i + 2]);     // SW of cell i + 2
    }
    return c;
}

#if defined(GS_WIN_TRACE)
__device__ unsigned long long gs_win_trace[1024 * 8 * 8];
#endif
#if defined(GS_WIN_TRACE) && GS_WIN_TRACE == 2
// GS_WIN_TRACE=2 (tools/window_step_timeline.py): EVERY wave of the first 256 workgroups stamps four points of each of the
// launch's last 4 steps: step begins, at the barrier, past the barrier, the neighbouring waves' rows are in registers.
#define GS_WIN_STEP_AT(SLOT)                                                                                       \
    do {                                                                                                           \
        if (lane == 0 && blockIdx.x < 256) {                                                                       \
            unsigned long long t_;                                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            gs_win_trace[((blockIdx.x * 16 + wave) * 4 + (step & 3)) * 4 + (SLOT)] = t_;                           \
        }                                                                                                          \
    } while (0)
#define GS_WIN_TRACE_AT(SLOT) do { } while (0)
#elif defined(GS_WIN_TRACE) && GS_WIN_TRACE == 4
// GS_WIN_TRACE=4 (tools/window_wave_budget.py): every wave adds up, in shader clocks, what it spends waiting for the waves above
// and below it inside the steps (word 0; word 3: how many of those waits found the rows not there yet), between its last
// step and its apron (word 1), and in all (word 2).
#define GS_WIN_BUDGET 1
#define GS_WIN_BUDGET_ADD(SLOT, V) do { if (lane == 0 && blockIdx.x < 256) atomicAdd(&gs_win_trace[(blockIdx.x * 16 + wave) * 4 + (SLOT)], (unsigned long long)(V)); } while (0)
#define GS_WIN_STEP_AT(SLOT) do { } while (0)
#define GS_WIN_TRACE_AT(SLOT) do { } while (0)
#elif defined(GS_WIN_TRACE) && GS_WIN_TRACE == 3
// GS_WIN_TRACE=3 (tools/window_wave_timeline.py): EVERY wave of the first 256 workgroups stamps three points of each of its last
// four super-steps -- begins, steps done (before the ring stores), apron in -- and leaves the number of polls in the fourth word.
#define GS_WIN_STEP_AT(SLOT) do { } while (0)
#define GS_WIN_TRACE_AT(SLOT)                                                                                      \
    do {                                                                                                           \
        if (lane == 0 && wg < 256 && s >= supers - 4 && ((SLOT) == 0 || (SLOT) == 1 || (SLOT) == 4)) {             \
            unsigned long long t_;                                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            gs_win_trace[((wg * 16 + wave) * 4 + ((s - (supers - 4)) & 3)) * 4 + ((SLOT) == 4 ? 2 : (SLOT))] = t_; \
            if ((SLOT) == 4) gs_win_trace[((wg * 16 + wave) * 4 + ((s - (supers - 4)) & 3)) * 4 + 3] = (unsigned long long)trace_polls; \
        }                                                                                                          \
    } while (0)
#elif defined(GS_WIN_TRACE)
#define GS_WIN_STEP_AT(SLOT) do { } while (0)
#define GS_WIN_TRACE_AT(SLOT)                                                                                      \
    do {                                                                                                           \
        if (wave == 0 && lane == 0 && s >= supers - 8) {                                                           \
            unsigned long long t_;                                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            gs_win_trace[(wg * 8 + ((s - (supers - 8)) & 7)) * 8 + (SLOT)] = t_;                                   \
            if ((SLOT) == 0) gs_win_trace[(wg * 8 + ((s - (supers - 8)) & 7)) * 8 + 7] = (unsigned long long)EDGE;  \
        }                                                                                                          \
    } while (0)
#else
#define GS_WIN_STEP_AT(SLOT) do { } while (0)
#define GS_WIN_TRACE_AT(SLOT) do { } while (0)
#endif

// `n` time steps of a window.  EDGE: 0 = window inside the grid; 1 = general path for every cell; 2 / 3 = window on the
// grid's left / right edge, touching neither top nor bottom (cell<2> / cell<3>); 4 = window on the top or bottom edge
// only (interior code but for the grid's first / last row, which take the general cell); 5 / 6 = corner windows,
// left / right (cell<2> / cell<3> but for the grid's first / last row); 7 = edge window under the zero-halo rule:
// interior code over cells that are zeros outside the grid and stay zeros.  `step` counts the steps of the launch
// (parity of the LDS buffer).
template <int RPW, int EDGE, int FAST, int ZH>
__device__ __forceinline__ void window_steps(const GsStepArgs &a, float *lds, int n, int &step, int gr, int gc, int wave, int lane,
                                             float (&u)[RPW][2], float (&v)[RPW][2])
{
    constexpr int P = kWinPitch;
    constexpr bool ROWS = EDGE == 1 || EDGE == 4 || EDGE == 5 || EDGE == 6;
    constexpr int SIDE = (EDGE == 2 || EDGE == 5) ? 2 : ((EDGE == 3 || EDGE == 6) ? 3 : 0);
    // row `which` (0 = first, 1 = last row of a wave's band) of wave w, species sp, buffer buf
    // (element of this lane's FIRST column; its second column is kWinHalf further on)
    auto row_of = [&](int buf, int sp, int w, int which) { return lds + ((((buf * 2 + sp) * kWinWaves + w) * 2 + which) * P) + 1 + lane; };
#if GS_WIN_PAIR_SYNC
    // Waves wait for their two neighbours only: a wave says which step's rows it has published (one LDS word per wave, written
    // behind the rows: the LDS serves a wave's operations in order) and, where a barrier stood, polls the words of the waves
    // above and below.  Two buffers suffice as with the barrier: a wave overwrites the rows of step s at step s + 2, which it
    // reaches only after both neighbours have published step s + 1, i.e. have read what they needed of step s.
    // (an LDS pointer by type: through a generic pointer a volatile access is a flat_load with a 64-bit address)
    typedef __attribute__((address_space(3))) volatile int *SaidPtr;
    const SaidPtr said = (SaidPtr)(lds + win_rows_floats());
    auto announce = [&](int st) {
        asm volatile("" ::: "memory");
        if (lane == 0) said[wave] = st + 1;
        asm volatile("" ::: "memory");
    };
    auto await = [&](int st) {
        bool waited = false;
#if defined(GS_WIN_BUDGET)
        const unsigned long long await_t0 = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
            const int fa = said[wave > 0 ? wave - 1 : 0], fb = said[wave < kWinWaves - 1 ? wave + 1 : kWinWaves - 1];
            const int m = __builtin_amdgcn_readfirstlane(fa < fb ? fa : fb);
            if (m > st) {
#if GS_WIN_PAIR_PRIO == 2
                // the SIMD issues by priority, then age: left alone, its oldest wave runs a step ahead and its youngest finishes
                // the super-step alone, at half the issue rate.  A wave that had to wait is ahead and steps back; one whose
                // neighbours are a step further is behind and goes first.
                if (waited) __builtin_amdgcn_s_setprio(GS_WIN_PRIO_AHEAD);
                else if (m > st + 1) __builtin_amdgcn_s_setprio(GS_WIN_PRIO_BEHIND);
                else __builtin_amdgcn_s_setprio(GS_WIN_PRIO_LEVEL);
#endif
                break;
            }
            waited = true;
#if GS_WIN_PAIR_POLL_SLEEP > 0
            __builtin_amdgcn_s_sleep(GS_WIN_PAIR_POLL_SLEEP);
#endif
        }
#if defined(GS_WIN_BUDGET)
        GS_WIN_BUDGET_ADD(0, __builtin_amdgcn_s_memtime() - await_t0);
        if (waited) GS_WIN_BUDGET_ADD(3, 1);
#endif
        asm volatile("" ::: "memory");
    };
#define GS_WIN_STEP_SYNC(ST) await(ST)
#define GS_WIN_STEP_SAY(ST) announce(ST)
#else
#define GS_WIN_STEP_SYNC(ST) __syncthreads()
#define GS_WIN_STEP_SAY(ST) do { } while (0)
#endif
    // cells outside the grid are zeros and stay zeros: rows are wave-uniform (scalar tests), columns per lane
    bool col_in[2];
    uint32_t la[2], ra[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        la[j] = ((EDGE == 1 || SIDE == 2) && j == 0 && gc == 0) ? 0xffffffffu : 0u; // gc is even: only a lane's first cell
        ra[j] = ((EDGE == 1 || SIDE == 3) && (gc + j + 1 >= a.cols)) ? 0xffffffffu : 0u;
        col_in[j] = gc + j >= 0 && gc + j < a.cols;
    }
    const int wa = wave > 0 ? wave - 1 : 0, wb = wave < kWinWaves - 1 ? wave + 1 : kWinWaves - 1;
    // The columns next to a lane's two come from the adjacent lanes by DPP wave shifts: VALU work (4 % of a step) rather
    // than the LDS crossbar, which 16 waves in lock-step all want at the same moment (ds_bpermute_b32: a step 24 % longer,
    // profiles/r04_sweeps.md, section 2).
    auto widen = [](const float (&cu)[2], const float (&cv)[2]) {
        RowT<2> w;
        w.u[1] = cu[0]; w.u[2] = cu[1]; w.v[1] = cv[0]; w.v[2] = cv[1];
        w.u[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cu[1]), 0x138, 0xf, 0xf, true));
        w.u[3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cu[0]), 0x130, 0xf, 0xf, true));
        w.v[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cv[1]), 0x138, 0xf, 0xf, true));
        w.v[3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cv[0]), 0x130, 0xf, 0xf, true));
        return w;
    };
    // One cell row: old rows (m, z, p) -> new values of row r, written in place.
    // (windows that hold the grid's first or last row test every row's position per step: `grow` is made opaque once per
    // step, or the compiler keeps fifteen loop-invariant row conditions as 64-bit masks in thirty SGPRs across the loop
    // and spills them to VGPR lanes -- 25 v_readlane inside the step loop of the corner windows)
    int grow = gr;
    const int nrows = a.rows;
    auto update = [&](int r, const RowT<2> &m, const RowT<2> &z, const RowT<2> &p) {
        const int row = grow + r; // wave-uniform
        if (EDGE != 0 && (row < 0 || row >= nrows)) return; // a row outside the grid: zeros that stay zeros
        const bool mrow = !ROWS || row > 0, prow = !ROWS || row + 1 < nrows;
        float nu[2], nv[2];
        if constexpr (EDGE == 0 || EDGE == 7) {
            cells_interior<FAST, 2, ZH>(a, m, z, p, nu, nv);
        } else if constexpr (EDGE == 4 || EDGE == 5 || EDGE == 6) {
            if (mrow && prow) {
                if constexpr (EDGE == 4) {
                    cells_interior<FAST, 2, ZH>(a, m, z, p, nu, nv);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) cell<SIDE, FAST, RowT<2>, ZH>(a, m, z, p, 1 + j, true, true, la[j], ra[j], nu[j], nv[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) cell<1, FAST, RowT<2>, ZH>(a, m, z, p, 1 + j, mrow, prow, la[j], ra[j], nu[j], nv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) cell<EDGE, FAST, RowT<2>, ZH>(a, m, z, p, 1 + j, mrow, prow, la[j], ra[j], nu[j], nv[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = EDGE == 0 || col_in[j];
            u[r][j] = in ? nu[j] : 0.0f;
            v[r][j] = in ? nv[j] : 0.0f;
        }
    };
    // the first and the last row of this wave's band, for the waves above and below
    auto publish = [&](int buf) {
        auto put = [](float *p, float c0, float c1) { p[0] = c0; p[kWinHalf] = c1; }; // one ds_write2_b32
        put(row_of(buf, 0, wave, 0), u[0][0], u[0][1]);
        put(row_of(buf, 1, wave, 0), v[0][0], v[0][1]);
        put(row_of(buf, 0, wave, 1), u[RPW - 1][0], u[RPW - 1][1]);
        put(row_of(buf, 1, wave, 1), v[RPW - 1][0], v[RPW - 1][1]);
    };
    // A step: publish, the rows that need nothing from other waves (the other waves' rows arrive meanwhile), the wait
    // for the waves above and below (GS_WIN_STEP_SYNC: their LDS words; the workgroup's barrier in the round-5 build), the
    // rows above and below from LDS, the band's first and last row.  (Reads first and the publish for the next step
    // right before the barrier -- the LDS latency behind the middle rows -- was measured: the waves of a workgroup
    // drift apart, 418 k against 461 k at 1080 x 1920, profiles/r04_window_kernel.md.)
    // Full difference sharing inside a wave's band (windows inside the grid, FAST & 4: the stencil's diagonal weights pair
    // up, strict build): the three taps a row takes from the row above it are, negated, the three taps that row took from
    // this one -- cells_vshare, the marching kernel's form (gs_march.h), whose carry walks down the band.  Only the band's
    // first row forms its N taps afresh (from the row another wave published), and the last row its S taps: 5 instead of
    // 10 three-tap sets per species and band, 440 instead of 520 arithmetic instructions per wave and step.
    constexpr bool SHARE = EDGE == 0 && (FAST & 5) == 5 && !GS_MATH_FUSED && RPW >= 3;
    for (int s = 0; s < n; ++s, ++step) {
        const int buf = step & 1;
        if constexpr (ROWS) {
            asm volatile("" : "+s"(grow));
            grow = __builtin_amdgcn_readfirstlane(grow);
        }
        GS_WIN_STEP_AT(0);
#if GS_WIN_PAIR_SYNC && GS_WIN_PAIR_PRIO == 1
        // (every step another of a SIMD's four waves goes first: SIMD = wave % 4)
        switch (((wave >> 2) + step) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        publish(buf);
        GS_WIN_STEP_SAY(step);
        auto shared_step = [&]() {
          if constexpr ((FAST & 5) == 5 && !GS_MATH_FUSED) {
            const RowT<2> first = widen(u[0], v[0]), second = widen(u[1], v[1]); // old rows 0 and 1: row 0 waits for the row above
            TapCarry<2> c = win_carry_of(a, first, second);                     // row 0's S / SE / SW taps: row 1's N / NW / NE
            RowT<2> cur = second;
            float nu[2], nv[2];
#pragma unroll
            for (int r = 1; r < RPW - 1; ++r) {
                const RowT<2> next = widen(u[r + 1], v[r + 1]);                 // old row r + 1 (not overwritten yet)
                cells_vshare<FAST, 2>(a, cur, next, c, nu, nv);
                u[r][0] = nu[0]; u[r][1] = nu[1]; v[r][0] = nv[0]; v[r][1] = nv[1];
                cur = next;
            }
            GS_WIN_STEP_AT(1);
            GS_WIN_STEP_SYNC(step);
            GS_WIN_STEP_AT(2);
            RowT<2> above, below;
            {
                auto get = [](const float *p, float (&w)[4]) { w[1] = p[0]; w[3] = p[1]; w[0] = p[kWinHalf - 1]; w[2] = p[kWinHalf]; };
                get(row_of(buf, 0, wa, 1), above.u);
                get(row_of(buf, 1, wa, 1), above.v);
                get(row_of(buf, 0, wb, 0), below.u);
                get(row_of(buf, 1, wb, 0), below.v);
            }
            GS_WIN_STEP_AT(3);
            // the band's last row: its N taps from the carry, its S taps from the row below it (another wave's)
            cells_vshare<FAST, 2>(a, cur, below, c, nu, nv);
            u[RPW - 1][0] = nu[0]; u[RPW - 1][1] = nu[1]; v[RPW - 1][0] = nv[0]; v[RPW - 1][1] = nv[1];
            // the band's first row: the taps of the row above it (another wave's) formed here, its own S taps once more
            TapCarry<2> ca = win_carry_of(a, above, first);
            cells_vshare<FAST, 2>(a, first, second, ca, nu, nv);
            u[0][0] = nu[0]; u[0][1] = nu[1]; v[0][0] = nv[0]; v[0][1] = nv[1];
          }
        };
        if constexpr (SHARE) {
            shared_step();
            continue;
        }
        // ... and in the windows on the grid's top or bottom edge (not in a corner), for every wave whose band and the rows
        // above and below it lie inside the grid: all but the wave that holds the grid's first or last row.  Those windows
        // set the pace of the whole grid (every workgroup waits for its neighbours at every exchange): 12.2-12.3 us per 4
        // steps against the 10.96 of the windows inside (profiles/r06_window_kernel.md).
        if constexpr (EDGE == 4 && (FAST & 5) == 5 && !GS_MATH_FUSED && RPW >= 3) {
            if (grow >= 1 && grow + RPW < nrows) { // (wave-uniform)
                shared_step();
                continue;
            }
        }
        // Top down with a sliding window of widened OLD rows: a row is widened just before the row above it is
        // overwritten, so at most five widened rows are alive -- the window of three, old row 1 (kept for row 0) and
        // old row RPW - 2 (for the last row) -- instead of all RPW + 2.
        RowT<2> first = widen(u[0], v[0]);                 // old row 0
        RowT<2> second = widen(u[RPW > 1 ? 1 : 0], v[RPW > 1 ? 1 : 0]); // old row 1: needed again for row 0
        RowT<2> prev = first, cur = second;
        // GS_WIN_LATE_ROW (gs_experiments.h): the last of the rows that need nothing from other waves is computed BEHIND
        // the barrier, after the reads of the neighbouring waves' rows have been issued -- all 16 waves of the workgroup
        // issue those reads at the same moment, and the LDS pipe serves them one after the other
        constexpr int kLate = (GS_WIN_LATE_ROW && RPW >= 4) ? 1 : 0;
#pragma unroll
        for (int r = 1; r < RPW - 1 - kLate; ++r) {
            const RowT<2> next = widen(u[r + 1], v[r + 1]); // old row r + 1 (not overwritten yet)
            update(r, prev, cur, next);
            prev = cur;
            cur = next;
        }
        // now (kLate = 0): prev = old row RPW - 2, cur = old row RPW - 1 (RPW >= 3); RPW == 2: prev = old row 0, cur = old row 1
        GS_WIN_STEP_SYNC(step);
        RowT<2> above, below;
        {
            // [0] = second column of lane - 1, [1] [2] = own columns, [3] = first column of lane + 1
            auto get = [](const float *p, float (&w)[4]) { w[1] = p[0]; w[3] = p[1]; w[0] = p[kWinHalf - 1]; w[2] = p[kWinHalf]; };
            get(row_of(buf, 0, wa, 1), above.u);
            get(row_of(buf, 1, wa, 1), above.v);
            get(row_of(buf, 0, wb, 0), below.u);
            get(row_of(buf, 1, wb, 0), below.v);
        }
        if constexpr (kLate) {
            const RowT<2> next = widen(u[RPW - 1], v[RPW - 1]);
            update(RPW - 2, prev, cur, next);
            prev = cur;
            cur = next;
        }
        if (RPW == 1) {
            update(0, above, first, below);
        } else {
            update(0, above, first, second);
            update(RPW - 1, prev, cur, below);
        }
    }
}

#undef GS_WIN_STEP_SYNC
#undef GS_WIN_STEP_SAY

__device__ __forceinline__ __amdgpu_buffer_rsrc_t win_rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

// The whole run of one workgroup: super-steps of K steps, an exchange after each but the last, the final store.  One
// instantiation per kind of window (the kernel branches ONCE: with the branch inside the loop the compiler hoists the
// loop-invariant values of every kind above it and the register file does not hold them all).
// GS_WIN_TRACE (diagnostic builds, tools/window_timeline.py): wave 0 of every workgroup stamps the 100 MHz real-time
// counter at seven points of each of its last 8 super-steps: start, steps done, ring stored and drained, barrier
// passed, poll matched, barrier passed, apron loaded.

template <int RPW, int EDGE, int FAST, int ZH>
__device__ __forceinline__ void window_run(const GsStepArgs &a, const GsWindowArgs &x, const GsWindowDesc *d, int OH, int OW, float *lds,
                                           int *go, int wg, int gr, int gc, int wave, int lane, float (&u)[RPW][2], float (&v)[RPW][2])
{
    // OH x OW: the cells this workgroup owns = window rows [K, K + OH) x window columns [K, K + OW)
    constexpr int SC1 = 16;
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int K = x.k, wc = 2 * lane; // wc: this lane's first window column
    int step = 0;
#if defined(GS_WIN_BUDGET)
    const unsigned long long run_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int supers = (x.steps + K - 1) / K;
    for (int s = 0; s < supers; ++s) {
        int trace_polls = 0; // (GS_WIN_TRACE == 3)
        (void)trace_polls;
        GS_WIN_TRACE_AT(0);
        // the short super-step first
        window_steps<RPW, EDGE, FAST, ZH>(a, lds, (s == 0 && x.steps % K) ? x.steps % K : K, step, gr, gc, wave, lane, u, v);
        GS_WIN_TRACE_AT(1);
        if (s == supers - 1) break;
#if defined(GS_WIN_BUDGET)
        const unsigned long long xchg_t0 = __builtin_amdgcn_s_memtime();
#endif
#if GS_WIN_TAGGED
        // ---- exchange s, data-tagged granules: ring out, then every lane polls the granules of its own apron cells ----
        // A granule is {value, tag}: 8 bytes, naturally aligned, written by ONE sc1 store (a lane's two columns: one 16-byte
        // store of two granules) and read by sc1 loads; the tag is the number of the exchange (epoch + s + 1: unique over
        // the launches of a context; the planes are zeroed when the numbering starts over).  A reader that finds the tag it
        // waits for has the value that belongs to it: no drain, no flag, no barrier between a neighbour's stores and my
        // loads (the hand-off form MI355X_MICROARCH.md prices as handoff-1to1, 1.7-1.9x cheaper than a drained flag).
        // Exchanges alternate between two sets of planes: a neighbour that is one exchange ahead never overwrites what I
        // still have to read (it cannot be two ahead: it needs my ring of the exchange in between).
        // (granules travel as integer vectors: a tag is a small integer -- as a float a sub-normal, which this build's float
        // mode would flush to zero in any instruction that treats it as a number)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t xu = win_rsrc(x.xu[s & 1]), xv = win_rsrc(x.xv[s & 1]);
        const int tag_i = x.epoch + s + 1;
        const unsigned tag = (unsigned)tag_i;
        auto bits = [](float f) { return __builtin_bit_cast(unsigned, f); };
        auto flt = [](unsigned w) { return __builtin_bit_cast(float, w); };
        const bool lane_in = gc >= 0 && gc < a.cols;
        const bool lane_apron = (wc < K || wc >= K + OW) && wc < 2 * K + OW && lane_in;
        const bool in1 = gc + 1 < a.cols;
        auto wants = [&](int r) { // (does this lane hold an apron cell in row r of its band?)
            const int wr = wave * RPW + r;
            const bool row_in = gr + r >= 0 && gr + r < a.rows;
            const bool row_apron = (wr < K || wr >= K + OH) && wr < 2 * K + OH;
            return row_in && ((row_apron && lane_in && wc < 2 * K + OW) || (lane_apron && wr < 2 * K + OH));
        };
        const bool lane_owned = wc >= K && wc < K + OW && gc < a.cols;
        const bool lane_ring = wc < 2 * K || wc >= OW;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int wr = wave * RPW + r; // wave-uniform
            const bool row_owned = wr >= K && wr < K + OH && gr + r < a.rows;
            const bool row_ring = wr < 2 * K || wr >= OH;
            if (row_owned && lane_owned && (row_ring || lane_ring)) {
                const int off = ((gr + r) * a.pitch + gc) * (int)(2 * sizeof(float)); // granules: 8 bytes per cell
                const v4u su = {bits(u[r][0]), tag, bits(u[r][1]), tag}, sv = {bits(v[r][0]), tag, bits(v[r][1]), tag};
                __builtin_amdgcn_raw_buffer_store_b128(su, xu, off, 0, SC1);
                __builtin_amdgcn_raw_buffer_store_b128(sv, xv, off, 0, SC1);
            }
        }
        GS_WIN_TRACE_AT(2);
        GS_WIN_TRACE_AT(3);
        bool failed = false;
        {
            int spins = 0;
#if GS_WIN_FIRST_POLL_SLEEP > 0
            // The neighbours' stores need about half a microsecond to land, and a poll that comes too early costs a whole
            // round trip (1.8 us): every window waits that long before its first poll (496 k -> 508-522 k at 1080 x 1920; 8 ... 28
            // units tried, 20-24 best; the edge windows polling at once, whose neighbours have stored when they arrive: 497-501 k).
            if constexpr (EDGE == 0 || !GS_WIN_EDGE_POLLS_AT_ONCE) __builtin_amdgcn_s_sleep(GS_WIN_FIRST_POLL_SLEEP);
#endif
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    if (wants(r)) {
                        const int off = ((gr + r) * a.pitch + gc) * (int)(2 * sizeof(float));
                        const v4u fu = __builtin_amdgcn_raw_buffer_load_b128(xu, off, 0, SC1);
                        const v4u fv = __builtin_amdgcn_raw_buffer_load_b128(xv, off, 0, SC1);
                        const bool here = fu[1] == tag && fv[1] == tag && (!in1 || (fu[3] == tag && fv[3] == tag));
                        if (here) {
                            u[r][0] = flt(fu[0]); u[r][1] = in1 ? flt(fu[2]) : 0.0f;
                            v[r][0] = flt(fv[0]); v[r][1] = in1 ? flt(fv[2]) : 0.0f;
                        }
                        ok = ok && here;
                    }
                }
                if (!__builtin_amdgcn_ballot_w64(!ok)) break; // every lane of this wave has its apron
                // bounded: neighbours that are not resident never store (the GPU is shared with another long-running
                // kernel); a workgroup of this launch that gave up says so in the abort word
                if (++spins > x.patience || ((spins & 63) == 0 && __builtin_amdgcn_raw_buffer_load_b32(win_rsrc(x.abort), 0, 0, SC1) != 0)) { failed = true; break; }
#if GS_WIN_POLL_SLEEP > 0
                __builtin_amdgcn_s_sleep(GS_WIN_POLL_SLEEP);
#endif
            }
            trace_polls = spins + 1;
            if (failed && lane == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(x.seq, win_rsrc(x.abort), 0, 0, SC1);
                // (diagnostics, read by resolve_window under GS_HIP_TRACE_TUNER: which wave of which workgroup ran out of patience
                // at which exchange, after how many polls -- the flag words are not used by this form of the exchange)
                __builtin_amdgcn_raw_buffer_store_b32((wave << 24) | (s & 0xffffff), win_rsrc(x.flags), wg * 8, 0, SC1);
                __builtin_amdgcn_raw_buffer_store_b32(spins, win_rsrc(x.flags), wg * 8 + 4, 0, SC1);
                *go = 0;
            }
        }
        GS_WIN_TRACE_AT(4);
#if defined(GS_WIN_BUDGET)
        GS_WIN_BUDGET_ADD(1, __builtin_amdgcn_s_memtime() - xchg_t0);
#endif
#if GS_WIN_PAIR_SYNC && GS_WIN_WAVES_LEAVE_ALONE
        // No barrier: a wave that has its apron goes on (inside the steps it waits for the waves above and below it only).  A
        // wave that gave up says that its rows will never come and ends; the others compute on with what they have -- the
        // launch is void -- until their own next exchange finds the abort word.  The exchange planes stay safe without the
        // barrier.  Let wave w store exchange s + 2 over its granules of exchange s, and let y (of another window) be a reader
        // of one of them, c.  c lies within K cells of a cell c* that y's window owns, held by y itself or (aprons deeper than a
        // band: a window's first and last band own nothing) by the band next to y; and c* lies in the apron of w's window, in
        // a band at most two away from w's (K <= 8 rows, bands of 5).  w has finished super-step s + 2, so the bands one and
        // two away from it are inside that super-step (its last step needed the neighbours' last step, theirs the step
        // before from the bands next to them): the reader of c* has passed exchange s + 1, so c* had been stored for
        // exchange s + 1 -- by a band that had finished super-step s + 1, whose last step needed y in that step: y had passed
        // exchange s, i.e. had read c.
        if (failed) {
            if (lane == 0) ((__attribute__((address_space(3))) volatile int *)(lds + win_rows_floats()))[wave] = 0x7fffffff;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_endpgm(); // (the wave ends here: no way out of the loop for the compiler to keep values for)
        }
#else
        __syncthreads(); // (the waves agree: all go on or all leave)
        if (!*go) return;
#endif
        GS_WIN_TRACE_AT(5);
        GS_WIN_TRACE_AT(6);
#else
        // ---- exchange s: ring out, flag, poll, apron in -------------------------------------------------------
        const __amdgpu_buffer_rsrc_t xu = win_rsrc(x.xu[s & 1]), xv = win_rsrc(x.xv[s & 1]);
        const bool lane_owned = wc >= K && wc < K + OW && gc < a.cols;
        const bool lane_ring = wc < 2 * K || wc >= OW;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int wr = wave * RPW + r; // wave-uniform
            const bool row_owned = wr >= K && wr < K + OH && gr + r < a.rows;
            const bool row_ring = wr < 2 * K || wr >= OH;
            if (row_owned && lane_owned && (row_ring || lane_ring)) {
                const int off = ((gr + r) * a.pitch + gc) * (int)sizeof(float);
                const v2f su = {u[r][0], u[r][1]}, sv = {v[r][0], v[r][1]};
                __builtin_amdgcn_raw_buffer_store_b64(su, xu, off, 0, SC1);
                __builtin_amdgcn_raw_buffer_store_b64(sv, xv, off, 0, SC1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GS_WIN_TRACE_AT(2);
        __syncthreads();
        GS_WIN_TRACE_AT(3);
        if (wave == 0) {
            const int target = x.epoch + s + 1;
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(target, win_rsrc(x.flags), wg * 4, 0, SC1);
            // one lane per workgroup whose cells this window's apron covers: one vector load polls them all
            const bool watch = lane < d->n_nbr;
            const int theirs = watch ? d->nbr[lane] : 0;
            int ok = 1, spins = 0;
            for (;;) {
                const int seen = watch ? __builtin_amdgcn_raw_buffer_load_b32(win_rsrc(x.flags), theirs * 4, 0, SC1) : target;
                if (!__builtin_amdgcn_ballot_w64(seen - target < 0)) break;
                if (++spins > x.patience || __builtin_amdgcn_raw_buffer_load_b32(win_rsrc(x.abort), 0, 0, SC1) != 0) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) {
                // (the number of THIS launch: the launches before it ran to their end, what the host needs to know --
                // a launch that finds the word set leaves at once, so only workgroups of one launch ever write it)
                if (!ok) __builtin_amdgcn_raw_buffer_store_b32(x.seq, win_rsrc(x.abort), 0, 0, SC1);
                *go = ok;
            }
        }
        GS_WIN_TRACE_AT(4);
        __syncthreads();
        if (!*go) return; // (workgroup-uniform)
        GS_WIN_TRACE_AT(5);
        const bool lane_in = gc >= 0 && gc < a.cols;
        const bool lane_apron = (wc < K || wc >= K + OW) && wc < 2 * K + OW && lane_in;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int wr = wave * RPW + r;
            const bool row_in = gr + r >= 0 && gr + r < a.rows;
            const bool row_apron = (wr < K || wr >= K + OH) && wr < 2 * K + OH;
            if (row_in && ((row_apron && lane_in && wc < 2 * K + OW) || (lane_apron && wr < 2 * K + OH))) {
                const int off = ((gr + r) * a.pitch + gc) * (int)sizeof(float);
                const v2f fu = __builtin_amdgcn_raw_buffer_load_b64(xu, off, 0, SC1);
                const v2f fv = __builtin_amdgcn_raw_buffer_load_b64(xv, off, 0, SC1);
                const bool in1 = gc + 1 < a.cols;
                u[r][0] = fu[0]; u[r][1] = in1 ? fu[1] : 0.0f;
                v[r][0] = fv[0]; v[r][1] = in1 ? fv[1] : 0.0f;
            }
        }
#if defined(GS_WIN_TRACE)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        GS_WIN_TRACE_AT(6);
#endif
    }
#if defined(GS_WIN_BUDGET)
    GS_WIN_BUDGET_ADD(2, __builtin_amdgcn_s_memtime() - run_t0);
#endif
    // the cells this workgroup owns, where they lie in the grid (8-byte stores; a second column beyond `cols` lands in
    // the planes' padding columns, which nothing reads)
    if (wc >= K && wc < K + OW && gc < a.cols) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int wr = wave * RPW + r;
            if (wr >= K && wr < K + OH && gr + r < a.rows) {
                const int off = ((gr + r) * a.pitch + gc) * (int)sizeof(float);
                const v2f su = {u[r][0], u[r][1]}, sv = {v[r][0], v[r][1]};
                __builtin_amdgcn_raw_buffer_store_b64(su, win_rsrc(a.out_u), off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(sv, win_rsrc(a.out_v), off, 0, 0);
            }
        }
    }
}

template <int RPW, int FAST>
__global__ __launch_bounds__(kWinWaves * 64) void GS_SUFFIX(gs_run_window_k)(GsStepArgs a, GsWindowArgs x)
{
    if ((FAST & 1) && !GS_MATH_FUSED) __builtin_amdgcn_s_setreg(1 | (9 << 6), 0); // half_diff: MODE.IEEE = 0
    extern __shared__ float lds[];
    __shared__ int go;
    constexpr int SC1 = 16;
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int K = x.k;
    const int wg = (int)blockIdx.x;
    const GsWindowDesc *d = x.desc + wg;                    // (uniform: scalar loads)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = d->active, OH = d->oh, OW = d->ow;        // window rows in use; owned rows and columns
    const int gr0 = d->r0 - K, gc0 = d->c0 - K;             // global coordinates of window cell (0, 0)
    const int gr = gr0 + wave * RPW, gc = gc0 + 2 * lane;   // this lane's first cell
    // A launch enqueued behind one that gave up leaves at once (nothing of it is valid anyway).  ONE wave reads the
    // word for the whole workgroup: waves that read it for themselves could disagree (a workgroup of this launch may
    // give up at any time) and a barrier below would wait for waves that have left.
    if (wave == 0 && lane == 0) go = __builtin_amdgcn_raw_buffer_load_b32(win_rsrc(x.abort), 0, 0, SC1) == 0;
#if GS_WIN_PAIR_SYNC
    // (a wave beyond the window's rows in use has published its zeros for every step to come)
    if (lane == 0) ((__attribute__((address_space(3))) volatile int *)(lds + win_rows_floats()))[wave] = wave * RPW >= d->active ? 0x7fffffff : 0;
#endif
    __syncthreads();
    if (!go) return; // (workgroup-uniform)
    // elements 0 and 65 of both arrays of this wave's published rows (window columns -2, -1, 128, 129) are never written by
    // a step; they are read into cells that are discarded, and zeroed once so that nothing depends on earlier contents of
    // the LDS
    if (lane < 4)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            lds[(((b >> 1) * kWinWaves + wave) * 2 + (b & 1)) * kWinPitch + (lane & 1) * kWinHalf + (lane >> 1) * (kWinHalf - 1)] = 0.0f;
    if (wave * RPW >= H) {
        // A wave beyond the window's rows in use publishes zeros once (the last wave in use reads them as its row below)
        // and (builds with barriers inside the run) then only keeps the workgroup's barrier count.
#pragma unroll
        for (int b = 0; b < 8; ++b)
            { float *p = lds + (((b >> 1) * kWinWaves + wave) * 2 + (b & 1)) * kWinPitch + 1 + lane; p[0] = 0.0f; p[kWinHalf] = 0.0f; }
#if GS_WIN_PAIR_SYNC
        __syncthreads(); // (the zeros are there before anybody reads them: no barrier inside the steps)
#endif
#if GS_WIN_PAIR_SYNC && GS_WIN_TAGGED && GS_WIN_WAVES_LEAVE_ALONE
        return; // (no barrier left to keep count of)
#endif
        const int supers = (x.steps + K - 1) / K;
        for (int s = 0; s < supers; ++s) {
#if !GS_WIN_PAIR_SYNC
            const int n = (s == 0 && x.steps % K) ? x.steps % K : K;
            for (int i = 0; i < n; ++i) __syncthreads();
#endif
            if (s == supers - 1) break;
#if !GS_WIN_TAGGED
            __syncthreads();
#endif
            __syncthreads();
            if (!go) return;
        }
        return;
    }
#if GS_WIN_PAIR_SYNC
    __syncthreads();
#endif
    float u[RPW][2], v[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const bool in = gr + r >= 0 && gr + r < a.rows && gc >= 0 && gc < a.cols && wave * RPW + r < H;
        v2f fu = {0.0f, 0.0f}, fv = {0.0f, 0.0f};
        if (in) { // 8-byte loads: gc is even and the row pitch a multiple of 64 floats
            const int off = ((gr + r) * a.pitch + gc) * (int)sizeof(float);
            fu = __builtin_amdgcn_raw_buffer_load_b64(win_rsrc(a.in_u), off, 0, 0);
            fv = __builtin_amdgcn_raw_buffer_load_b64(win_rsrc(a.in_v), off, 0, 0);
        }
        const bool in1 = in && gc + 1 < a.cols;
        u[r][0] = fu[0]; u[r][1] = in1 ? fu[1] : 0.0f;
        v[r][0] = fv[0]; v[r][1] = in1 ? fv[1] : 0.0f;
    }
    const bool left = gc0 <= 0, right = gc0 + 2 * K + OW >= a.cols, ends = gr0 <= 0 || gr0 + H >= a.rows;
    const bool edge = left || right || ends;
    constexpr bool KINDS = (FAST & 1) && !GS_MATH_FUSED;
#define GS_WIN_RUN(E, Z) window_run<RPW, E, FAST, Z>(a, x, d, OH, OW, lds, &go, wg, gr, gc, wave, lane, u, v)
    // One branch per workgroup, one instantiation per kind of window (as gs_step_tb_k): the cheap kinds exist for the
    // clipped rule with the default side weights in the strict build; a grid narrower than one window, general
    // weights and the fused build take the general path in their edge windows.
    const bool cheap = KINDS && a.edge_kinds;
    if (!edge) GS_WIN_RUN(0, -1);
    else if (a.zero_halo) GS_WIN_RUN(7, 1);
    else if (cheap && left && !right && !ends) GS_WIN_RUN(KINDS ? 2 : 1, 0);
    else if (cheap && right && !left && !ends) GS_WIN_RUN(KINDS ? 3 : 1, 0);
    else if (cheap && ends && !left && !right) GS_WIN_RUN(KINDS ? 4 : 1, 0);
    else if (cheap && left && !right) GS_WIN_RUN(KINDS ? 5 : 1, 0);
    else if (cheap && right && !left) GS_WIN_RUN(KINDS ? 6 : 1, 0);
    else GS_WIN_RUN(1, 0);
#undef GS_WIN_RUN
}

} // namespace
