// gs_lds_resident.h -- small and mid-size grids: the whole run in one launch with the grid resident in LDS
// (gs_run_resident_k), and up to 8 steps per launch on LDS-resident windows (gs_run_tile_k).
// Part of the gfx950 step kernels: included by gs_step_kernels.hip (which sets GS_MATH_FUSED / GS_TB_OP_ONLY and the
// GS_SUFFIX / GS_TAP macros) inside one translation unit per arithmetic flavour; not a header to include elsewhere.
#pragma once

namespace {

// ------------------------------------------------------------------------------------
// Small grids: the whole run in ONE launch, the grid resident in LDS.
//
// A grid of up to kResidentCells cells is loaded once by one 1024-thread workgroup, advanced
// `steps` times LDS -> LDS with a barrier per step, and stored once.  For such grids a pass of the
// kernels above is a dependent launch of a few microseconds per 1-4 steps and nothing else -- the
// reference's criterion grid starts at 8 x 16 cells; here a step is one sweep of 4 waves per SIMD
// over LDS.  Per-cell code = the general (edge) flavour of cell(): every thread builds its 3 x 3
// window from LDS with clamped indices and passes per-thread presence flags / masks.
// ------------------------------------------------------------------------------------
constexpr int kResidentCells = kGsResidentCells;
constexpr int kResidentThreads = 1024;

struct Row3 { float u[3], v[3]; }; // [0] = column c-1, [1] = c, [2] = c+1

// One cell on or near the grid's border, clipped-window rule, with the eight neighbour weights
// of THIS cell in E (row-major, centre left out): the reference indexes its weight table from the top-left
// corner of the clipped window, so a cell without a row above / a column to its left uses the table shifted
// by one row / column, and a neighbour outside the grid has no tap at all -- weight 0 here, which adds
// +-0 to an accumulator that starts at +0: the same bits as no tap, as long as the neighbour's VALUE is
// finite (cells outside the grid are kept at 0).  Every lane runs the same straight-line code: no selects.
template <int FAST>
__device__ __forceinline__ void cell_border(const GsStepArgs &a, const float (&E)[8], const Row3 &m, const Row3 &z, const Row3 &p,
                                            float &out_u, float &out_v)
{
    const float u = z.u[1], v = z.v[1];
    float acc_u = 0.0f, acc_v = 0.0f;
    GS_TAP(acc_u, E[0], m.u[0], u); GS_TAP(acc_v, E[0], m.v[0], v);
    GS_TAP(acc_u, E[1], m.u[1], u); GS_TAP(acc_v, E[1], m.v[1], v);
    GS_TAP(acc_u, E[2], m.u[2], u); GS_TAP(acc_v, E[2], m.v[2], v);
    GS_TAP(acc_u, E[3], z.u[0], u); GS_TAP(acc_v, E[3], z.v[0], v);
    GS_TAP(acc_u, E[4], z.u[2], u); GS_TAP(acc_v, E[4], z.v[2], v);
    GS_TAP(acc_u, E[5], p.u[0], u); GS_TAP(acc_v, E[5], p.v[0], v);
    GS_TAP(acc_u, E[6], p.u[1], u); GS_TAP(acc_v, E[6], p.v[1], v);
    GS_TAP(acc_u, E[7], p.u[2], u); GS_TAP(acc_v, E[7], p.v[2], v);
    react<(FAST & 2) != 0>(a, u, v, acc_u, acc_v, out_u, out_v);
}

// The eight weights of the cell at (r, c) for the clipped-window rule (see cell_border): the table shifted by
// one row for a cell of row 0 and by one column for a cell of column 0, 0 for a neighbour outside the grid.
__device__ __forceinline__ void border_weights(const GsStepArgs &a, int r, int c, float (&E)[8])
{
    const int rs = r == 0 ? 1 : 0;
    const bool cs = c == 0, left = c - 1 >= 0 && c - 1 < a.cols, right = c + 1 >= 0 && c + 1 < a.cols;
    int t = 0;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
        const bool row_present = r + dr >= 0 && r + dr < a.rows;
        const int ri = dr + 1 - rs < 0 ? 0 : dr + 1 - rs; // (-1 only for a row that does not exist)
        float wrow[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            wrow[j] = !row_present ? 0.0f : (ri == 0 ? a.w[0][j] : (ri == 1 ? a.w[1][j] : a.w[2][j]));
        E[t++] = left ? wrow[0] : 0.0f; // (no left neighbour at column 0)
        if (dr != 0) E[t++] = cs ? wrow[0] : wrow[1];
        E[t++] = right ? (cs ? wrow[1] : wrow[2]) : 0.0f;
    }
}

// The grid lives in LDS with a ring of zeros around it (pitch cols + 2, rows + 2 rows; two buffers per
// species): every neighbour is addressable at a fixed offset and a neighbour outside the grid reads 0.
// That IS the zero-halo rule (interior code for every cell, ZH = 1); for the clipped-window rule every cell
// carries its own eight weights (cell_border).  No selects, no divergent branches in the step loop.
template <int FAST, int ZH>
__global__ __launch_bounds__(kResidentThreads) void GS_SUFFIX(gs_run_resident_k)(GsStepArgs a, int steps, int to_out)
{
    if ((FAST & 1) && !GS_MATH_FUSED) __builtin_amdgcn_s_setreg(1 | (9 << 6), 0); // half_diff: MODE.IEEE = 0
    extern __shared__ float lds[];
    const int cells = a.rows * a.cols, cols = a.cols, P = cols + 2, plane = (a.rows + 2) * P;
    // planes in LDS: U buffer 0, U buffer 1, V buffer 0, V buffer 1 -- addressed by offset (a select between
    // pointers would make the compiler lose the address space and emit flat_load)
    constexpr int CPT = (kResidentCells + kResidentThreads - 1) / kResidentThreads; // cells per thread, at most
    const int nthreads = (int)blockDim.x; // as many waves as hold cells, at most kResidentThreads (the launcher)
    for (int i = threadIdx.x; i < 4 * plane; i += nthreads) lds[i] = 0.0f;               // the rings (and everything else)
    __syncthreads();
    int o[CPT], g[CPT];
    bool live[CPT];
    float E[CPT][8];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int idx = (int)threadIdx.x + k * nthreads;
        live[k] = idx < cells;
        const int r = live[k] ? idx / cols : 0, c = live[k] ? idx - r * cols : 0;
        o[k] = (r + 1) * P + c + 1;
        g[k] = r * a.pitch + c;
        if (ZH == 0) border_weights(a, r, c, E[k]);
        if (live[k]) {
            lds[o[k]] = a.in_u[g[k]];
            lds[2 * plane + o[k]] = a.in_v[g[k]];
        }
    }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        const float *su = lds + cur * plane, *sv = lds + (2 + cur) * plane;
        float *du = lds + (cur ^ 1) * plane, *dv = lds + (2 + (cur ^ 1)) * plane;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (!live[k]) continue;
            Row3 R[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = o[k] + (i - 1) * P;
                R[i].u[0] = su[q - 1]; R[i].u[1] = su[q]; R[i].u[2] = su[q + 1];
                R[i].v[0] = sv[q - 1]; R[i].v[1] = sv[q]; R[i].v[2] = sv[q + 1];
            }
            float nu, nv;
            if (ZH == 0)
                cell_border<FAST>(a, E[k], R[0], R[1], R[2], nu, nv);
            else
                cell<false, FAST, Row3>(a, R[0], R[1], R[2], 1, true, true, 0u, 0u, nu, nv);
            du[o[k]] = nu;
            dv[o[k]] = nv;
        }
        __syncthreads();
        cur ^= 1;
    }
    float *gu = to_out ? a.out_u : const_cast<float *>(a.in_u);
    float *gv = to_out ? a.out_v : const_cast<float *>(a.in_v);
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (live[k]) {
            gu[g[k]] = lds[cur * plane + o[k]];
            gv[g[k]] = lds[(2 + cur) * plane + o[k]];
        }
}

// ------------------------------------------------------------------------------------
// Mid-size grids: K <= 8 time steps per launch on LDS-resident windows, one cell per lane and row.
//
// Between the single-workgroup resident kernel (<= 1536 cells) and grids that fill the chip with
// marching waves (~1 M cells and up), a pass of gs_step_tb_k is bound by the LENGTH of a wave's march
// (unit height + 2K ticks of K levels, one wave per SIMD issuing every 4th cycle) plus a dependent
// launch per K <= 4 steps: 2.3-3.4 us per step whatever the grid (profiles/archive/r02_criterion_grid.md).
// Here a workgroup of 16 waves owns a window of 16 * RPW rows x 64 columns: wave w holds rows
// w * RPW ... in registers, one column per lane.  Per step every wave publishes its rows in LDS
// (double-buffered: one workgroup barrier per step), reads the rows above and below its own and the
// left / right neighbours of its own cells back (ds_read_b32 at constant offsets from one address), and
// updates its cells through the same cell<> code as every other kernel: bit-identical.  Nothing is
// exchanged with other workgroups: the ring of cells whose neighbours lie outside the window loses its
// validity, one ring per step, so after K steps the window shrunk by K cells on every side is exact
// and is what the workgroup stores (windows overlap by 2K).  Where a window leaves the grid, the cells
// outside are zeros and stay zeros: that is the zero-halo rule as it stands, and for the clipped-window rule
// every cell of such a window carries its own eight weights (cell_border: the table shifted as the
// reference's corner-anchored indexing shifts it, 0 for a neighbour that does not exist).
// The first form of this kernel (4-cell strips, 2-8 waves per tile; profiles/archive/r02_sweeps.md, section 4)
// spent 3.3-5.5 us per step on a 16 x 40 window: a wave alone on its SIMD issues one instruction per 4
// cycles and a strip was a chain of ~250 of them.  With 16 waves per window every SIMD has 4 waves to
// issue from and a step is ~55 * RPW instructions per wave.
// ------------------------------------------------------------------------------------
constexpr int kTileMaxK = kGsTileMaxSteps;
constexpr int kTileCols = 64;                 // window columns = lanes
constexpr int kTilePitch = kTileCols + 2;     // + window columns -1 and 64 (never valid, only addressable; zeroed)
constexpr int kTileWaves = 16;                // 1024 threads
__host__ __device__ constexpr int tile_rows(int rpw) { return kTileWaves * rpw; }
// 2 buffers x 2 species x (rows + the rows above and below the window) x pitch
__host__ __device__ constexpr size_t tile_lds_bytes(int rpw) { return (size_t)4 * (tile_rows(rpw) + 2) * kTilePitch * sizeof(float); }

// K steps of a window.  EDGE: the window touches the grid's border.  Its cells outside the grid are zeros
// and stay zeros; with the zero-halo rule (ZH = 1) that IS the rule and every cell runs the interior code;
// with the clipped rule (ZH = 0) every cell runs cell_border with its own weights.  (The general flavour of
// cell<>, per-tap selects, costs 1.57x an interior cell -- and while every workgroup has a CU to itself the
// launch lasts as long as its slowest workgroup, a border window: this form costs 1.04x / 1.19x.)
template <int RPW, bool EDGE, int FAST, int ZH>
__device__ __forceinline__ void tile_steps(const GsStepArgs &a, float *lds, int K, int gr, int gc, int wave, int lane,
                                           float (&u)[RPW], float (&v)[RPW])
{
    constexpr int H = tile_rows(RPW), P = kTilePitch, plane = (H + 2) * P; // plane: one species of one buffer
    // element (buffer b, species s, window row r, window column c) = (2 b + s) * plane + (r + 1) * P + c + 1;
    // `o` = this lane's first cell in species 0 of buffer 0
    const int o = (wave * RPW + 1) * P + lane + 1;
    bool inside[RPW];
    float E[RPW][8];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        inside[i] = !EDGE || (gr + i >= 0 && gr + i < a.rows && gc >= 0 && gc < a.cols);
        if (EDGE && ZH == 0) border_weights(a, gr + i, gc, E[i]); // rows are wave-uniform: scalar selects
    }
    // The ring around the window (rows -1 and H, columns -1 and 64 of all four planes) is only ever read
    // into cells whose values are discarded; it is zeroed once per launch so that nothing -- not even a
    // discarded value -- depends on what an earlier workgroup left in LDS.
    {
        const int ring_row = wave == 0 ? 0 : H + 1; // waves 0 and 15 also own the row above / below the window
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (lane < 2)
#pragma unroll
                for (int i = 0; i < RPW; ++i) lds[b * plane + (wave * RPW + 1 + i) * P + lane * (P - 1)] = 0.0f;
            if (wave == 0 || wave == kTileWaves - 1) {
                lds[b * plane + ring_row * P + lane + 1] = 0.0f;
                if (lane < 2) lds[b * plane + ring_row * P + lane * (P - 1)] = 0.0f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) { lds[o + i * P] = u[i]; lds[plane + o + i * P] = v[i]; }
    __syncthreads();
    int cur = 0;
    for (int s = 1; s <= K; ++s) {
        const float *su = lds + cur * 2 * plane + o, *sv = su + plane;
        Row3 R[RPW + 2]; // R[0] = the row above this wave's rows, R[1 + i] = its row i, R[RPW + 1] = the row below
#pragma unroll
        for (int i = 0; i < RPW + 2; ++i) {
            const int d = (i - 1) * P;
            R[i].u[0] = su[d - 1]; R[i].u[2] = su[d + 1];
            R[i].v[0] = sv[d - 1]; R[i].v[2] = sv[d + 1];
            if (i == 0 || i == RPW + 1) { R[i].u[1] = su[d]; R[i].v[1] = sv[d]; }
            else { R[i].u[1] = u[i - 1]; R[i].v[1] = v[i - 1]; }
        }
        float nu[RPW], nv[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if (EDGE && ZH == 0)
                cell_border<FAST>(a, E[i], R[i], R[i + 1], R[i + 2], nu[i], nv[i]);
            else
                cell<false, FAST, Row3>(a, R[i], R[i + 1], R[i + 2], 1, true, true, 0u, 0u, nu[i], nv[i]);
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) { u[i] = inside[i] ? nu[i] : 0.0f; v[i] = inside[i] ? nv[i] : 0.0f; }
        if (s < K) { // publish for the next step (the other buffer: no wave can still be reading it)
            float *du = lds + (cur ^ 1) * 2 * plane + o;
#pragma unroll
            for (int i = 0; i < RPW; ++i) { du[i * P] = u[i]; du[plane + i * P] = v[i]; }
            __syncthreads();
            cur ^= 1;
        }
    }
}

template <int RPW, int FAST>
__global__ __launch_bounds__(kTileWaves * 64) void GS_SUFFIX(gs_run_tile_k)(GsStepArgs a, int K)
{
    if ((FAST & 1) && !GS_MATH_FUSED) __builtin_amdgcn_s_setreg(1 | (9 << 6), 0); // half_diff: MODE.IEEE = 0
    extern __shared__ float lds[];
    constexpr int H = tile_rows(RPW);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HO = H - 2 * K, WO = kTileCols - 2 * K; // output rows / columns per window
    const int tiles_c = (a.cols + WO - 1) / WO;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int gr0 = tr * HO - K, gc0 = tc * WO - K; // global coordinates of window cell (0, 0)
    const int gr = gr0 + wave * RPW, gc = gc0 + lane; // this lane's first cell
    // load; cells outside the grid are zeros (and stay zeros: tile_steps)
    float u[RPW], v[RPW];
    const int cc = min(max(gc, 0), a.cols - 1);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const ptrdiff_t g = (ptrdiff_t)min(max(gr + i, 0), a.rows - 1) * a.pitch + cc;
        const bool in = gr + i >= 0 && gr + i < a.rows && gc >= 0 && gc < a.cols;
        u[i] = in ? a.in_u[g] : 0.0f;
        v[i] = in ? a.in_v[g] : 0.0f;
    }
    // A window inside the grid runs code without any bounds logic; the others the general flavour, one
    // instantiation per boundary rule (as gs_step_tb_k).
    const bool edge = gr0 <= 0 || gc0 <= 0 || gr0 + H >= a.rows || gc0 + kTileCols >= a.cols;
    if (!edge)
        tile_steps<RPW, false, FAST, -1>(a, lds, K, gr, gc, wave, lane, u, v);
    else if (a.zero_halo)
        tile_steps<RPW, true, FAST, 1>(a, lds, K, gr, gc, wave, lane, u, v);
    else
        tile_steps<RPW, true, FAST, 0>(a, lds, K, gr, gc, wave, lane, u, v);
    // store the window shrunk by K, where it lies in the grid
    if (lane >= K && lane < kTileCols - K && gc < a.cols) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int wr = wave * RPW + i;
            if (wr >= K && wr < H - K && gr + i < a.rows) {
                const ptrdiff_t g = (ptrdiff_t)(gr + i) * a.pitch + gc;
                a.out_u[g] = u[i];
                a.out_v[g] = v[i];
            }
        }
    }
}

} // namespace
