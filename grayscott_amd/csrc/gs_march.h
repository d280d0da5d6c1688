// gs_march.h -- the production kernel of gs_run: temporal blocking, K <= 4 time steps per launch, K register-resident
// time levels per wave; the variant with full difference sharing and the halo board in LDS memory; 4- and 16-wave forms.
// Part of the gfx950 step kernels: included by gs_step_kernels.hip (which sets GS_MATH_FUSED / GS_TB_OP_ONLY and the
// GS_SUFFIX / GS_TAP macros) inside one translation unit per arithmetic flavour; not a header to include elsewhere.
#pragma once

namespace {

// ------------------------------------------------------------------------------------
// Temporal blocking: K time steps per launch (one HBM read + one HBM write per K steps).
//
// Same per-cell arithmetic, so the results are bit-identical to K single-step launches.
// A wave loads a 256-column window [248*s - 4, 248*s + 252) of its strip and marches down
// the rows with a software pipeline of K time levels, each keeping a 3-row window in
// registers: per "tick" it takes one new level-0 row from the prefetch queue, computes one
// row of level 1 from the level-0 window, one row of level 2 from the level-1 window, ...
// and stores one row of level K.  Lanes 0 and 63 are sacrificial: their outermost columns
// lose one column of validity per level (no neighbour to read), so after K <= 4 levels
// lanes 1..62 (248 columns) still hold exact values.  No halo loads, no LDS, no barriers;
// redundant work is 8/256 of the columns plus 2K rows per unit.
//   HBM traffic per launch ~ 16 B per cell (+ ~3 % column overlap, + 2K/rows_per_unit rows),
//   algorithmic traffic 16 B * K per cell: the kernel moves from HBM-bound (K = 1, 2)
//   towards VALU-bound (K = 4).
// ------------------------------------------------------------------------------------
// Columns per lane (CPL).  The wide layout above (4 columns per lane, 16-B accesses) is the one
// for large grids.  Small grids do not have enough 248-column strips x row units to fill 256 CUs,
// so the same march also exists with 2 and 1 columns per lane: 2x / 4x more waves for the same
// unit height.  A sacrificial lane of CPL columns absorbs CPL levels, so ceil(K / CPL) lanes per
// side are sacrificial and a wave produces (64 - 2 * ceil(K / CPL)) * CPL output columns.
__host__ __device__ constexpr int tb_sacrificial_lanes(int k, int cpl) { return (k + cpl - 1) / cpl; }
__host__ __device__ constexpr int tb_cols_per_wave(int k, int cpl) { return (64 - 2 * tb_sacrificial_lanes(k, cpl)) * cpl; }
static_assert(tb_cols_per_wave(4, 4) == 248 && tb_cols_per_wave(4, 1) == 56 && tb_cols_per_wave(3, 2) == 120, "");

// Measured on MI355X while tuning this kernel (tools/ubench/valu_rate.hip, sweeps under
// profiles/): packed v_pk_{add,mul}_f32 have the same lane throughput as scalar VALU ops
// (so (u,v)-pair arithmetic buys nothing), DPP moves cost ~1.5 scalar ops (so neighbour
// columns are fetched once per row and kept, not re-read at each use), v_cndmask is ~8x a
// scalar op (kept out of the interior path), and 2 or 4 waves per SIMD issue at full rate
// while 3 do not.  The kernel is VALU-issue bound for K >= 3.
template <int CPL>
struct RowQ { // a level-0 row as fetched (no halo columns: sacrificial lanes instead)
    float u[CPL], v[CPL];
};
template <int CPL>
struct RowT { // [0] = column c-1, [1..CPL] = own columns, [CPL+1] = column c+CPL
    float u[CPL + 2], v[CPL + 2];
};

// Neighbour-lane reads of the temporally blocked kernel, whose outermost lanes are sacrificial (they
// may receive anything).  Measured on MI355X (tools/ubench/valu_rate2.hip, profiles/archive/r02_sweeps.md): a
// DPP instruction issues at half the VALU rate and, mixed into ordinary VALU code, costs the wave 3-5
// issue slots; ds_bpermute_b32 goes through the LDS crossbar (no LDS memory, ~6 cycles per CU and
// wave-instruction) and takes no VALU slot at all.  At 4 exchanges per row and level the crossbar is
// ~40 % busy, so the exchange is free: +7 % at 16384^2 over the DPP form (GS_TB_XLANE=0, kept for A/B).
#if GS_TB_XLANE
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// lane i receives lane i-1's `own` (lane 0: lane 63's)
__device__ __forceinline__ float shift_from_prev_lane(float own)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane_id() - 1) & 63) << 2, __builtin_bit_cast(int, own)));
}
// lane i receives lane i+1's `own` (lane 63: lane 0's)
__device__ __forceinline__ float shift_from_next_lane(float own)
{
    // the previous lane's address + 8: the add folds into the instruction's offset field (one address
    // register for both directions), and the crossbar takes the lane index modulo 64
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((((lane_id() - 1) & 63) << 2) + 8, __builtin_bit_cast(int, own)));
}
#else
// DPP wave shifts with bound_ctrl (0 for the lane without a source), no `old` operand.
__device__ __forceinline__ float shift_from_prev_lane(float own)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shift_from_next_lane(float own)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x130, 0xf, 0xf, true));
}
#endif

template <int CPL>
__device__ __forceinline__ RowT<CPL> widen_tb(const float (&u)[CPL], const float (&v)[CPL])
{
    RowT<CPL> w;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { w.u[1 + i] = u[i]; w.v[1 + i] = v[i]; }
    w.u[0] = shift_from_prev_lane(u[CPL - 1]);
    w.u[CPL + 1] = shift_from_next_lane(u[0]);
    w.v[0] = shift_from_prev_lane(v[CPL - 1]);
    w.v[CPL + 1] = shift_from_next_lane(v[0]);
    return w;
}

// The CPL interior cells a lane computes in one row.  With the default side weights in the strict build (FAST & 1)
// two cells side by side share a difference: IEEE subtraction is antisymmetric, half_diff(a, b) == -half_diff(b, a)
// bit for bit except that a zero comes out as +0 on both sides, and the accumulator, never -0, does not tell +0
// from -0 (see half_diff).  So the right-hand tap of a cell is kept and SUBTRACTED as the left-hand tap of the next
// cell: one instruction less per pair of neighbours, the reference's order of additions unchanged.
template <int FAST, int CPL, int ZH>
__device__ __forceinline__ void cells_interior(const GsStepArgs &a, const RowT<CPL> &m, const RowT<CPL> &z, const RowT<CPL> &p,
                                               float (&nu)[CPL], float (&nv)[CPL])
{
    if constexpr (GS_TB_HSHARE && CPL > 1 && (FAST & 1) && !GS_MATH_FUSED) {
        float hu = 0.0f, hv = 0.0f; // the previous cell's right-hand tap
#pragma unroll
        for (int k = 1; k <= CPL; ++k) {
            const float u = z.u[k], v = z.v[k];
            float acc_u = 0.0f, acc_v = 0.0f;
            GS_TAP(acc_u, a.w[0][0], m.u[k - 1], u); GS_TAP(acc_v, a.w[0][0], m.v[k - 1], v);
            GS_TAP_HALF(acc_u, m.u[k], u);           GS_TAP_HALF(acc_v, m.v[k], v);
            GS_TAP(acc_u, a.w[0][2], m.u[k + 1], u); GS_TAP(acc_v, a.w[0][2], m.v[k + 1], v);
            if (k == 1) {
                GS_TAP_HALF(acc_u, z.u[k - 1], u);   GS_TAP_HALF(acc_v, z.v[k - 1], v);
            } else {
                acc_u = acc_u - hu;                  acc_v = acc_v - hv;
            }
            hu = half_diff(z.u[k + 1], u);           hv = half_diff(z.v[k + 1], v);
            acc_u = acc_u + hu;                      acc_v = acc_v + hv;
            GS_TAP(acc_u, a.w[2][0], p.u[k - 1], u); GS_TAP(acc_v, a.w[2][0], p.v[k - 1], v);
            GS_TAP_HALF(acc_u, p.u[k], u);           GS_TAP_HALF(acc_v, p.v[k], v);
            GS_TAP(acc_u, a.w[2][2], p.u[k + 1], u); GS_TAP(acc_v, a.w[2][2], p.v[k + 1], v);
            react<(FAST & 2) != 0>(a, u, v, acc_u, acc_v, nu[k - 1], nv[k - 1]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPL; ++k) cell<0, FAST, RowT<CPL>, ZH>(a, m, z, p, k + 1, true, true, 0u, 0u, nu[k], nv[k]);
    }
}

// Full difference sharing (FAST & 4: side weights 0.5 AND w[0][0] == w[2][2], w[0][2] == w[2][0]; strict build).
// The three taps a cell takes from the row BELOW it are, negated, the three taps the cells of that row take from the
// row above them:  S(r, c) = (x[r+1][c] - x[r][c]) / 2 = -N(r+1, c),  SE(r, c) = w22 (x[r+1][c+1] - x[r][c]) =
// -NW(r+1, c+1) when w00 == w22,  SW(r, c) = w20 (x[r+1][c-1] - x[r][c]) = -NE(r+1, c-1) when w02 == w20 -- bit for
// bit: IEEE subtraction and multiplication are odd functions of their operands (round-to-nearest is symmetric), up to
// the sign of an exact or flushed zero, which an accumulator that is never -0 cannot see (the argument of half_diff
// and of the E / W pair in cells_interior).  So a march keeps, per level, the S / SE / SW taps of the row it has just
// finished (TapCarry) and the next row's N / NW / NE taps are one subtraction each, in the reference's order:
//   acc = 0 - SE'(k-1);  acc -= S'(k);  acc -= SW'(k+1);  acc -= E(k-1);  acc += E(k);  acc += SW(k);  acc += S(k);  acc += SE(k)
// ("0.0f -" is kept like the fold's "0.0f +": it maps a -0 to +0.)  The row above is no longer needed at all: a level
// is two widened rows and a carry instead of three rows.  Per lane-row and species 14 CPL + 5 arithmetic instructions
// instead of 20 CPL - (CPL - 1): 46 per cell-step instead of 52 at 2 columns per lane (the SE tap of the column left of
// the lane's and the SW tap of the column right of it are computed in-lane from the widened rows, no extra exchange).
template <int CPL>
struct TapCarry { // of the row a level has just finished, [i] for i = 0 .. CPL - 1: what own cell i + 1 of the next row needs
    float s_u[CPL], s_v[CPL];   // S tap of cell i + 1  (-> N tap of cell i + 1)
    float se_u[CPL], se_v[CPL]; // SE tap of cell i     (-> NW tap of cell i + 1; cell 0 = the column left of the lane's)
    float sw_u[CPL], sw_v[CPL]; // SW tap of cell i + 2 (-> NE tap of cell i + 1; cell CPL + 1 = the column right of the lane's)
};
// One row of a level: z = the row, p = the row below it, c = the taps carried from the row above, replaced in place by
// this row's (every old value is read before the new one of its slot exists, but for SE, which waits one cell).
template <int FAST, int CPL>
__device__ __forceinline__ void cells_vshare(const GsStepArgs &a, const RowT<CPL> &z, const RowT<CPL> &p, TapCarry<CPL> &c,
                                             float (&nu)[CPL], float (&nv)[CPL])
{
    static_assert((FAST & 5) == 5 && !GS_MATH_FUSED, "full difference sharing is a specialisation of the strict build");
    float eu = half_diff(z.u[1], z.u[0]), ev = half_diff(z.v[1], z.v[0]);                       // E tap of cell 0
    float seu = a.w[2][2] * (p.u[1] - z.u[0]), sev = a.w[2][2] * (p.v[1] - z.v[0]);             // SE tap of cell 0
#pragma unroll
    for (int k = 1; k <= CPL; ++k) {
        const float u = z.u[k], v = z.v[k];
        float acc_u = zero_minus(c.se_u[k - 1]), acc_v = zero_minus(c.se_v[k - 1]);             // NW
        c.se_u[k - 1] = seu;                                  c.se_v[k - 1] = sev;
        acc_u = acc_u - c.s_u[k - 1];                         acc_v = acc_v - c.s_v[k - 1];     // N
        acc_u = acc_u - c.sw_u[k - 1];                        acc_v = acc_v - c.sw_v[k - 1];    // NE
        acc_u = acc_u - eu;                                   acc_v = acc_v - ev;               // W
        eu = half_diff(z.u[k + 1], u);                        ev = half_diff(z.v[k + 1], v);
        acc_u = acc_u + eu;                                   acc_v = acc_v + ev;               // E
        const float swu = a.w[2][0] * (p.u[k - 1] - u), swv = a.w[2][0] * (p.v[k - 1] - v);
        if (k >= 2) { c.sw_u[k - 2] = swu; c.sw_v[k - 2] = swv; }
        acc_u = acc_u + swu;                                  acc_v = acc_v + swv;              // SW
        c.s_u[k - 1] = half_diff(p.u[k], u);                  c.s_v[k - 1] = half_diff(p.v[k], v);
        acc_u = acc_u + c.s_u[k - 1];                         acc_v = acc_v + c.s_v[k - 1];     // S
        seu = a.w[2][2] * (p.u[k + 1] - u);                   sev = a.w[2][2] * (p.v[k + 1] - v);
        acc_u = acc_u + seu;                                  acc_v = acc_v + sev;              // SE
        react<(FAST & 2) != 0>(a, u, v, acc_u, acc_v, nu[k - 1], nv[k - 1]);
    }
    c.sw_u[CPL - 1] = a.w[2][0] * (p.u[CPL] - z.u[CPL + 1]);  c.sw_v[CPL - 1] = a.w[2][0] * (p.v[CPL] - z.v[CPL + 1]); // SW tap of cell CPL + 1
}

// ... and sharing ACROSS lanes (FAST & 8).  Three differences cross the boundary between two lanes in every row -- the side
// tap between the columns left and right of it and the two diagonals -- and in cells_vshare both lanes compute all three
// from each other's halo columns: 10 of its 92 arithmetic instructions per lane-row.  Here the lane on the RIGHT of a
// boundary computes them, from its left neighbour's second column (zl, pl: the only halo column left, half the board and
// half its traffic), and the lane on the left takes them as DPP operands (wave_shl:1) of the three accumulations they
// belong to: -W of the neighbour's cell 1 is the E tap of its own cell 2, the neighbour's SE tap of its left column the
// SE tap, and the neighbour's SW tap of the tick before, negated, the NE tap.  82 arithmetic instructions per lane-row,
// 41 per cell-step, 6 of them with a DPP operand -- which stalls the SIMD for about three issue slots and costs no joules:
// 3-5 % faster wherever the power cap sets the clock, within 1 % elsewhere (profiles/r05_cross_lane.md).
//
// acc -/+ the RIGHT neighbour's x in one instruction (lane 63, sacrificial, takes 0).  Volatile: never moved into the
// branch around the stores, where the sacrificial lanes -- whose taps their neighbours need -- are off.  s_nop 1: the two
// wait states a DPP read needs behind the VALU write of its register, whichever instruction wrote it (the compiler does
// not look into half_diff's asm; without the s_nop the same build runs 4-6 % SLOWER).
__device__ __forceinline__ float minus_next(float acc, float x)
{
    float r;
    asm volatile("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ float plus_next(float acc, float x)
{
    float r;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
// One row of a level, 2 columns per lane: z = the row (own columns), p = the row below it, zl / pl = (U, V) of the left
// neighbour's second column in those rows.  The carry's slots here: s = S of cells 1, 2; se = SE of the column left of the
// lane's and of cell 1; sw[0] = SW of cell 1 (the LEFT NEIGHBOUR reads it, a tick later), sw[1] = SW of cell 2 (own use).
template <int FAST>
__device__ __forceinline__ void cells_xshare(const GsStepArgs &a, const RowQ<2> &z, const RowQ<2> &p, float zl_u, float zl_v,
                                             float pl_u, float pl_v, TapCarry<2> &c, float (&nu)[2], float (&nv)[2])
{
    static_assert((FAST & 13) == 13 && !GS_MATH_FUSED, "a specialisation of the strict build");
    // the three differences across the left boundary
    const float w1u = half_diff(zl_u, z.u[0]), w1v = half_diff(zl_v, z.v[0]);                     // W of cell 1
    const float se0u = a.w[2][2] * (p.u[0] - zl_u), se0v = a.w[2][2] * (p.v[0] - zl_v);           // SE of the column left of it
    const float sw1u = a.w[2][0] * (pl_u - z.u[0]), sw1v = a.w[2][0] * (pl_v - z.v[0]);           // SW of cell 1
    // cell 1
    const float u = z.u[0], v = z.v[0];
    float acc_u = zero_minus(c.se_u[0]), acc_v = zero_minus(c.se_v[0]);                           // NW
    c.se_u[0] = se0u;                                 c.se_v[0] = se0v;
    acc_u = acc_u - c.s_u[0];                         acc_v = acc_v - c.s_v[0];                   // N
    acc_u = acc_u - c.sw_u[1];                        acc_v = acc_v - c.sw_v[1];                  // NE
    acc_u = acc_u + w1u;                              acc_v = acc_v + w1v;                        // W
    const float eu = half_diff(z.u[1], u), ev = half_diff(z.v[1], v);
    acc_u = acc_u + eu;                               acc_v = acc_v + ev;                         // E
    acc_u = acc_u + sw1u;                             acc_v = acc_v + sw1v;                       // SW
    c.s_u[0] = half_diff(p.u[0], u);                  c.s_v[0] = half_diff(p.v[0], v);
    acc_u = acc_u + c.s_u[0];                         acc_v = acc_v + c.s_v[0];                   // S
    const float seu = a.w[2][2] * (p.u[1] - u), sev = a.w[2][2] * (p.v[1] - v);
    acc_u = acc_u + seu;                              acc_v = acc_v + sev;                        // SE
    react<(FAST & 2) != 0>(a, u, v, acc_u, acc_v, nu[0], nv[0]);
    // cell 2
    const float u2 = z.u[1], v2 = z.v[1];
    acc_u = zero_minus(c.se_u[1]);                    acc_v = zero_minus(c.se_v[1]);              // NW
    c.se_u[1] = seu;                                  c.se_v[1] = sev;
    acc_u = acc_u - c.s_u[1];                         acc_v = acc_v - c.s_v[1];                   // N
    acc_u = minus_next(acc_u, c.sw_u[0]);             acc_v = minus_next(acc_v, c.sw_v[0]);       // NE
    c.sw_u[0] = sw1u;                                 c.sw_v[0] = sw1v;
    acc_u = acc_u - eu;                               acc_v = acc_v - ev;                         // W
    acc_u = minus_next(acc_u, w1u);                   acc_v = minus_next(acc_v, w1v);             // E
    c.sw_u[1] = a.w[2][0] * (p.u[0] - u2);            c.sw_v[1] = a.w[2][0] * (p.v[0] - v2);
    acc_u = acc_u + c.sw_u[1];                        acc_v = acc_v + c.sw_v[1];                  // SW
    c.s_u[1] = half_diff(p.u[1], u2);                 c.s_v[1] = half_diff(p.v[1], v2);
    acc_u = acc_u + c.s_u[1];                         acc_v = acc_v + c.s_v[1];                   // S
    acc_u = plus_next(acc_u, se0u);                   acc_v = plus_next(acc_v, se0v);             // SE
    react<(FAST & 2) != 0>(a, u2, v2, acc_u, acc_v, nu[1], nv[1]);
}

// Buffer-instruction forms of the plane accesses: address = 128-bit resource in SGPRs (base pointer of
// the unit's first row) + per-lane byte offset (one VGPR for the whole march) + scalar byte offset of
// the row: no 64-bit per-lane addresses to keep or to recompute per row.  The resource is raw (stride 0)
// with the widest record count: the units never step outside their planes, so nothing relies on the
// range check.  With them, the late fetch (GS_TB_LATE_FETCH, gs_experiments.h) and the edge path's column
// masks kept as lane masks in SGPRs, the whole kernel entry -- general path included -- fits 126
// registers: 4 waves per SIMD instead of 3, +9 % at 16384^2 (profiles/archive/r02_sweeps.md, section 8).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
}
template <int CPL>
__device__ __forceinline__ void load_cols_buf(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&out)[CPL])
{
    if constexpr (CPL == 4) {
        const auto x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, GS_TB_AUX_LOAD);
        __builtin_memcpy(out, &x, sizeof x);
    } else if constexpr (CPL == 2) {
        const auto x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, GS_TB_AUX_LOAD);
        __builtin_memcpy(out, &x, sizeof x);
    } else {
        const auto x = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, GS_TB_AUX_LOAD);
        __builtin_memcpy(out, &x, sizeof x);
    }
}
template <int CPL>
__device__ __forceinline__ void store_cols_buf(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&in)[CPL])
{
    if constexpr (CPL == 4) {
        decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) x;
        __builtin_memcpy(&x, in, sizeof x);
        __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, GS_TB_AUX_STORE);
    } else if constexpr (CPL == 2) {
        decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0)) x;
        __builtin_memcpy(&x, in, sizeof x);
        __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, GS_TB_AUX_STORE);
    } else {
        decltype(__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0)) x;
        __builtin_memcpy(&x, in, sizeof x);
        __builtin_amdgcn_raw_buffer_store_b32(x, r, voff, soff, GS_TB_AUX_STORE);
    }
}

// Fair progress in launches of about one round of wave slots (FAIR, 16-wave workgroups).  The SIMD's issue
// arbitration is priority, then AGE: of four waves with equal work the two oldest take nearly every slot, and
// the four finish one after the other -- the last one alone on its SIMD for 15-20 % of the launch, where a
// lone wave issues at most every 4th cycle, half the SIMD's rate (tools/wave_timeline.py, profiles/archive/r03_sweeps.md
// section 1).  With all 16 waves of a CU in one workgroup every wave publishes its progress (256ths of its
// ticks) in an LDS word per tick and reads the words of the waves that share its SIMD: whoever is ahead of
// another runs at priority 0, the others at 3, so the four stay within a tick of each other and end together.
// Nobody ever waits for anybody: the board only steers s_setprio.
struct FairBoard {
    int *progress; // LDS: words 0..15 progress per wave (0 ... 256, INT_MAX once finished), 16..31 the SIMD it runs on
    int simd;      // this wave's SIMD (HW_REG_HW_ID bits 5:4)
    int wave;
    float *halo;   // LDS: this wave's halo board (tb_halo_floats floats) in the variants with full difference sharing
};

// Halo board of the march with full difference sharing (2 columns per lane).  The columns next to a lane's two are
// needed in TWO consecutive ticks there (a row is the lower row of one tick's differences and the upper row of the
// next's); kept in registers they cost 4 per level on top of the carried taps, and the march needs 148: three waves per
// SIMD, which issue at 0.80-0.85 of the rate of four (profiles/r05_energy.md).  So a wave hands its rows' columns
// to its neighbouring lanes through LDS MEMORY instead of the crossbar: every new row is written once (two
// ds_write2_b32) and its halo columns are read in the tick it appears and again in the next (one ds_read2_b32 per side
// and tick): the same load on the LDS pipe as four ds_bpermute_b32 and 16 registers less.  Layout per level and slot
// (tick & 1): four arrays of 66 floats -- U and V of the lanes' first and of their second column, element 1 + lane --
// so that the left halo (second column of lane - 1) and the right halo (first column of lane + 1) are conflict-free
// 4-byte accesses; elements 0 and 65 are only read by the sacrificial lanes (zeroed once).
constexpr int kHaloArray = 66, kHaloRow = 4 * kHaloArray;
// ... and of the march that also shares the differences ACROSS lanes (FAST & 8, tb_march): only the left neighbour's
// second column is needed; per level and slot 65 (U, V) pairs, element 1 + lane, element 0 for lane 0's reads.
constexpr int kCrossLanes = 65;
__host__ __device__ constexpr int tb_halo_floats(int k, bool cross) { return cross ? k * 2 * 2 * kCrossLanes : k * 2 * kHaloRow; }

// EDGE: 0 = interior unit; 1 = general path; 2 / 3 = strip on the grid's left / right edge that touches neither its
// top nor its bottom (cell<2> / cell<3>); 4 = interior strip that touches the top or bottom edge: interior code for
// every row but the grid's first / last, which take the general cell (wave-uniform branch per level-row).
template <int K, int EDGE, int FAST, int CPL, int ZH = -1, bool FAIR = false>
__device__ __forceinline__ void tb_march(const GsStepArgs &a, int ur0, int ur1, int strip, int lane,
                                         const FairBoard &fb GS_TRACE_PARAM)
{
    constexpr int S = tb_sacrificial_lanes(K, CPL), W = tb_cols_per_wave(K, CPL);
    const int c = strip * W + (lane - S) * CPL; // first column of this lane (may be negative)
    constexpr bool COLS = EDGE == 1 || EDGE == 2 || EDGE == 3; // the strip may leave the grid's columns
    constexpr bool ROWS = EDGE == 1 || EDGE == 4;              // the unit may touch the grid's first / last row
    const bool load_ok = !COLS || (c >= 0 && c < a.pitch);
    const bool store_ok = (lane >= S) && (lane < 64 - S) && (!COLS || c < a.pitch);
    const ptrdiff_t pitch = a.pitch;

    // Level-0 rows needed: [ur0 - K, ur1 + K) clipped to the rows that exist: the slab's own
    // rows plus, on a slab seam, `ghost` rows of the neighbouring slab.
    const int row_lo = max(ur0 - K, a.top_present ? -a.ghost : 0);
    const int row_hi = min(ur1 + K - 1, a.bottom_present ? a.rows + a.ghost - 1 : a.rows - 1);
    constexpr bool LATE = GS_TB_LATE_FETCH && K == 4 && CPL == 2;
    // resources based at the unit's first input row (row_lo) / first output row (ur0): scalar row offsets
    // stay small and positive whatever the size of the plane
    const __amdgpu_buffer_rsrc_t ru = plane_rsrc(a.in_u + (ptrdiff_t)row_lo * pitch), rv = plane_rsrc(a.in_v + (ptrdiff_t)row_lo * pitch);
    const __amdgpu_buffer_rsrc_t wu = plane_rsrc(a.out_u + (ptrdiff_t)ur0 * pitch), wv = plane_rsrc(a.out_v + (ptrdiff_t)ur0 * pitch);
    const int voff = c * (int)sizeof(float), pitch_bytes = a.pitch * (int)sizeof(float);
    auto fetch = [&](int row) {
        RowQ<CPL> r;
        const int rr = min(max(row, row_lo), row_hi);
        if (load_ok) {
            load_cols_buf<CPL>(ru, voff, (rr - row_lo) * pitch_bytes, r.u);
            load_cols_buf<CPL>(rv, voff, (rr - row_lo) * pitch_bytes, r.v);
        } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) { r.u[i] = 0.f; r.v[i] = 0.f; }
        }
        return r;
    };

    // per-lane masks (all ones = that neighbour column is clipped away).  c is a multiple of CPL,
    // so only the first of a lane's cells can sit on the global left edge.
    uint32_t la[CPL], ra[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        // Plain comparisons: the compiler keeps them as lane masks in SGPR pairs and selects with
        // v_cndmask_b32.  Round 1 kept opaque all-ones / all-zeros words in VGPRs and blended bitwise
        // (v_cndmask is ~10x a plain VALU op on gfx950), which made the edge units 0.5 % of a pass
        // faster -- and cost the 3 registers that kept the whole kernel at 3 waves per SIMD.
        la[k] = ((EDGE == 1 || EDGE == 2) && k == 0 && c == 0) ? 0xffffffffu : 0u;
        ra[k] = ((EDGE == 1 || EDGE == 3) && (c + k + 1 >= a.cols)) ? 0xffffffffu : 0u;
    }

    RowQ<CPL> q[3];    // prefetch queue of level-0 rows, 3 ticks deep
    const int first = ur0 - K; // level-0 row of tick 0
    const int nticks = (ur1 - ur0) + 2 * K;
    const int fair_scale = FAIR ? (256 << 16) / nticks : 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = fetch(first + i);
    // The in-step form's progress board (FairBoard): publish this wave's progress, steer its priority.
    auto fair_tick = [&](int tick) {
        if constexpr (FAIR) {
            const int mine = tick * fair_scale >> 16;       // 0 ... 256
            if (lane == 0) fb.progress[fb.wave] = mine;
            if (mine >= a.fair_from) { // (before: free-running, out of phase as the arbitration leaves them)
                const int theirs = fb.progress[lane & 15], their_simd = fb.progress[16 + (lane & 15)];
                const unsigned long long behind = __builtin_amdgcn_ballot_w64(their_simd == fb.simd && theirs < mine);
                if (behind) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(3);
            }
        }
    };

    // Full difference sharing (cells_vshare) on interior units of the variant built for it.  A level keeps its two
    // newest rows (slot = tick & 1) and the taps carried from the row before them, updated in place, instead of a
    // window of three rows.  Level j's row l0 - j is computed in the tick in which row l0 - j + 1 of level j - 1
    // appears, from tick 2 j - 1 on -- one row more at the top than the three-row form computes: the first row a level
    // needs takes its N / NW / NE taps from the tick before it; what that extra row itself comes to is never used
    // and never stored -- and every level runs until the last tick.  So the first 2 K ticks are peeled with the levels
    // in use known at compile time, and the loop behind them (6 ticks per trip: the row slots' 2 x the queue's 3) has no
    // test but "ticks left": every slot index is static, nothing is copied from register to register.
    constexpr bool VS = EDGE == 0 && (FAST & 5) == 5 && !GS_MATH_FUSED && CPL == 2;
    if constexpr (VS) {
        constexpr bool XS = (FAST & 8) != 0; // ... and across lanes (cells_xshare): only the left neighbour's second column is handed over
        RowQ<CPL> R[K][2]; // own columns of the two newest rows of level j
        TapCarry<CPL> C[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int e = 0; e < CPL; ++e) { R[j][sl].u[e] = 0.f; R[j][sl].v[e] = 0.f; }
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                C[j].s_u[e] = 0.f; C[j].s_v[e] = 0.f; C[j].se_u[e] = 0.f; C[j].se_v[e] = 0.f; C[j].sw_u[e] = 0.f; C[j].sw_v[e] = 0.f;
            }
        }
        typedef float f2 __attribute__((ext_vector_type(2)));
        float *const mine = fb.halo + 1 + lane; // this lane's element of the first array of (level 0, slot 0)
        f2 *const mine2 = reinterpret_cast<f2 *>(fb.halo) + 1 + lane; // XS: its (U, V) pair; the left neighbour's is mine2[-1]
        // the pads, read by the sacrificial lanes only: elements 0 and 65 of every array / element 0 of every row of pairs
        if constexpr (XS) {
            if (lane == 0)
#pragma unroll
                for (int i = 0; i < 2 * K; ++i) mine2[i * kCrossLanes - 1] = f2{0.f, 0.f};
        } else {
            if (lane < 2)
#pragma unroll
                for (int i = 0; i < K * 2 * 4; ++i) fb.halo[i * kHaloArray + lane * (kHaloArray - 1)] = 0.0f;
        }
        // a new row of level j: its columns go to the board, for the neighbouring lanes
        auto put = [&](int j, int slot, const RowQ<CPL> &r) {
#if GS_VS_ABLATE_HALO
            return; // (timing experiment: no halo traffic at all; results are wrong)
#endif
            if constexpr (XS) {
                mine2[(j * 2 + slot) * kCrossLanes] = f2{r.u[1], r.v[1]};
            } else {
                float *b = mine + (j * 2 + slot) * kHaloRow;
                b[0] = r.u[0]; b[kHaloArray] = r.v[0]; b[2 * kHaloArray] = r.u[1]; b[3 * kHaloArray] = r.v[1];
            }
            // The elements a lane reads back are written by its NEIGHBOURS, in the same instructions: to the
            // compiler, which sees one lane, they are unrelated to the lane's own stores and could be read first.  The
            // LDS executes a wave's instructions in order; the fence pair keeps the compiler from moving the reads up.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // the row of (level j, slot) with the columns next to this lane's
        auto widened = [&](int j, int slot) {
            const float *b = mine + (j * 2 + slot) * kHaloRow;
            RowT<CPL> w;
#if GS_VS_ABLATE_HALO
            w.u[0] = R[j][slot].u[1]; w.v[0] = R[j][slot].v[1]; w.u[3] = R[j][slot].u[0]; w.v[3] = R[j][slot].v[0];
            (void)b;
#else
            w.u[0] = b[2 * kHaloArray - 1]; w.v[0] = b[3 * kHaloArray - 1]; // second column of lane - 1
            w.u[3] = b[1];                  w.v[3] = b[kHaloArray + 1];     // first column of lane + 1
#endif
            w.u[1] = R[j][slot].u[0]; w.u[2] = R[j][slot].u[1]; w.v[1] = R[j][slot].v[0]; w.v[2] = R[j][slot].v[1];
            return w;
        };
        // one tick; `par` = tick & 1, `qs` = tick % 3 and `levels` (levels 1 .. `levels` run) are constants after unrolling
        auto vs_tick = [&](int tick, int par, int qs, int levels, bool store) {
            const int l0 = first + tick;
            GS_TRACE_AT(tick == 3, 1);
            GS_TRACE_AT(tick == 2 * K, 2);
            GS_TRACE_AT(tick == nticks - 2 * K, 3);
            fair_tick(tick);
            R[0][par] = q[qs];
            put(0, par, R[0][par]);
            if constexpr (!LATE) q[qs] = fetch(l0 + 3);
#pragma unroll
            for (int j = 1; j <= K; ++j) {
                if (j > levels) break;
                float nu[CPL], nv[CPL];
                if constexpr (XS) {
                    const f2 zl = mine2[((j - 1) * 2 + (par ^ 1)) * kCrossLanes - 1], pl = mine2[((j - 1) * 2 + par) * kCrossLanes - 1];
                    cells_xshare<FAST>(a, R[j - 1][par ^ 1], R[j - 1][par], zl.x, zl.y, pl.x, pl.y, C[j - 1], nu, nv);
                } else {
                    const RowT<CPL> z = widened(j - 1, par ^ 1), p = widened(j - 1, par);
                    cells_vshare<FAST, CPL>(a, z, p, C[j - 1], nu, nv);
                }
                if (j < K) {
#pragma unroll
                    for (int e = 0; e < CPL; ++e) { R[j][par].u[e] = nu[e]; R[j][par].v[e] = nv[e]; }
                    put(j, par, R[j][par]);
                } else if (store && store_ok) {
                    store_cols_buf<CPL>(wu, voff, (l0 - K - ur0) * pitch_bytes, nu);
                    store_cols_buf<CPL>(wv, voff, (l0 - K - ur0) * pitch_bytes, nv);
                }
            }
            if constexpr (LATE) q[qs] = fetch(l0 + 3); // two rows in flight while the levels are computed
        };
#pragma unroll
        for (int tick = 0; tick < 2 * K; ++tick) vs_tick(tick, tick & 1, tick % 3, (tick + 1) / 2, false); // level j from tick 2 j - 1
        // (whole trips without a test inside: a tick that may be skipped is a block of its own, and every value carried
        // from tick to tick -- 16 per level -- then meets its successor in a register copy at the block's end)
        int t = 2 * K;
        for (; t + 6 <= nticks; t += 6) {
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) vs_tick(t + s6, s6 & 1, (2 * K + s6) % 3, K, true);
        }
#pragma unroll
        for (int s6 = 0; s6 < 5; ++s6)
            if (t + s6 < nticks) vs_tick(t + s6, s6 & 1, (2 * K + s6) % 3, K, true);
        return;
    }

    RowT<CPL> w[K][3]; // w[j][slot]: level-j rows, newest in slot (tick % 3)
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int e = 0; e < CPL + 2; ++e) { w[j][sl].u[e] = 0.f; w[j][sl].v[e] = 0.f; }

    for (int t = 0; t < nticks; t += 3) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int tick = t + s3;
            if (tick < nticks) {
                const int l0 = first + tick; // level-0 row entering the pipeline
                GS_TRACE_AT(tick == 3, 1);
                GS_TRACE_AT(tick == 2 * K, 2);
                GS_TRACE_AT(tick == nticks - 2 * K, 3);
                fair_tick(tick);
                w[0][s3] = widen_tb<CPL>(q[s3].u, q[s3].v);
                if constexpr (!LATE) q[s3] = fetch(l0 + 3);
#pragma unroll
                for (int j = 1; j <= K; ++j) {
                    const int row = l0 - j; // level-j row produced in this tick
                    // needed for this unit's outputs, and a row of the global grid?
                    const bool need = (row >= ur0 - (K - j)) && (row < ur1 + (K - j)) &&
                                      (!ROWS || ((row >= 0 || a.top_present) && (row < a.rows || a.bottom_present)));
                    if (need) {
                        const RowT<CPL> &m = w[j - 1][(s3 + 1) % 3]; // row - 1
                        const RowT<CPL> &z = w[j - 1][(s3 + 2) % 3]; // row
                        const RowT<CPL> &p = w[j - 1][s3];           // row + 1
                        const bool mrow = !ROWS || (row > 0) || a.top_present;
                        const bool prow = !ROWS || (row + 1 < a.rows) || a.bottom_present;
                        float nu[CPL], nv[CPL];
                        if constexpr (EDGE == 4) {
                            if (mrow && prow) {
                                cells_interior<FAST, CPL, ZH>(a, m, z, p, nu, nv);
                            } else {
#pragma unroll
                                for (int k = 0; k < CPL; ++k) cell<1, FAST, RowT<CPL>, ZH>(a, m, z, p, k + 1, mrow, prow, 0u, 0u, nu[k], nv[k]);
                            }
                        } else if constexpr (EDGE == 0) {
                            cells_interior<FAST, CPL, ZH>(a, m, z, p, nu, nv);
                        } else {
#pragma unroll
                            for (int k = 0; k < CPL; ++k)
                                cell<EDGE, FAST, RowT<CPL>, ZH>(a, m, z, p, k + 1, mrow, prow, la[k], ra[k], nu[k], nv[k]);
                        }
                        if (j < K) {
                            w[j][s3] = widen_tb<CPL>(nu, nv);
                        } else if (store_ok) {
                            store_cols_buf<CPL>(wu, voff, (row - ur0) * pitch_bytes, nu);
                            store_cols_buf<CPL>(wv, voff, (row - ur0) * pitch_bytes, nv);
                        }
                    }
                }
                if constexpr (LATE) q[s3] = fetch(l0 + 3); // two rows in flight while the levels are computed
            }
        }
    }
}

// WG: waves per workgroup.  4 independent waves, or all 16 of a CU with the progress board of tb_march<FAIR>.
template <int K, int FAST, int CPL, int WG>
__device__ __forceinline__ void tb_unit(const GsStepArgs &a)
{
    // half_diff needs MODE.IEEE = 0: hwreg(HW_REG_MODE, offset 9, width 1).  The bit only governs
    // the quieting of signalling NaNs otherwise, which parity does not cover.
    if ((FAST & 1) && !GS_MATH_FUSED) __builtin_amdgcn_s_setreg(1 | (9 << 6), 0);
    constexpr int W = tb_cols_per_wave(K, CPL), S = tb_sacrificial_lanes(K, CPL);
    constexpr bool FAIR = WG == 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, see above
    FairBoard fb{nullptr, 0, wave, nullptr};
    if constexpr ((FAST & 5) == 5 && !GS_MATH_FUSED && CPL == 2) { // the variants with full difference sharing: halo boards
        constexpr int n = tb_halo_floats(K, (FAST & 8) != 0);
        __shared__ __attribute__((aligned(16))) float halo_boards[WG * n];
        fb.halo = halo_boards + wave * n;
    }
    if constexpr (FAIR) {
        // The board.  No barrier: a wave starts marching as soon as it is dispatched (a barrier here held every
        // wave until the 16th of its workgroup had arrived: -2 ... -7 % on a whole pass).  What a wave reads of a
        // peer that has not started yet is whatever the previous workgroup left in LDS -- a wrong guess at a
        // priority for a few ticks, never at a result.
        __shared__ int board[32];
        fb.simd = (int)(__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)));
        if (lane == 0) { board[wave] = 0; board[16 + wave] = fb.simd; }
        fb.progress = board;
    }
    // a wave without a unit marks itself finished (never "behind") and leaves
#define GS_TB_LEAVE do { if constexpr (FAIR) { if (lane == 0) fb.progress[wave] = 0x7fffffff; } return; } while (0)
    const int strips = (a.cols + W - 1) / W;
    // Units in dispatch order (edge units first).  4-wave workgroups take four consecutive ones; the dispatcher
    // deals the workgroups over the CUs.  A 16-wave workgroup of the 1-column layout takes every gridDim.x-th unit
    // instead: with consecutive units the first 31 workgroups would hold nothing but edge units, whose half-height
    // general-path marches are longer than the interior's when units are short (13 ticks x 1.6 against 18 at
    // 10-row units: those CUs end 15 % late; 390 k -> 419 k).  With 2 columns per lane the edge halves are the
    // shorter ones (27 x 1.57 against 46 ticks at 38 rows) and 16 neighbouring strips on one CU read 1 % faster.
    int block = (int)blockIdx.x;
    if (a.xcd_m > 0 && block >= a.xcd_first) {
        const int per = 8 * a.xcd_m, r = block - a.xcd_first, g = r / per, o = r - g * per;
        if ((g + 1) * per <= (int)gridDim.x - a.xcd_first) block = a.xcd_first + g * per + (o & 7) * a.xcd_m + (o >> 3);
    }
    const int unit = FAIR && CPL == 1 ? wave * (int)gridDim.x + (int)blockIdx.x : block * WG + wave;
    const int rpu = a.rows_per_unit;
    const int small = a.small_rpu;
    const int rest_a = a.ra1 - a.ra0 - a.big_chunks * rpu; // rows of range a behind the full-height chunks
    const int chunks_a = a.mid_chunks < 0 ? a.big_chunks + (rest_a + small - 1) / small
                                          : a.big_chunks + a.mid_chunks + (rest_a - a.mid_chunks * small + a.tiny_rpu - 1) / a.tiny_rpu;
    const int chunks_b = (a.rb1 - a.rb0 + rpu - 1) / rpu;
    const int chunks = chunks_a + chunks_b;
    // Dispatch order.  Units on a global edge take the general path, which is 1.6x as slow
    // (per-lane selects); a slow unit that starts in the last round of a launch stretches
    // its tail, so all edge units go first: the left-most and right-most strips of every chunk,
    // then (below) the last `bot_first` and the first chunks of range a -- the ones a grid edge can
    // touch -- then everything else.  A strip is a right-edge strip when its window, sacrificial
    // lanes included, reaches the last column.
    const int er = ((strips - 1) * W + S * CPL >= a.cols && strips >= 2) ? 2 : 1; // edge strips on the right
    const int ne = 1 + er;                                                        // ... per chunk
    // With edge_split = 2 the edge units come as two half-height units each (half = 0 / 1): they are the
    // outer strips of every chunk and every strip of the first edge_chunks chunks in dispatch order.
    const int es = a.edge_split == 2 ? 2 : 1;
    int chunk, strip, half = -1;
    if (strips <= ne) {
        if (unit >= chunks * strips * es) GS_TB_LEAVE; // wave-uniform
        chunk = unit / (strips * es);
        const int rem = unit - chunk * strips * es;
        strip = rem / es;
        if (es == 2) half = rem - strip * es;
    } else if (unit < chunks * ne * es) {
        chunk = unit / (ne * es);
        const int rem = unit - chunk * ne * es, se = rem / es;
        if (es == 2) half = rem - se * es;
        strip = se == 0 ? 0 : strips - er + (se - 1);
    } else {
        const int ni = strips - ne, nec = es == 2 ? min(a.edge_chunks, chunks) : 0;
        int v = unit - chunks * ne * es;
        if (v < nec * ni * es) {
            chunk = v / (ni * es);
            const int rem = v - chunk * ni * es, si = rem / es;
            half = rem - si * es;
            strip = 1 + si;
        } else {
            v -= nec * ni * es;
            chunk = nec + v / ni;
            if (chunk >= chunks) GS_TB_LEAVE; // wave-uniform
            strip = 1 + (v - (v / ni) * ni);
        }
    }
    int ur0, ur1;
    if (chunk < chunks_a) {
        // the last chunks of the range first, then chunks 0, 1, 2, ... (bottom / top edge chunks)
        const int bf = min(a.bot_first, chunks_a);
        const int cc = chunk < bf ? chunks_a - 1 - chunk : chunk - bf;
        if (cc < a.big_chunks) {
            ur0 = a.ra0 + cc * rpu;
            ur1 = ur0 + rpu;
        } else if (a.mid_chunks < 0 || cc < a.big_chunks + a.mid_chunks) { // tapered tail: short units are dispatched last
            ur0 = a.ra0 + a.big_chunks * rpu + (cc - a.big_chunks) * small;
            ur1 = min(ur0 + small, a.ra1);
        } else { // ... and the shortest ones at the very end
            ur0 = a.ra0 + a.big_chunks * rpu + a.mid_chunks * small + (cc - a.big_chunks - a.mid_chunks) * a.tiny_rpu;
            ur1 = min(ur0 + a.tiny_rpu, a.ra1);
        }
    } else {
        ur0 = a.rb0 + (chunk - chunks_a) * rpu;
        ur1 = min(ur0 + rpu, a.rb1);
    }
    if (half >= 0) { // this unit is one half of its chunk's rows
        const int hh = (ur1 - ur0 + 1) >> 1;
        if (half == 0) ur1 = min(ur0 + hh, ur1);
        else ur0 = ur0 + hh;
        if (ur0 >= ur1) GS_TB_LEAVE; // a one-row chunk has no second half (wave-uniform)
    }
    const bool left = strip == 0, right = (strip + 1) * W + S * CPL >= a.cols;
    const bool ends = (ur0 - K < 0 && !a.top_present) || (ur1 + K > a.rows && !a.bottom_present);
    const bool edge = left || right || ends;
#if defined(GS_TB_TRACE)
    unsigned long long ts[5] = {trace_now(), 0, 0, 0, 0};
    const unsigned long long cycles0 = __builtin_readcyclecounter(); // s_memtime: the shader clock's counter
#endif
    // One branch per unit (all of it wave-uniform), one instantiation per kind of unit and boundary rule: with a
    // run-time test inside the cell the compiler hoists the other kinds' selects above the branch.  The cheap edge
    // kinds (cell<2>, cell<3>, EDGE = 4) exist for the clipped rule with the default side weights in the strict
    // build; every other combination -- corners, a grid narrower than two strips, the zero-halo rule, general
    // weights -- takes the general path.  GsStepArgs::edge_kinds = 0 sends every edge unit there (A/B timing).
    constexpr bool KINDS = (FAST & 1) && !GS_MATH_FUSED;
    if (!edge)
        tb_march<K, 0, FAST, CPL, -1, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    else if (a.zero_halo)
        tb_march<K, 1, FAST, CPL, 1, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    else if (KINDS && a.edge_kinds && left && !right && !ends)
        tb_march<K, KINDS ? 2 : 1, FAST, CPL, 0, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    else if (KINDS && a.edge_kinds && right && !left && !ends)
        tb_march<K, KINDS ? 3 : 1, FAST, CPL, 0, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    else if (KINDS && a.edge_kinds && ends && !left && !right)
        tb_march<K, KINDS ? 4 : 1, FAST, CPL, 0, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    else
        tb_march<K, 1, FAST, CPL, 0, FAIR>(a, ur0, ur1, strip, lane, fb GS_TRACE_ARG);
    if constexpr (FAIR) { if (lane == 0) fb.progress[wave] = 0x7fffffff; }
#undef GS_TB_LEAVE
#if defined(GS_TB_TRACE)
    ts[4] = trace_now();
    const unsigned long long cycles = __builtin_readcyclecounter() - cycles0;
    if (lane == 0 && unit < kTraceUnits) {
        unsigned long long *rec = gs_trace_buf + (size_t)unit * kTraceWords;
        for (int i = 0; i < 5; ++i) rec[i] = ts[i];
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
        rec[5] = ((unsigned long long)xcc << 32) | hw;
        rec[6] = ((unsigned long long)(unsigned)ur0 << 32) | (unsigned)((ur1 - ur0) | (edge ? 0x40000000 : 0));
        // shader cycles between entry and exit (in-kernel clock = cycles / (ts[4] - ts[0]) x 100 MHz) | strip
        rec[7] = (cycles << 32) | (unsigned)strip;
    }
#endif
}

template <int K, int FAST, int CPL, int WG = 4>
__global__ __launch_bounds__(WG * 64) void GS_SUFFIX(gs_step_tb_k)(GsStepArgs a)
{
    tb_unit<K, FAST, CPL, WG>(a);
}
// The variants with full difference sharing (FAST = 7, and 15 = also across lanes; 2 columns per lane) are kernels of
// their own: built for four waves per SIMD (the register allocator is told so; left to itself it settles a few
// registers above 128).
template <int K, int WG = 4>
__global__ __launch_bounds__(WG * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void GS_SUFFIX(gs_step_tb_ds_k)(GsStepArgs a)
{
    tb_unit<K, 7, 2, WG>(a);
}
template <int K, int WG = 4>
__global__ __launch_bounds__(WG * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void GS_SUFFIX(gs_step_tb_dx_k)(GsStepArgs a)
{
    tb_unit<K, 15, 2, WG>(a);
}

} // namespace
