// gs_cell.h -- the per-cell arithmetic every kernel shares: react (the reaction update), half_diff / zero_minus (single
// instructions the compiler must not touch), the general and the cheap edge flavours of cell<>, lane-shift helpers.
// Part of the gfx950 step kernels: included by gs_step_kernels.hip (which sets GS_MATH_FUSED / GS_TB_OP_ONLY and the
// GS_SUFFIX / GS_TAP macros) inside one translation unit per arithmetic flavour; not a header to include elsewhere.
#pragma once

namespace {

// compute/naive/src/lib.rs:74-79, one rounded op per reference op.  DT1: time_step == 1.0f,
// where `du * dt` is the identity on every f32 (NaNs stay NaNs) and is not issued.
template <bool DT1 = false>
__device__ __forceinline__ void react(const GsStepArgs &a, float u, float v, float acc_u,
                                      float acc_v, float &out_u, float &out_v)
{
    const float uv_square = (u * v) * v;
    const float du = (a.du * acc_u - uv_square) + a.feed * (1.0f - u);
    const float dv = (a.dv * acc_v + uv_square) - a.feed_plus_kill * v;
    out_u = DT1 ? u + du : u + du * a.dt;
    out_v = DT1 ? v + dv : v + dv * a.dt;
}

// (s - c) * 0.5f in ONE instruction: v_sub_f32 with the VOP3 output modifier div:2.  The hardware
// applies the modifier to the rounded difference, so the result has the bits of the two-operation
// sequence -- measured on gfx950 over 1.4 M operand pairs including sub-normal, huge and non-finite
// ones (tools/ubench/omod_probe.hip) -- with two provisos, both met by the strict build:
//   * the modifier is ignored unless f32 results are flushed (FP_DENORM: the strict build's mode)
//     and MODE.IEEE is clear (the kernels that use it clear the bit on entry);
//   * a flushed result is +0 where the multiply gives -0.  A tap is only ever ADDED to the
//     accumulator, which starts at +0 and therefore is never -0, and x + (+0) == x + (-0) for every
//     x other than -0: the accumulator's bits are the same.
__device__ __forceinline__ float half_diff(float s, float c)
{
    float r;
    asm("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(r) : "v"(s), "v"(c));
    return r;
}
#define GS_TAP_HALF(acc, s, c) (acc) = (acc) + half_diff((s), (c))

// 0.0f - x as ONE instruction the compiler cannot touch: it folds `(0.0f - x) - y` into `(-x) - y` even without
// fast-math flags, which is -0 instead of +0 for x == y == +0 (the accumulator of cells_vshare must never be -0).
__device__ __forceinline__ float zero_minus(float x)
{
    float r;
    asm("v_sub_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// m ? a : b for a per-lane all-ones / all-zeros mask: one v_bfi_b32, a full-rate VALU op
// (v_cndmask_b32 measured ~8x slower on gfx950: tools/ubench/valu_rate.hip).
__device__ __forceinline__ float blend(uint32_t m, float a, float b)
{
    return __builtin_bit_cast(float, (m & __builtin_bit_cast(uint32_t, a)) | (~m & __builtin_bit_cast(uint32_t, b)));
}

// Map a linear "row slot" onto the two row ranges of GsStepArgs.
__device__ __forceinline__ int range_row(const GsStepArgs &a, int slot)
{
    const int na = a.ra1 - a.ra0;
    return slot < na ? a.ra0 + slot : a.rb0 + (slot - na);
}

// ------------------------------------------------------------------------------------
// Production kernel: register sliding window + DPP halo exchange.
// ------------------------------------------------------------------------------------

// lane i receives lane i-1's `own`; lane 0 keeps `lane0_value`  (DPP wave_shr:1)
__device__ __forceinline__ float from_prev_lane(float own, float lane0_value)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0_value),
                                           __builtin_bit_cast(int, own), 0x138, 0xf, 0xf, false));
}
// lane i receives lane i+1's `own`; lane 63 keeps `lane63_value`  (DPP wave_shl:1)
__device__ __forceinline__ float from_next_lane(float own, float lane63_value)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane63_value),
                                           __builtin_bit_cast(int, own), 0x130, 0xf, 0xf, false));
}

struct RowIn { // one row of this lane's 4 columns as it arrives from memory
    float4 u, v;
    float hu, hv; // halo column: lane 0 holds column c0-1, lane 63 holds column c0+256
};
struct RowW { // the same row widened with the neighbouring lanes' edge columns
    float u[6], v[6]; // [0] = column c-1, [1..4] = own columns, [5] = column c+4
};

struct LaneCtx {
    bool lane_ok;  // this lane's 4 columns lie inside the row pitch
    bool halo_ok;  // this lane fetches a halo column
    int halo_off;  // -1 (lane 0) or +4 (lane 63)
};

template <bool EDGE>
__device__ __forceinline__ RowIn load_row(const float *pu, const float *pv, const LaneCtx &lc)
{
    RowIn r;
    if (!EDGE || lc.lane_ok) {
        r.u = *reinterpret_cast<const float4 *>(pu);
        r.v = *reinterpret_cast<const float4 *>(pv);
    } else {
        r.u = make_float4(0.f, 0.f, 0.f, 0.f);
        r.v = r.u;
    }
    r.hu = 0.f;
    r.hv = 0.f;
    if (lc.halo_ok) {
        r.hu = pu[lc.halo_off];
        r.hv = pv[lc.halo_off];
    }
    return r;
}

__device__ __forceinline__ RowW widen(const RowIn &r)
{
    RowW w;
    w.u[1] = r.u.x; w.u[2] = r.u.y; w.u[3] = r.u.z; w.u[4] = r.u.w;
    w.v[1] = r.v.x; w.v[2] = r.v.y; w.v[3] = r.v.z; w.v[4] = r.v.w;
    w.u[0] = from_prev_lane(r.u.w, r.hu);
    w.u[5] = from_next_lane(r.u.x, r.hu);
    w.v[0] = from_prev_lane(r.v.w, r.hv);
    w.v[5] = from_next_lane(r.v.x, r.hv);
    return w;
}

// One output cell.  k = 1..4 indexes the centre inside RowW.  Interior flavour: all eight
// neighbours exist.  EDGE flavour: `mrow` / `prow` say whether the row above / below exists
// (wave-uniform), `la` / `ra` whether the left / right neighbour column is absent (per lane).
// FAST (strict build, chosen by the host from the parameters; see gs_kernels.h): bit 0 = the four
// side weights are exactly 0.5f (interior cells fold `sub, mul` into half_diff), bit 1 = dt == 1.
// ZH (EDGE flavour): the boundary rule, -1 = read a.zero_halo at run time, 0 = clipped window, 1 = zero
// halo.  The temporally blocked kernel branches on the rule ONCE per unit and instantiates both: with
// a run-time test inside the cell the compiler hoists the other rule's selects above the branch
// (speculative execution) and every edge cell pays for both rules.
// EDGE: 0 = interior; 1 = general (any window clipping, per-tap selects); 2 / 3 = a cell of a strip on the grid's
// LEFT / RIGHT edge whose rows above and below exist, clipped rule, FAST & 1 (side weights 0.5), strict build:
// the reference's fold over the clipped window, whose weight table is anchored at the window's corner, IS the
// interior fold over substituted operands, and a substituted centre value contributes w * (u - u) = +0:
//   right edge (window columns c-1, c): the three right-hand operands := u;
//   left edge  (window columns c, c+1, weights shifted by one column): (left, centre, right) operands :=
//     (column c, column c+1, u) in the rows above and below; in the cell's own row (u, column c+1, u), where the
//     middle one meets the table's centre weight w[1][1] (0 in every stencil of the reference; the tap is
//     issued, so that a non-finite neighbour spreads as it does there).
// 6 (right) or 18 (left, first cell of a lane only) selects on top of the interior's 53 instructions, where the
// general path needs 83: the edge strips -- 8 % of the units of a 4096^2 launch -- cost 1.1-1.2x an interior
// strip instead of 1.57x.
template <int EDGE, int FAST = 0, typename Row = RowW, int ZH = -1>
__device__ __forceinline__ void cell(const GsStepArgs &a, const Row &m, const Row &z,
                                     const Row &p, int k, bool mrow, bool prow, uint32_t la, uint32_t ra,
                                     float &out_u, float &out_v)
{
    const float u = z.u[k], v = z.v[k];
    float acc_u = 0.0f, acc_v = 0.0f;
    if constexpr ((EDGE == 2 || EDGE == 3) && (FAST & 1) && !GS_MATH_FUSED) {
        const bool L = EDGE == 2 && k == 1 && la != 0u; // only the first cell of a lane can sit on column 0
        const bool R = EDGE == 3 && ra != 0u;
        auto pick = [](bool c, float x, float y) { return c ? x : y; };
        const float tlu = EDGE == 2 && k == 1 ? pick(L, m.u[k], m.u[k - 1]) : m.u[k - 1], tlv = EDGE == 2 && k == 1 ? pick(L, m.v[k], m.v[k - 1]) : m.v[k - 1];
        const float tu = EDGE == 2 && k == 1 ? pick(L, m.u[k + 1], m.u[k]) : m.u[k], tv = EDGE == 2 && k == 1 ? pick(L, m.v[k + 1], m.v[k]) : m.v[k];
        const float tru = pick(L || R, u, m.u[k + 1]), trv = pick(L || R, v, m.v[k + 1]);
        const float lu = pick(L, u, z.u[k - 1]), lv = pick(L, v, z.v[k - 1]);
        const float ru = pick(L || R, u, z.u[k + 1]), rv = pick(L || R, v, z.v[k + 1]);
        const float blu = EDGE == 2 && k == 1 ? pick(L, p.u[k], p.u[k - 1]) : p.u[k - 1], blv = EDGE == 2 && k == 1 ? pick(L, p.v[k], p.v[k - 1]) : p.v[k - 1];
        const float bu = EDGE == 2 && k == 1 ? pick(L, p.u[k + 1], p.u[k]) : p.u[k], bv = EDGE == 2 && k == 1 ? pick(L, p.v[k + 1], p.v[k]) : p.v[k];
        const float bru = pick(L || R, u, p.u[k + 1]), brv = pick(L || R, v, p.v[k + 1]);
        GS_TAP(acc_u, a.w[0][0], tlu, u); GS_TAP(acc_v, a.w[0][0], tlv, v);
        GS_TAP_HALF(acc_u, tu, u);        GS_TAP_HALF(acc_v, tv, v);
        GS_TAP(acc_u, a.w[0][2], tru, u); GS_TAP(acc_v, a.w[0][2], trv, v);
        GS_TAP_HALF(acc_u, lu, u);        GS_TAP_HALF(acc_v, lv, v);
        if (EDGE == 2 && k == 1) { // the table's centre weight on column c + 1 (left-edge lane), on the centre (+0) elsewhere
            GS_TAP(acc_u, a.w[1][1], pick(L, z.u[k + 1], u), u); GS_TAP(acc_v, a.w[1][1], pick(L, z.v[k + 1], v), v);
        }
        GS_TAP_HALF(acc_u, ru, u);        GS_TAP_HALF(acc_v, rv, v);
        GS_TAP(acc_u, a.w[2][0], blu, u); GS_TAP(acc_v, a.w[2][0], blv, v);
        GS_TAP_HALF(acc_u, bu, u);        GS_TAP_HALF(acc_v, bv, v);
        GS_TAP(acc_u, a.w[2][2], bru, u); GS_TAP(acc_v, a.w[2][2], brv, v);
    } else if (!EDGE && (FAST & 1) && !GS_MATH_FUSED) {
        GS_TAP(acc_u, a.w[0][0], m.u[k - 1], u); GS_TAP(acc_v, a.w[0][0], m.v[k - 1], v);
        GS_TAP_HALF(acc_u, m.u[k], u);           GS_TAP_HALF(acc_v, m.v[k], v);
        GS_TAP(acc_u, a.w[0][2], m.u[k + 1], u); GS_TAP(acc_v, a.w[0][2], m.v[k + 1], v);
        GS_TAP_HALF(acc_u, z.u[k - 1], u);       GS_TAP_HALF(acc_v, z.v[k - 1], v);
        GS_TAP_HALF(acc_u, z.u[k + 1], u);       GS_TAP_HALF(acc_v, z.v[k + 1], v);
        GS_TAP(acc_u, a.w[2][0], p.u[k - 1], u); GS_TAP(acc_v, a.w[2][0], p.v[k - 1], v);
        GS_TAP_HALF(acc_u, p.u[k], u);           GS_TAP_HALF(acc_v, p.v[k], v);
        GS_TAP(acc_u, a.w[2][2], p.u[k + 1], u); GS_TAP(acc_v, a.w[2][2], p.v[k + 1], v);
    } else if (!EDGE) {
        GS_TAP(acc_u, a.w[0][0], m.u[k - 1], u); GS_TAP(acc_v, a.w[0][0], m.v[k - 1], v);
        GS_TAP(acc_u, a.w[0][1], m.u[k], u);     GS_TAP(acc_v, a.w[0][1], m.v[k], v);
        GS_TAP(acc_u, a.w[0][2], m.u[k + 1], u); GS_TAP(acc_v, a.w[0][2], m.v[k + 1], v);
        GS_TAP(acc_u, a.w[1][0], z.u[k - 1], u); GS_TAP(acc_v, a.w[1][0], z.v[k - 1], v);
        GS_TAP(acc_u, a.w[1][2], z.u[k + 1], u); GS_TAP(acc_v, a.w[1][2], z.v[k + 1], v);
        GS_TAP(acc_u, a.w[2][0], p.u[k - 1], u); GS_TAP(acc_v, a.w[2][0], p.v[k - 1], v);
        GS_TAP(acc_u, a.w[2][1], p.u[k], u);     GS_TAP(acc_v, a.w[2][1], p.v[k], v);
        GS_TAP(acc_u, a.w[2][2], p.u[k + 1], u); GS_TAP(acc_v, a.w[2][2], p.v[k + 1], v);
    } else if (ZH < 0 ? a.zero_halo != 0 : ZH != 0) {
        // GS_BOUNDARY_ZERO_HALO: all nine taps, centred weights; a neighbour outside the grid reads
        // as 0: per-lane column masks, and for an absent row a wave-uniform all-zeros word ANDed in (a
        // `present ? x : 0` select would be a v_cndmask_b32, ~10x a plain VALU op on gfx950).
#define GS_ROW_TAPS_Z(R, WI, PRESENT, WITH_CENTRE)                                             \
    {                                                                                          \
        const uint32_t keep = (PRESENT) ? 0xffffffffu : 0u;                                    \
        const float ul = blend(keep & ~la, R.u[k - 1], 0.0f), vl = blend(keep & ~la, R.v[k - 1], 0.0f); \
        const float ur = blend(keep & ~ra, R.u[k + 1], 0.0f), vr = blend(keep & ~ra, R.v[k + 1], 0.0f); \
        GS_TAP(acc_u, a.w[WI][0], ul, u); GS_TAP(acc_v, a.w[WI][0], vl, v);                    \
        if (WITH_CENTRE) {                                                                     \
            GS_TAP(acc_u, a.w[WI][1], blend(keep, R.u[k], 0.0f), u);                           \
            GS_TAP(acc_v, a.w[WI][1], blend(keep, R.v[k], 0.0f), v);                           \
        }                                                                                      \
        GS_TAP(acc_u, a.w[WI][2], ur, u); GS_TAP(acc_v, a.w[WI][2], vr, v);                    \
    }
        GS_ROW_TAPS_Z(m, 0, mrow, true)
        GS_ROW_TAPS_Z(z, 1, true, false)
        GS_ROW_TAPS_Z(p, 2, prow, true)
#undef GS_ROW_TAPS_Z
    } else {
        // Weight row of the centre row: 1 normally, 0 when the row above is clipped away.
        // Weight column of the centre column: 1 normally, 0 when the left column is clipped.
        // An absent left/right neighbour is replaced by the centre value (adds +0).
        const int zi = mrow ? 1 : 0;
        const float wsel[3][3] = {{a.w[0][0], a.w[0][1], a.w[0][2]},
                                  {a.w[zi][0], a.w[zi][1], a.w[zi][2]},
                                  {a.w[zi + 1][0], a.w[zi + 1][1], a.w[zi + 1][2]}};
#define GS_ROW_TAPS(R, WI, WITH_CENTRE)                                                        \
    {                                                                                          \
        const float wl = wsel[WI][0];                                                          \
        const float wc = blend(la, wsel[WI][0], wsel[WI][1]);                                  \
        const float wr = blend(la, wsel[WI][1], wsel[WI][2]);                                  \
        const float ul = blend(la, u, R.u[k - 1]), vl = blend(la, v, R.v[k - 1]);              \
        const float ur = blend(ra, u, R.u[k + 1]), vr = blend(ra, v, R.v[k + 1]);              \
        GS_TAP(acc_u, wl, ul, u); GS_TAP(acc_v, wl, vl, v);                                    \
        if (WITH_CENTRE) { GS_TAP(acc_u, wc, R.u[k], u); GS_TAP(acc_v, wc, R.v[k], v); }       \
        GS_TAP(acc_u, wr, ur, u); GS_TAP(acc_v, wr, vr, v);                                    \
    }
        if (mrow) GS_ROW_TAPS(m, 0, true)
        GS_ROW_TAPS(z, 1, false)
        if (prow) GS_ROW_TAPS(p, 2, true)
#undef GS_ROW_TAPS
    }
    react<(FAST & 2) != 0>(a, u, v, acc_u, acc_v, out_u, out_v);
}

} // namespace
