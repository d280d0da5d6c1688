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

#if !GS_TB_OP_ONLY
// ------------------------------------------------------------------------------------
// Cross-check kernel: literal restatement, one thread per cell.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void GS_SUFFIX(gs_step_simple_k)(GsStepArgs a)
{
    const int bpr = (a.cols + 255) >> 8;
    const int slot = blockIdx.x / bpr;
    const int c = (blockIdx.x - slot * bpr) * 256 + threadIdx.x;
    const int r = range_row(a, slot);
    if (c >= a.cols) return;

    const bool top = (r > 0) || a.top_present;
    const bool bottom = (r + 1 < a.rows) || a.bottom_present;
    const bool left = c > 0;
    const bool right = c + 1 < a.cols;
    const ptrdiff_t pitch = a.pitch;
    const ptrdiff_t o = (ptrdiff_t)r * pitch + c;
    const float u = a.in_u[o], v = a.in_v[o];

    float acc_u = 0.0f, acc_v = 0.0f;
    if (a.zero_halo) { // full window, centred weights, zeros outside the grid
        for (int di = -1; di <= 1; ++di)
            for (int dj = -1; dj <= 1; ++dj) {
                const bool inside = (di >= 0 || top) && (di <= 0 || bottom) && (dj >= 0 || left) && (dj <= 0 || right);
                const float su = inside ? a.in_u[o + di * pitch + dj] : 0.0f;
                const float sv = inside ? a.in_v[o + di * pitch + dj] : 0.0f;
                GS_TAP(acc_u, a.w[di + 1][dj + 1], su, u);
                GS_TAP(acc_v, a.w[di + 1][dj + 1], sv, v);
            }
    } else {
        const int i_off = top ? 1 : 0, j_off = left ? 1 : 0;
        for (int di = top ? -1 : 0; di <= (bottom ? 1 : 0); ++di)
            for (int dj = left ? -1 : 0; dj <= (right ? 1 : 0); ++dj) {
                const float w = a.w[di + i_off][dj + j_off];
                const float su = a.in_u[o + di * pitch + dj];
                const float sv = a.in_v[o + di * pitch + dj];
                GS_TAP(acc_u, w, su, u);
                GS_TAP(acc_v, w, sv, v);
            }
    }
    float ou, ov;
    react(a, u, v, acc_u, acc_v, ou, ov);
    a.out_u[o] = ou;
    a.out_v[o] = ov;
}

#endif // !GS_TB_OP_ONLY

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

#if !GS_TB_OP_ONLY
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
#endif // !GS_TB_OP_ONLY

template <int G, bool EDGE>
__device__ __forceinline__ void march(const GsStepArgs &a, int ur0, int ur1, int c0, int lane)
{
    const int c = c0 + lane * 4;
    LaneCtx lc;
    lc.lane_ok = !EDGE || (c < a.pitch);
    lc.halo_off = (lane == 0) ? -1 : 4;
    lc.halo_ok = EDGE ? ((lane == 0 && c0 > 0) || (lane == 63 && c + 4 < a.pitch))
                      : (lane == 0 || lane == 63);

    const ptrdiff_t pitch = a.pitch;
    const float *bu = a.in_u + c, *bv = a.in_v + c; // row 0 of this lane's columns
    float *ou = a.out_u + (ptrdiff_t)ur0 * pitch + c;
    float *ov = a.out_v + (ptrdiff_t)ur0 * pitch + c;

    // Rows are fetched one group (G rows) ahead of the group being computed.  Row indices
    // are clamped to ur1 (the row below the last output row, at most the bottom ghost
    // row), so every load is in bounds and the tail needs no branches around loads.
    auto fetch = [&](int row) {
        const int rr = row < ur1 ? row : ur1;
        return load_row<EDGE>(bu + (ptrdiff_t)rr * pitch, bv + (ptrdiff_t)rr * pitch, lc);
    };

    RowW q[G + 2];
    RowIn n[G];
    q[0] = widen(fetch(ur0 - 1));
    q[1] = widen(fetch(ur0));
#pragma unroll
    for (int g = 0; g < G; ++g) n[g] = fetch(ur0 + 1 + g);

    // per-lane masks (all ones = that neighbour column is clipped away).  c is a multiple of 4,
    // so only the first of a lane's four cells can sit on the global left edge.
    uint32_t la[4], ra[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        la[k] = (EDGE && k == 0 && c == 0) ? 0xffffffffu : 0u;
        ra[k] = (EDGE && (c + k + 1 >= a.cols)) ? 0xffffffffu : 0u;
        if (EDGE) { // keep the masks opaque, or the compiler turns every blend back into v_cndmask
            if (k == 0) asm volatile("" : "+v"(la[k]));
            asm volatile("" : "+v"(ra[k]));
        }
    }

    for (int r = ur0; r < ur1; r += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) q[g + 2] = widen(n[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) n[g] = fetch(r + G + 1 + g);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int row = r + g;
            if (row < ur1) {
                const bool mrow = !EDGE || (row > 0) || a.top_present;
                const bool prow = !EDGE || (row + 1 < a.rows) || a.bottom_present;
                float4 nu, nv;
                cell<EDGE>(a, q[g], q[g + 1], q[g + 2], 1, mrow, prow, la[0], ra[0], nu.x, nv.x);
                cell<EDGE>(a, q[g], q[g + 1], q[g + 2], 2, mrow, prow, la[1], ra[1], nu.y, nv.y);
                cell<EDGE>(a, q[g], q[g + 1], q[g + 2], 3, mrow, prow, la[2], ra[2], nu.z, nv.z);
                cell<EDGE>(a, q[g], q[g + 1], q[g + 2], 4, mrow, prow, la[3], ra[3], nu.w, nv.w);
                if (lc.lane_ok) {
                    *reinterpret_cast<float4 *>(ou) = nu;
                    *reinterpret_cast<float4 *>(ov) = nv;
                }
                ou += pitch;
                ov += pitch;
            }
        }
        q[0] = q[G];
        q[1] = q[G + 1];
    }
}

template <int G>
__global__ __launch_bounds__(256) void GS_SUFFIX(gs_step_stream_k)(GsStepArgs a)
{
    const int lane = threadIdx.x & 63;
    // readfirstlane tells the compiler the wave index is wave-uniform: everything derived
    // from it (unit, row range, edge flags) then lives in SGPRs and branches are scalar.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strips = (a.cols + 255) >> 8;
    int block = (int)blockIdx.x;
    if (a.xcd_m > 0) { // XCD-aware order (GsStepArgs::xcd_m)
        const int per = 8 * a.xcd_m, g = block / per, o = block - g * per;
        if ((g + 1) * per <= (int)gridDim.x) block = g * per + (o & 7) * a.xcd_m + (o >> 3);
    }
    const int unit = block * 4 + wave;
    const int chunk = unit / strips;
    const int strip = unit - chunk * strips;
    const int rpu = a.rows_per_unit;
    const int chunks_a = (a.ra1 - a.ra0 + rpu - 1) / rpu;
    const int chunks_b = (a.rb1 - a.rb0 + rpu - 1) / rpu;
    if (chunk >= chunks_a + chunks_b) return; // wave-uniform

    int ur0, ur1;
    if (chunk < chunks_a) {
        ur0 = a.ra0 + chunk * rpu;
        ur1 = min(ur0 + rpu, a.ra1);
    } else {
        ur0 = a.rb0 + (chunk - chunks_a) * rpu;
        ur1 = min(ur0 + rpu, a.rb1);
    }
    const int c0 = strip << 8;
    // Units that touch a global edge or the ragged right end take the general path; the
    // interior path has no per-lane bounds logic at all.
    const bool edge = (c0 == 0) || (c0 + 256 >= a.cols) || (ur0 == 0 && !a.top_present) ||
                      (ur1 == a.rows && !a.bottom_present);
    if (edge)
        march<G, true>(a, ur0, ur1, c0, lane);
    else
        march<G, false>(a, ur0, ur1, c0, lane);
}

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
// may receive anything).  Measured on MI355X (tools/ubench/valu_rate2.hip, profiles/r02_sweeps.md): a
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

// Buffer-instruction forms of the plane accesses: address = 128-bit resource in SGPRs (base pointer of
// the unit's first row) + per-lane byte offset (one VGPR for the whole march) + scalar byte offset of
// the row: no 64-bit per-lane addresses to keep or to recompute per row.  The resource is raw (stride 0)
// with the widest record count: the units never step outside their planes, so nothing relies on the
// range check.  With them, the late fetch (GS_TB_LATE_FETCH, gs_experiments.h) and the edge path's column
// masks kept as lane masks in SGPRs, the whole kernel entry -- general path included -- fits 126
// registers: 4 waves per SIMD instead of 3, +9 % at 16384^2 (profiles/r02_sweeps.md, section 8).
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
// lone wave issues at most every 4th cycle, half the SIMD's rate (tools/wave_timeline.py, profiles/r03_sweeps.md
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
__host__ __device__ constexpr int tb_halo_floats(int k) { return k * 2 * kHaloRow; }

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
        float *const mine = fb.halo + 1 + lane; // this lane's element of the first array of (level 0, slot 0)
        if (lane < 2) // elements 0 and 65 of every array: read by the sacrificial lanes only
#pragma unroll
            for (int i = 0; i < K * 2 * 4; ++i) fb.halo[i * kHaloArray + lane * (kHaloArray - 1)] = 0.0f;
        // a new row of level j: its columns go to the board, for the neighbouring lanes
        auto put = [&](int j, int slot, const RowQ<CPL> &r) {
            float *b = mine + (j * 2 + slot) * kHaloRow;
            b[0] = r.u[0]; b[kHaloArray] = r.v[0]; b[2 * kHaloArray] = r.u[1]; b[3 * kHaloArray] = r.v[1];
            // The elements a lane reads back are written by its NEIGHBOURS, in the same two instructions: to the
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
            w.u[0] = b[2 * kHaloArray - 1]; w.v[0] = b[3 * kHaloArray - 1]; // second column of lane - 1
            w.u[3] = b[1];                  w.v[3] = b[kHaloArray + 1];     // first column of lane + 1
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
                const RowT<CPL> z = widened(j - 1, par ^ 1), p = widened(j - 1, par);
                cells_vshare<FAST, CPL>(a, z, p, C[j - 1], nu, nv);
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
    if constexpr ((FAST & 5) == 5 && !GS_MATH_FUSED && CPL == 2) { // the variant with full difference sharing: halo boards
        __shared__ float halo_boards[WG * tb_halo_floats(K)];
        fb.halo = halo_boards + wave * tb_halo_floats(K);
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
// The variants with full difference sharing (FAST = 7, 2 columns per lane) are kernels of their own: built for four
// waves per SIMD (the register allocator is told so; left to itself it settles a few registers above 128).
template <int K, int WG = 4>
__global__ __launch_bounds__(WG * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void GS_SUFFIX(gs_step_tb_ds_k)(GsStepArgs a)
{
    tb_unit<K, 7, 2, WG>(a);
}

#if !GS_TB_OP_ONLY
// ------------------------------------------------------------------------------------
// Mid-size grids: K <= 8 time steps per launch on LDS-resident windows, one cell per lane and row.
//
// Between the single-workgroup resident kernel (<= 1536 cells) and grids that fill the chip with
// marching waves (~1 M cells and up), a pass of gs_step_tb_k is bound by the LENGTH of a wave's march
// (unit height + 2K ticks of K levels, one wave per SIMD issuing every 4th cycle) plus a dependent
// launch per K <= 4 steps: 2.3-3.4 us per step whatever the grid (profiles/r02_criterion_grid.md).
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
// The first form of this kernel (4-cell strips, 2-8 waves per tile; profiles/r02_sweeps.md, section 4)
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

// ------------------------------------------------------------------------------------
// Grids of ONE round of register-resident windows (1.5-2.3 M cells, what 256 windows of 72 x 120 owned cells cover: the reference's default 1080 x 1920): the whole
// gs_run in one persistent launch, aprons traded between workgroups inside it.
//
// At these sizes a pass of the marching kernel is 19 us for 4 steps of which ~9 are fixed (launch gap, dispatch,
// first-rows burst, level-pipeline fill on memory latency) and its 10-row units recompute 30 % of their rows
// (profiles/r03_sweeps.md, sections 1-4).  Here a workgroup of 16 waves owns a window of 16 * RPW rows x 128
// columns for the whole run: a wave keeps RPW whole rows in registers, two columns per lane (10 cells per lane at
// RPW = 5).  Per step the columns next to a lane's two come from the adjacent lanes (DPP wave shifts), only the
// first and the last row of a wave's band go through LDS for the waves above and below (double-buffered by the
// step's parity: one workgroup barrier per step, reached after the RPW - 2 rows that need nothing from other waves),
// and every cell is updated by the same cell<> code as in every other kernel: bit-identical.  The window's outer K
// cells are an apron: they lose their validity one ring per step.  After K steps the workgroup stores the K-cell
// ring of the cells it OWNS (the window shrunk by K) into an exchange plane with sc1 stores, drains, raises its
// flag, polls the flags of its up to 8 neighbours and reloads its apron from their rings with sc1 loads -- the
// hand-off form MI355X_MICROARCH.md lists as valid for one workgroup per CU (one lane signals for all stores of the
// workgroup behind a barrier; the polling wave joins a barrier before anybody loads; all stores and loads sc1),
// measured for exactly this shape in tools/ubench/handoff_probe.hip: 4.3 us per exchange, no stale word.  Exchanges
// alternate between two sets of exchange planes, so a workgroup that is one exchange ahead never overwrites what a
// neighbour still has to read.  The input planes are only read and the output planes only written at the very end.
// Every poll is bounded: a workgroup that runs out of patience (its neighbours are not resident: the GPU is shared
// with another long-running kernel) sets a sticky abort word and every workgroup leaves; gs_sync reports it.
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
__host__ __device__ constexpr size_t win_lds_bytes() { return (size_t)2 * 2 * kWinWaves * 2 * kWinPitch * sizeof(float); }

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
    auto update = [&](int r, const RowT<2> &m, const RowT<2> &z, const RowT<2> &p) {
        const int row = gr + r; // wave-uniform
        if (EDGE != 0 && (row < 0 || row >= a.rows)) return; // a row outside the grid: zeros that stay zeros
        const bool mrow = !ROWS || row > 0, prow = !ROWS || row + 1 < a.rows;
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
    // A step: publish, the rows that need nothing from other waves (the other waves' rows arrive meanwhile), barrier, the
    // rows above and below from LDS, the band's first and last row.  (Reads first and the publish for the next step
    // right before the barrier -- the LDS latency behind the middle rows -- was measured: the waves of a workgroup
    // drift apart, 418 k against 461 k at 1080 x 1920, profiles/r04_window_kernel.md.)
    for (int s = 0; s < n; ++s, ++step) {
        const int buf = step & 1;
        publish(buf);
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
        __syncthreads();
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
#if defined(GS_WIN_TRACE)
__device__ unsigned long long gs_win_trace[1024 * 8 * 8];
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
#define GS_WIN_TRACE_AT(SLOT) do { } while (0)
#endif

template <int RPW, int EDGE, int FAST, int ZH>
__device__ __forceinline__ void window_run(const GsStepArgs &a, const GsWindowArgs &x, const GsWindowDesc *d, int OH, int OW, float *lds,
                                           int *go, int wg, int gr, int gc, int wave, int lane, float (&u)[RPW][2], float (&v)[RPW][2])
{
    // OH x OW: the cells this workgroup owns = window rows [K, K + OH) x window columns [K, K + OW)
    constexpr int SC1 = 16;
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int K = x.k, wc = 2 * lane; // wc: this lane's first window column
    int step = 0;
    const int supers = (x.steps + K - 1) / K;
    for (int s = 0; s < supers; ++s) {
        GS_WIN_TRACE_AT(0);
        // the short super-step first
        window_steps<RPW, EDGE, FAST, ZH>(a, lds, (s == 0 && x.steps % K) ? x.steps % K : K, step, gr, gc, wave, lane, u, v);
        GS_WIN_TRACE_AT(1);
        if (s == supers - 1) break;
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
    }
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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = x.k;
    const int wg = (int)blockIdx.x;
    const GsWindowDesc *d = x.desc + wg;                    // (uniform: scalar loads)
    const int H = d->active, OH = d->oh, OW = d->ow;        // window rows in use; owned rows and columns
    const int gr0 = d->r0 - K, gc0 = d->c0 - K;             // global coordinates of window cell (0, 0)
    const int gr = gr0 + wave * RPW, gc = gc0 + 2 * lane;   // this lane's first cell
    // A launch enqueued behind one that gave up leaves at once (nothing of it is valid anyway).  ONE wave reads the
    // word for the whole workgroup: waves that read it for themselves could disagree (a workgroup of this launch may
    // give up at any time) and a barrier below would wait for waves that have left.
    if (wave == 0 && lane == 0) go = __builtin_amdgcn_raw_buffer_load_b32(win_rsrc(x.abort), 0, 0, SC1) == 0;
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
        // and then only keeps the workgroup's barrier count: one per step, two per exchange.
#pragma unroll
        for (int b = 0; b < 8; ++b)
            { float *p = lds + (((b >> 1) * kWinWaves + wave) * 2 + (b & 1)) * kWinPitch + 1 + lane; p[0] = 0.0f; p[kWinHalf] = 0.0f; }
        const int supers = (x.steps + K - 1) / K;
        for (int s = 0; s < supers; ++s) {
            const int n = (s == 0 && x.steps % K) ? x.steps % K : K;
            for (int i = 0; i < n; ++i) __syncthreads();
            if (s == supers - 1) break;
            __syncthreads();
            __syncthreads();
            if (!go) return;
        }
        return;
    }
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
#endif // !GS_TB_OP_ONLY

#if !GS_TB_OP_ONLY
// ------------------------------------------------------------------------------------
// LDS-staged variant (one step per launch): the (tile + halo) stencil window of a block is
// staged in LDS, then every lane reads its 3 x 6 neighbourhood back with ds_read_b128 +
// two ds_read_b32 per row and species.  Kept as a measured alternative to the register
// sliding window of gs_step_stream_k (north_star names LDS staging explicitly): it moves the
// same HBM bytes, but adds an LDS write + read pass and a barrier per tile, and loses the
// row-to-row register reuse (each input row is read from LDS three times).  Slower than the
// stream kernel on MI355X, so GS_KERNEL_AUTO never picks it.
// ------------------------------------------------------------------------------------
constexpr int kLdsTileRows = 16;          // output rows per block (38 KB of LDS -> 4 blocks per CU)
constexpr int kLdsRowFloats = 256 + 8;    // 4 halo floats each side keep float4 alignment

template <bool EDGE>
__device__ __forceinline__ void lds_tile(const GsStepArgs &a, int tr0, int tr1, int c0, float *su, float *sv)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = c0 + lane * 4;
    const ptrdiff_t pitch = a.pitch;
    const bool lane_ok = !EDGE || (c < a.pitch);
    const bool halo_l = (lane == 0) && (!EDGE || c0 > 0);
    const bool halo_r = (lane == 63) && (!EDGE || c + 4 < a.pitch);
    // stage rows [tr0 - 1, tr1 + 1) of both species; row r lands in LDS row (r - tr0 + 1)
    const int nrows = tr1 - tr0 + 2;
    for (int lr = wave; lr < nrows; lr += 4) {
        const int r = tr0 - 1 + lr; // ghost rows exist physically, so every row is loadable
        float4 fu = make_float4(0.f, 0.f, 0.f, 0.f), fv = fu;
        if (lane_ok) {
            fu = *reinterpret_cast<const float4 *>(a.in_u + (ptrdiff_t)r * pitch + c);
            fv = *reinterpret_cast<const float4 *>(a.in_v + (ptrdiff_t)r * pitch + c);
        }
        float *du = su + lr * kLdsRowFloats + 4 + lane * 4;
        float *dv = sv + lr * kLdsRowFloats + 4 + lane * 4;
        *reinterpret_cast<float4 *>(du) = fu;
        *reinterpret_cast<float4 *>(dv) = fv;
        if (halo_l) {
            du[-1] = a.in_u[(ptrdiff_t)r * pitch + c - 1];
            dv[-1] = a.in_v[(ptrdiff_t)r * pitch + c - 1];
        }
        if (halo_r) {
            du[4] = a.in_u[(ptrdiff_t)r * pitch + c + 4];
            dv[4] = a.in_v[(ptrdiff_t)r * pitch + c + 4];
        }
    }
    __syncthreads();

    // per-lane masks (all ones = that neighbour column is clipped away).  c is a multiple of 4,
    // so only the first of a lane's four cells can sit on the global left edge.
    uint32_t la[4], ra[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        la[k] = (EDGE && k == 0 && c == 0) ? 0xffffffffu : 0u;
        ra[k] = (EDGE && (c + k + 1 >= a.cols)) ? 0xffffffffu : 0u;
        if (EDGE) { // keep the masks opaque, or the compiler turns every blend back into v_cndmask
            if (k == 0) asm volatile("" : "+v"(la[k]));
            asm volatile("" : "+v"(ra[k]));
        }
    }
    auto read_row = [&](int lr) {
        RowW w;
        const float *pu = su + lr * kLdsRowFloats + 4 + lane * 4;
        const float *pv = sv + lr * kLdsRowFloats + 4 + lane * 4;
        const float4 fu = *reinterpret_cast<const float4 *>(pu);
        const float4 fv = *reinterpret_cast<const float4 *>(pv);
        w.u[1] = fu.x; w.u[2] = fu.y; w.u[3] = fu.z; w.u[4] = fu.w;
        w.v[1] = fv.x; w.v[2] = fv.y; w.v[3] = fv.z; w.v[4] = fv.w;
        w.u[0] = pu[-1]; w.u[5] = pu[4];
        w.v[0] = pv[-1]; w.v[5] = pv[4];
        return w;
    };
    for (int r = tr0 + wave; r < tr1; r += 4) {
        const int lr = r - tr0 + 1;
        const RowW m = read_row(lr - 1), z = read_row(lr), p = read_row(lr + 1);
        const bool mrow = !EDGE || (r > 0) || a.top_present;
        const bool prow = !EDGE || (r + 1 < a.rows) || a.bottom_present;
        float4 nu, nv;
        cell<EDGE>(a, m, z, p, 1, mrow, prow, la[0], ra[0], nu.x, nv.x);
        cell<EDGE>(a, m, z, p, 2, mrow, prow, la[1], ra[1], nu.y, nv.y);
        cell<EDGE>(a, m, z, p, 3, mrow, prow, la[2], ra[2], nu.z, nv.z);
        cell<EDGE>(a, m, z, p, 4, mrow, prow, la[3], ra[3], nu.w, nv.w);
        if (lane_ok) {
            *reinterpret_cast<float4 *>(a.out_u + (ptrdiff_t)r * pitch + c) = nu;
            *reinterpret_cast<float4 *>(a.out_v + (ptrdiff_t)r * pitch + c) = nv;
        }
    }
}

__global__ __launch_bounds__(256) void GS_SUFFIX(gs_step_lds_k)(GsStepArgs a)
{
    __shared__ __attribute__((aligned(16))) float su[(kLdsTileRows + 2) * kLdsRowFloats];
    __shared__ __attribute__((aligned(16))) float sv[(kLdsTileRows + 2) * kLdsRowFloats];
    const int strips = (a.cols + 255) >> 8;
    const int chunk = blockIdx.x / strips;
    const int strip = blockIdx.x - chunk * strips;
    const int chunks_a = (a.ra1 - a.ra0 + kLdsTileRows - 1) / kLdsTileRows;
    int tr0, tr1;
    if (chunk < chunks_a) {
        tr0 = a.ra0 + chunk * kLdsTileRows;
        tr1 = min(tr0 + kLdsTileRows, a.ra1);
    } else {
        tr0 = a.rb0 + (chunk - chunks_a) * kLdsTileRows;
        tr1 = min(tr0 + kLdsTileRows, a.rb1);
    }
    const int c0 = strip << 8;
    const bool edge = (c0 == 0) || (c0 + 256 >= a.cols) || (tr0 == 0 && !a.top_present) ||
                      (tr1 == a.rows && !a.bottom_present);
    if (edge)
        lds_tile<true>(a, tr0, tr1, c0, su, sv);
    else
        lds_tile<false>(a, tr0, tr1, c0, su, sv);
}

#endif // !GS_TB_OP_ONLY

} // namespace

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
    static const char *const names[2] = {"window-r5/" GS_MATH_NAME, "window-r5/" GS_MATH_NAME ".op"};
    if (a.rows <= 0 || a.cols <= 0 || a.top_present || a.bottom_present || rpw != 5 || x.steps < 1 || x.k < 2 ||
        x.k > 8 || (x.k & 1) || 2 * x.k >= win_rows(rpw) || 2 * x.k + 2 > kWinCols || !x.flags || !x.abort || !x.xu[0] || !x.xu[1] || !x.xv[0] || !x.xv[1])
        return hipErrorInvalidValue;
    if (!x.desc || x.n_windows < 1 || x.seq < 1) return hipErrorInvalidValue;
    // byte offsets inside a plane are 32-bit in the kernel
    if ((long)(a.rows + 8) * a.pitch * 4 > 0x7fffffffL) return hipErrorInvalidValue;
    int fast = a.fast & (GS_MATH_FUSED ? 0 : 3);
    if (fast != 3) fast = 0; // only the variant for the default parameters is built besides the general one
    if (name) *name = names[fast ? 1 : 0];
    const void *fn = nullptr;
#define GS_WIN_FN(R) (fast ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_run_window_k)<R, GS_MATH_FUSED ? 0 : 3>) \
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
    // Groups of 8 x 64 workgroups lose 6 % (profiles/r03_sweeps.md, section 12).  GS_HIP_XCD_M_STREAM = 0 / n: off / 8 n.
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
        return cpl == 1 ? reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 0, 1, 16>)
                        : reinterpret_cast<const void *>(&GS_SUFFIX(gs_step_tb_k)<4, 0, 2, 16>);
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
// workgroup: 0 (general), 1 (side weights 0.5), 3 (and dt == 1) or 7 (and full difference sharing: built for 2 columns
// per lane, 2 to 4 fused steps).
static int tb_reduce_fast(int fast, int k = 0, int cpl = 0, int wg = 4)
{
    // The fused build has no use for bit 0 (its taps are sub + fma already) and measured slower
    // with bit 1 (profiles/r01_sweeps.md, runs 48/49): it always runs the general variant.  dt == 1
    // alone (fast == 2) is not worth a variant either, and bit 2 means nothing without the other two.
    fast &= GS_MATH_FUSED ? 0 : 7;
    if (!(fast & 1)) return 0;
    if (fast == 7) {
#if !GS_MATH_FUSED
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
      "tb-k4" C "/" GS_MATH_NAME ".op.ds"}}
    static const char *const names[3][3][4] = {GS_TB_NAMES("c1"), GS_TB_NAMES("c2"), GS_TB_NAMES("")};
#undef GS_TB_NAMES
    // "f": the fair-progress form (16-wave workgroups) of one-round launches
    static const char *const names16[2][3] = {{"tb-k4c1f/" GS_MATH_NAME, "tb-k4c1f/" GS_MATH_NAME ".op", "tb-k4c1f/" GS_MATH_NAME ".op.ds"},
                                              {"tb-k4c2f/" GS_MATH_NAME, "tb-k4c2f/" GS_MATH_NAME ".op", "tb-k4c2f/" GS_MATH_NAME ".op.ds"}};
    if (k < 1 || k > 4 || a.cols <= 0 || a.rows_per_unit <= 0) return hipErrorInvalidValue;
    const int cpl = a.cpl == 0 ? 4 : a.cpl;
    if (cpl != 1 && cpl != 2 && cpl != 4) return hipErrorInvalidValue;
    const int fast = tb_reduce_fast(a.fast, k, cpl);
    if (name) *name = names[cpl == 1 ? 0 : (cpl == 2 ? 1 : 2)][fast == 7 ? 2 : (fast ? 1 : 0)][k - 1];
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
    // Measured at 16384^2 (profiles/r02_sweeps.md, section 7): +1...2 % over round 1's single level
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
    // profiles/r02_sweeps.md, section 11).  The kernel's dispatch order: the outer strips of every chunk, then all strips of the
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
    // waits.  Free-running / in step, same box (profiles/r03_sweeps.md, section 2): 1 column per lane, 10-row
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
        if (name) *name = names16[cpl == 1 ? 0 : 1][fast16 == 7 ? 2 : (fast16 ? 1 : 0)];
        return hipLaunchKernel(fair_fn, dim3((unsigned)((units + 15) / 16)), dim3(1024), kargs, 0, s);
    }
    const long blocks = (units + 3) / 4;
    if (blocks > 0x7fffffffL) return hipErrorInvalidConfiguration;
    // XCD-aware unit order (GsStepArgs::xcd_m): the dispatcher deals workgroups over the 8 XCDs round-robin, so
    // four-strip neighbours in the grid land on eight different L2s and each fetches the columns and rows their
    // windows share for itself.  With every XCD taking 16 consecutive workgroups of each group of 128, the HBM
    // reads of a 16384^2 launch fall from 2.376 to 2.239 GiB (minimum 2.0; FETCH_SIZE, tools/fetch_ab.sh) and the
    // launch gains 0.3-0.5 % (8192^2 +0.8 %, 4 / 8 slabs on one GPU +1.4 / +0.6 %).  Larger groups read no less
    // (68: 2.226 GiB) and run slower (-2 %, 136: -6 %: an XCD's share of the last groups is all tall or all short
    // units).  Launches of about one round keep the plain order: 1080 x 1920 loses 1.2 % with the renumbering
    // (profiles/r03_sweeps.md, section 12).  GS_HIP_XCD_M = 0 / n forces it off / to groups of 8 n.
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
