// gs_single_step.h -- one step per launch: the one-thread-per-cell cross-check kernel, the streaming kernel of gs_step
// (HBM-bound: the leg north_star asks the rocprof evidence for) and the LDS-staged alternative.
// Part of the gfx950 step kernels: included by gs_step_kernels.hip (which sets GS_MATH_FUSED / GS_TB_OP_ONLY and the
// GS_SUFFIX / GS_TAP macros) inside one translation unit per arithmetic flavour; not a header to include elsewhere.
#pragma once

namespace {

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

} // namespace
