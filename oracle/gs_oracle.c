/*
 * gs_oracle.c -- CPU restatement of the reference's *naive* Gray-Scott backend.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The shipped path is the HIP library
 * (grayscott_amd/csrc -> libgs_hip.so) and it has no CPU fallback.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference workspace holds no golden
 * vectors, known-answer tests or fixtures for this path (its only #[test] is an
 * #[ignore]d Vulkan set-up check, compute/shared/src/gpu/mod.rs:103-124) and it
 * cannot be built here (Rust; no cargo/rustc in the image).  This restatement
 * is therefore pinned by (1) known answers derived by hand from the cited
 * source (tests/test_oracle_kat.py), and (2) agreement, bit for bit, with an
 * independently written numpy restatement (oracle/numpy_ref.py).
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   gs_oracle_step_rows   compute/naive/src/lib.rs:42-83   (the arithmetic spec)
 *   gs_oracle_init        data/src/concentration/mod.rs:36-59  (Species::new)
 *   gs_oracle_run         compute/shared/src/cpu.rs:30-42  (step; flip loop)
 *   gs_oracle_set_ftz     compute/shared/src/lib.rs:161-180 (DenormalsFlusher: MXCSR FTZ)
 *   default constants     data/src/parameters.rs:72-83,116-122
 *
 * Build: see oracle/Makefile.  -ffp-contract=off is mandatory: Rust never
 * contracts a*b+c, so every operation below is a separately rounded f32 op.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float w[3][3];          /* StencilWeights, row-major   (parameters.rs:87-88)  */
    float du, dv;           /* diffusion_rate_u / _v       (parameters.rs:19-22)  */
    float feed, kill, dt;   /* feed_rate, kill_rate, time_step (:25-32)           */
} gs_oracle_params;

/* Parameters::default(), data/src/parameters.rs:72-83 with the Oono-Puri
 * STENCIL_WEIGHTS of :116-122 (the default cargo feature set). */
void gs_oracle_default_params(gs_oracle_params *p)
{
    static const float w[3][3] = {{0.25f, 0.5f, 0.25f}, {0.5f, 0.0f, 0.5f}, {0.25f, 0.5f, 0.25f}};
    memcpy(p->w, w, sizeof w);
    p->du = 0.1f;
    p->dv = 0.05f;
    p->feed = 0.014f;
    p->kill = 0.054f;
    p->dt = 1.0f;
}

/* DenormalsFlusher (compute/shared/src/lib.rs:161-180): sets bit 0x8000 (FTZ)
 * of MXCSR on the calling thread; DAZ is left alone.  Returns the previous FTZ
 * state (0/1).  On non-SSE targets this is a no-op, as in the reference. */
int gs_oracle_set_ftz(int on)
{
#if defined(__SSE__)
    unsigned int csr = _mm_getcsr();
    int was = (csr & 0x8000u) != 0;
    csr = on ? (csr | 0x8000u) : (csr & ~0x8000u);
    _mm_setcsr(csr);
    return was;
#else
    (void)on;
    return 0;
#endif
}

/* Species::new (data/src/concentration/mod.rs:36-59): U=1, V=0 everywhere,
 * then the centre slice gets U=0, V=1.  The slice is, per axis i,
 *   [ (shape[i]*7/16).saturating_sub(shift) , (shape[i]*8/16).saturating_sub(shift) )
 * with shift = 4 for rows (i == 0) and 0 for columns. */
static size_t sat_sub(size_t a, size_t b) { return a > b ? a - b : 0; }

void gs_oracle_seed_ranges(size_t rows, size_t cols, size_t out[4])
{
    out[0] = sat_sub(rows * 7 / 16, 4);
    out[1] = sat_sub(rows * 8 / 16, 4);
    out[2] = cols * 7 / 16;
    out[3] = cols * 8 / 16;
}

void gs_oracle_init(float *u, float *v, size_t rows, size_t cols)
{
    size_t s[4];
    gs_oracle_seed_ranges(rows, cols, s);
    for (size_t i = 0; i < rows * cols; ++i) {
        u[i] = 1.0f;
        v[i] = 0.0f;
    }
    for (size_t r = s[0]; r < s[1]; ++r)
        for (size_t c = s[2]; c < s[3]; ++c) {
            u[r * cols + c] = 0.0f;
            v[r * cols + c] = 1.0f;
        }
}

/* One output cell, exactly as the closure of compute/naive/src/lib.rs:54-80:
 * the stencil window is clipped to the array (:57-60), the fold runs row-major
 * over the *clipped slice* and looks the weight up by the index inside that
 * slice (:63-71) -- so on the top row / left column the weights are anchored at
 * the window's top-left corner, not at the centre. */
static inline void naive_cell(const float *in_u, const float *in_v, float *out_u, float *out_v,
                              size_t rows, size_t cols, size_t r, size_t c,
                              const gs_oracle_params *p)
{
    const size_t rs = r > 0 ? r - 1 : 0;                  /* saturating_sub(stencil_offset) */
    const size_t cs = c > 0 ? c - 1 : 0;
    const size_t re = r + 2 < rows ? r + 2 : rows;        /* (pos + offset + 1).min(shape)  */
    const size_t ce = c + 2 < cols ? c + 2 : cols;
    const float u = in_u[r * cols + c];
    const float v = in_v[r * cols + c];
    float acc_u = 0.0f, acc_v = 0.0f;
    for (size_t i = 0; i < re - rs; ++i)
        for (size_t j = 0; j < ce - cs; ++j) {
            const float weight = p->w[i][j];
            const float su = in_u[(rs + i) * cols + (cs + j)];
            const float sv = in_v[(rs + i) * cols + (cs + j)];
            acc_u = acc_u + weight * (su - u);
            acc_v = acc_v + weight * (sv - v);
        }
    const float uv_square = u * v * v;                                        /* :74 */
    const float du = p->du * acc_u - uv_square + p->feed * (1.0f - u);        /* :75 */
    const float dv = p->dv * acc_v + uv_square - (p->feed + p->kill) * v;     /* :76-77 */
    out_u[r * cols + c] = u + du * p->dt;                                     /* :78 */
    out_v[r * cols + c] = v + dv * p->dt;                                     /* :79 */
}

/* The OTHER boundary rule of the reference (boundary-rule summary): full
 * 3x3 window, weights aligned on the centre, cells outside the grid read as 0 -- the Vulkan
 * backends' sampler (ClampToBorder + FloatOpaqueBlack, compute/gpu/naive/src/pipeline.rs:105-113;
 * the fold compute/gpu/naive/src/main.comp:37-44) and the zero halo of the SIMD storage
 * (data/src/concentration/simd/mod.rs:281-326).  Those backends fix neither the tap order nor
 * the contraction of their arithmetic, so this restatement keeps naive's: row-major fold,
 * acc = acc + w * (elem - centre), one rounding per operation, then the same update as above.
 * It is the checker of the library's GS_BOUNDARY_ZERO_HALO option; the parity target of the
 * round stays the clipped-window rule above. */
static inline void zero_halo_cell(const float *in_u, const float *in_v, float *out_u, float *out_v,
                                  size_t rows, size_t cols, size_t r, size_t c,
                                  const gs_oracle_params *p)
{
    const float u = in_u[r * cols + c];
    const float v = in_v[r * cols + c];
    float acc_u = 0.0f, acc_v = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const long rr = (long)r + i - 1, cc = (long)c + j - 1;
            const int inside = rr >= 0 && rr < (long)rows && cc >= 0 && cc < (long)cols;
            const float su = inside ? in_u[(size_t)rr * cols + (size_t)cc] : 0.0f;
            const float sv = inside ? in_v[(size_t)rr * cols + (size_t)cc] : 0.0f;
            acc_u = acc_u + p->w[i][j] * (su - u);
            acc_v = acc_v + p->w[i][j] * (sv - v);
        }
    const float uv_square = u * v * v;
    const float du = p->du * acc_u - uv_square + p->feed * (1.0f - u);
    const float dv = p->dv * acc_v + uv_square - (p->feed + p->kill) * v;
    out_u[r * cols + c] = u + du * p->dt;
    out_v[r * cols + c] = v + dv * p->dt;
}

static int g_boundary = 0; /* 0 = clipped window (naive), 1 = zero halo; see gs_oracle_set_boundary */

/* Selects the boundary rule of the following steps (test infrastructure: not thread-safe);
 * returns the previous one. */
int gs_oracle_set_boundary(int boundary)
{
    const int was = g_boundary;
    g_boundary = boundary ? 1 : 0;
    return was;
}

/* One step over output rows [r0, r1) of a dense [rows, cols] array.  Cells of
 * one step are independent, so any row partition (threads, slabs) is bit-exact. */
void gs_oracle_step_rows(const float *in_u, const float *in_v, float *out_u, float *out_v,
                         size_t rows, size_t cols, const gs_oracle_params *p,
                         size_t r0, size_t r1, int ftz, int nthreads)
{
    if (r1 > rows) r1 = rows;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
    {
        const int was = gs_oracle_set_ftz(ftz);
#pragma omp for schedule(static)
        for (size_t r = r0; r < r1; ++r)
            for (size_t c = 0; c < cols; ++c)
                if (g_boundary) zero_halo_cell(in_u, in_v, out_u, out_v, rows, cols, r, c, p);
                else naive_cell(in_u, in_v, out_u, out_v, rows, cols, r, c, p);
        gs_oracle_set_ftz(was);
    }
#else
    (void)nthreads;
    const int was = gs_oracle_set_ftz(ftz);
    for (size_t r = r0; r < r1; ++r)
        for (size_t c = 0; c < cols; ++c)
            if (g_boundary) zero_halo_cell(in_u, in_v, out_u, out_v, rows, cols, r, c, p);
            else naive_cell(in_u, in_v, out_u, out_v, rows, cols, r, c, p);
    gs_oracle_set_ftz(was);
#endif
}

/* Simulate::perform_steps for SimulateStep backends (compute/shared/src/cpu.rs:30-42):
 * `steps` x (perform_step; species.flip()).  buf[0]/buf[1] are the two slots of
 * U, buf[2]/buf[3] those of V; slot 0 is the input on entry.  Returns the slot
 * (0 or 1) that holds the result, i.e. the slot that is "input" after the last
 * flip (cpu.rs:38, concentration/mod.rs:181-186). */
int gs_oracle_run(float *u0, float *u1, float *v0, float *v1, size_t rows, size_t cols,
                  const gs_oracle_params *p, size_t steps, int ftz, int nthreads)
{
    float *u[2] = {u0, u1}, *v[2] = {v0, v1};
    int in = 0;
    for (size_t s = 0; s < steps; ++s) {
        gs_oracle_step_rows(u[in], v[in], u[1 - in], v[1 - in], rows, cols, p, 0, rows, ftz,
                            nthreads);
        in = 1 - in;
    }
    return in;
}
