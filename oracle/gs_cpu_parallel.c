/*
 * gs_cpu_parallel.c -- CPU restatement of the reference's parallel(block(autovec))
 * backend, used ONLY as the timing baseline that bench.py reports next to the
 * GPU number ("cpu_baseline", kind "port").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never on the product path.
 *
 * It is NOT the parity target: this backend family uses a different boundary
 * rule from `naive` (full 3x3 window over a zero halo, SURVEY.md section 8
 * "Boundary-rule summary") and a different association (FMA chains), so its
 * results differ from the naive oracle on the border and in the last bits.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   storage layout / halo rebuild   data/src/concentration/simd/mod.rs:21-66,281-326
 *   fill_slice on that layout       data/src/concentration/simd/mod.rs:248-279
 *   inner kernel (3 FMA chains)     compute/autovec/src/lib.rs:63-115, width :120-138
 *   cache blocking recursion        compute/block/src/lib.rs:62-111
 *   fork-join row decomposition     compute/parallel/src/lib.rs:64-120
 *   grid_len / grid_line_len / split_grid   compute/shared/src/cpu.rs:92-154
 *   corrected weights, -(F+k)       data/src/parameters.rs:57-69
 *
 * Deviations, all timing-neutral or in the CPU's favour:
 *   - rayon's fork-join over an adaptive splitter is replaced by a persistent team with a
 *     static partition: the leaves that compute/parallel/src/lib.rs:103-117 produces when
 *     it splits all the way down to the sequential threshold are listed once, in row order,
 *     and thread t of T takes the t-th contiguous share of them every step (one OpenMP
 *     parallel region per perform_steps call, two barriers per step).  rayon keeps a hot
 *     pool and steals; a fork/join per step with OpenMP tasks does neither, and under a
 *     cgroup CPU quota its idle spinning throttled the whole process (round 1: the port ran
 *     10x SLOWER than the naive restatement at 1080x1920 on the GPU box);
 *   - FTZ is set on every worker thread when `ftz` is non-zero; the reference's
 *     DenormalsFlusher only covers the calling thread
 *     (compute/shared/src/lib.rs:112-113), so its rayon workers run unflushed.
 */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Vector width selection, compute/autovec/src/lib.rs:120-138 (AVX-512 disabled there). */
#if defined(__AVX__)
#define GSW 8
typedef __m256 vf;
#define V_SET1(x) _mm256_set1_ps(x)
#define V_MUL(a, b) _mm256_mul_ps(a, b)
#define V_ADD(a, b) _mm256_add_ps(a, b)
#define V_SUB(a, b) _mm256_sub_ps(a, b)
#if defined(__FMA__)
#define V_FMA(a, b, c) _mm256_fmadd_ps(a, b, c)
#else
#define V_FMA(a, b, c) _mm256_add_ps(_mm256_mul_ps(a, b), c)
#endif
#else
#define GSW 4
typedef __m128 vf;
#define V_SET1(x) _mm_set1_ps(x)
#define V_MUL(a, b) _mm_mul_ps(a, b)
#define V_ADD(a, b) _mm_add_ps(a, b)
#define V_SUB(a, b) _mm_sub_ps(a, b)
#if defined(__FMA__)
#define V_FMA(a, b, c) _mm_fmadd_ps(a, b, c)
#else
#define V_FMA(a, b, c) _mm_add_ps(_mm_mul_ps(a, b), c)
#endif
#endif

typedef union {
    vf v;
    float f[GSW];
} vlanes;

typedef struct {
    float w[3][3];
    float du, dv, feed, kill, dt;
} gs_par_params;

/* One SIMDConcentration: [(L+2), (C+2)] vectors, L = rows / GSW. */
typedef struct {
    vf *simd;
    size_t rows, cols; /* scalar shape */
    size_t L, stride;  /* simd rows of the centre, row stride in vectors (= cols + 2) */
} gs_par_conc;

typedef struct {
    gs_par_params p;
    gs_par_conc c[4]; /* u slot0, u slot1, v slot0, v slot1 */
    int in;           /* which slot is the input */
    size_t max_values_per_line, max_values_per_block, seq_len_threshold;
    int nthreads, ftz;
    void *leaves; /* leaf_t[n_leaves]: the parallel decomposition, built once */
    size_t n_leaves;
} gs_par_sim;

int gs_par_width(void) { return GSW; }

static int set_ftz(int on)
{
    unsigned int csr = _mm_getcsr();
    int was = (csr & 0x8000u) != 0;
    _mm_setcsr(on ? (csr | 0x8000u) : (csr & ~0x8000u));
    return was;
}

/* from_scalar_elem (simd/mod.rs:93-113): broadcast, then zero the left/right edge columns. */
static int conc_alloc(gs_par_conc *c, size_t rows, size_t cols, float elem)
{
    if (rows % GSW) return -1; /* simd_shape assert, simd/mod.rs:83-87 */
    c->rows = rows;
    c->cols = cols;
    c->L = rows / GSW;
    c->stride = cols + 2;
    size_t n = (c->L + 2) * c->stride;
    if (posix_memalign((void **)&c->simd, 64, n * sizeof(vf))) return -2;
    const vf e = V_SET1(elem), z = V_SET1(0.0f);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < c->L + 2; ++i) {
        vf *row = c->simd + i * c->stride;
        row[0] = z;
        for (size_t j = 1; j <= cols; ++j) row[j] = e;
        row[cols + 1] = z;
    }
    return 0;
}

/* fill_slice (simd/mod.rs:248-279): scalar row r lives in lane r / L, simd row r % L. */
static void conc_fill_slice(gs_par_conc *c, size_t r0, size_t r1, size_t c0, size_t c1, float value)
{
    for (size_t r = r0; r < r1; ++r) {
        size_t lane = r / c->L, i = r % c->L;
        vf *row = c->simd + (i + 1) * c->stride + 1;
        for (size_t j = c0; j < c1; ++j) {
            vlanes t;
            t.v = row[j];
            t.f[lane] = value;
            row[j] = t.v;
        }
    }
}

/* finalize (simd/mod.rs:281-326) for a stencil offset of 1: the bottom halo row is the
 * first centre row shifted one lane towards lane 0, the top halo row is the last centre
 * row shifted one lane away from lane 0; zeros are shifted in. */
static void conc_finalize(gs_par_conc *c, size_t j0, size_t j1) /* columns [j0, j1) */
{
    vf *top = c->simd + 1, *bottom = c->simd + (c->L + 1) * c->stride + 1;
    const vf *first = c->simd + 1 * c->stride + 1, *last = c->simd + c->L * c->stride + 1;
    for (size_t j = j0; j < j1; ++j) {
        vlanes s, d;
        s.v = first[j];
        for (int l = 0; l < GSW - 1; ++l) d.f[l] = s.f[l + 1];
        d.f[GSW - 1] = 0.0f;
        bottom[j] = d.v;
        s.v = last[j];
        for (int l = GSW - 1; l > 0; --l) d.f[l] = s.f[l - 1];
        d.f[0] = 0.0f;
        top[j] = d.v;
    }
}

/* CpuGrid (compute/shared/src/cpu.rs:162-165): input views include the 1-wide halo. */
typedef struct {
    const vf *in_u, *in_v;
    vf *out_u, *out_v;
    size_t out_rows, out_cols, stride;
} grid_t;

static inline size_t grid_len(const grid_t *g) /* cpu.rs:92-96 */
{
    return 2 * (g->out_rows + 2) * (g->out_cols + 2) + 2 * g->out_rows * g->out_cols;
}
static inline size_t grid_line_len(const grid_t *g) /* cpu.rs:101-105 */
{
    return 6 * (g->out_cols + 2) + 2 * g->out_cols;
}

/* split_grid (cpu.rs:111-154); axis < 0 = longest output axis.  Iterator::max_by_key keeps
 * the LAST maximum and the axes are enumerated rows first, so a tie picks the columns. */
static void split_grid(const grid_t *g, int axis, grid_t out[2])
{
    if (axis < 0) axis = (g->out_cols >= g->out_rows) ? 1 : 0;
    out[0] = out[1] = *g;
    if (axis == 0) {
        size_t sp = g->out_rows / 2;
        out[0].out_rows = sp;
        out[1].out_rows = g->out_rows - sp;
        out[1].in_u += sp * g->stride;
        out[1].in_v += sp * g->stride;
        out[1].out_u += sp * g->stride;
        out[1].out_v += sp * g->stride;
    } else {
        size_t sp = g->out_cols / 2;
        out[0].out_cols = sp;
        out[1].out_cols = g->out_cols - sp;
        out[1].in_u += sp;
        out[1].in_v += sp;
        out[1].out_u += sp;
        out[1].out_v += sp;
    }
}

/* compute_autovec::Simulation::unchecked_step_impl, compute/autovec/src/lib.rs:63-115. */
static void autovec_step(const gs_par_sim *s, const grid_t *g)
{
    float cw[3][3];
    float sum = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            cw[i][j] = s->p.w[i][j];
            sum += s->p.w[i][j]; /* into_iter().flatten().sum(), parameters.rs:60-61 */
        }
    cw[1][1] -= sum;
    vf w[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) w[i][j] = V_SET1(cw[i][j]);
    const vf Du = V_SET1(s->p.du), Dv = V_SET1(s->p.dv), F = V_SET1(s->p.feed);
    const vf mfk = V_SET1(-(s->p.feed + s->p.kill)); /* min_feed_kill, parameters.rs:67-69 */
    const vf dt = V_SET1(s->p.dt), ones = V_SET1(1.0f);
    const size_t st = g->stride;
    for (size_t i = 0; i < g->out_rows; ++i) {
        const vf *u0 = g->in_u + i * st, *u1 = u0 + st, *u2 = u1 + st;
        const vf *v0 = g->in_v + i * st, *v1 = v0 + st, *v2 = v1 + st;
        vf *ou = g->out_u + i * st, *ov = g->out_v + i * st;
        for (size_t j = 0; j < g->out_cols; ++j) {
            const vf u = u1[j + 1], v = v1[j + 1];
            vf fu1 = V_MUL(u0[j], w[0][0]), fv1 = V_MUL(v0[j], w[0][0]);
            vf fu2 = V_MUL(u0[j + 1], w[0][1]), fv2 = V_MUL(v0[j + 1], w[0][1]);
            vf fu3 = V_MUL(u0[j + 2], w[0][2]), fv3 = V_MUL(v0[j + 2], w[0][2]);
            fu1 = V_FMA(u1[j], w[1][0], fu1);
            fv1 = V_FMA(v1[j], w[1][0], fv1);
            fu2 = V_FMA(u1[j + 1], w[1][1], fu2);
            fv2 = V_FMA(v1[j + 1], w[1][1], fv2);
            fu3 = V_FMA(u1[j + 2], w[1][2], fu3);
            fv3 = V_FMA(v1[j + 2], w[1][2], fv3);
            fu1 = V_FMA(u2[j], w[2][0], fu1);
            fv1 = V_FMA(v2[j], w[2][0], fv1);
            fu2 = V_FMA(u2[j + 1], w[2][1], fu2);
            fv2 = V_FMA(v2[j + 1], w[2][1], fv2);
            fu3 = V_FMA(u2[j + 2], w[2][2], fu3);
            fv3 = V_FMA(v2[j + 2], w[2][2], fv3);
            const vf full_u = V_ADD(V_ADD(fu1, fu2), fu3);
            const vf full_v = V_ADD(V_ADD(fv1, fv2), fv3);
            const vf uvv = V_MUL(V_MUL(u, v), v);
            const vf du = V_ADD(V_SUB(V_MUL(Du, full_u), uvv), V_MUL(F, V_SUB(ones, u)));
            const vf dv = V_ADD(V_ADD(V_MUL(Dv, full_v), uvv), V_MUL(mfk, v));
            ou[j] = V_ADD(u, V_MUL(du, dt));
            ov[j] = V_ADD(v, V_MUL(dv, dt));
        }
    }
}

/* BlockWiseSimulation::unchecked_step_impl, compute/block/src/lib.rs:92-111. */
static void block_step(const gs_par_sim *s, const grid_t *g)
{
    if (g->out_rows == 0 || g->out_cols == 0) return;
    if (grid_line_len(g) <= s->max_values_per_line || g->out_cols == 1) {
        autovec_step(s, g);
    } else {
        grid_t h[2];
        split_grid(g, grid_len(g) > s->max_values_per_block ? -1 : 1, h);
        block_step(s, &h[0]);
        block_step(s, &h[1]);
    }
}

/* ParallelSimulation::unchecked_step_impl, compute/parallel/src/lib.rs:100-120: the leaves of
 * rayon::iter::split -- a sub-grid is a leaf when grid_len <= the sequential threshold (or it is a
 * single element), else it is bisected along the rows while more than one row is left, then along
 * the columns.  Leaves are rectangles of the output grid, listed top to bottom, left to right. */
typedef struct {
    size_t r0, rows, c0, cols;
} leaf_t;

static int collect_leaves(const gs_par_sim *s, leaf_t g, leaf_t **list, size_t *n, size_t *cap)
{
    const grid_t probe = {NULL, NULL, NULL, NULL, g.rows, g.cols, 0};
    if (grid_len(&probe) <= s->seq_len_threshold || (g.rows <= 1 && g.cols <= 1)) {
        if (*n == *cap) {
            size_t ncap = *cap ? 2 * *cap : 1024;
            leaf_t *nl = realloc(*list, ncap * sizeof **list);
            if (!nl) return -1;
            *list = nl;
            *cap = ncap;
        }
        (*list)[(*n)++] = g;
        return 0;
    }
    leaf_t a = g, b = g;
    if (g.rows > 1) { /* split_grid(axis 0), cpu.rs:111-154 */
        a.rows = g.rows / 2;
        b.r0 = g.r0 + a.rows;
        b.rows = g.rows - a.rows;
    } else {
        a.cols = g.cols / 2;
        b.c0 = g.c0 + a.cols;
        b.cols = g.cols - a.cols;
    }
    return collect_leaves(s, a, list, n, cap) || collect_leaves(s, b, list, n, cap);
}

/* ---- public C entry points (bound from Python with ctypes) ------------------------- */
void gs_par_destroy(gs_par_sim *s);

/* SimulateCreate::new for parallel(block(autovec)): block sizes in BYTES as the reference's
 * CLI takes them (block/src/args.rs:65-108, parallel/src/args.rs:10-23); defaults are the
 * caller's job (per-thread L1d/2, L2/2 -- parallel/src/block.rs:20-38). */
gs_par_sim *gs_par_create(const gs_par_params *p, size_t rows, size_t cols, size_t l1_block_bytes,
                          size_t l2_block_bytes, size_t seq_block_bytes, int nthreads, int ftz)
{
    gs_par_sim *s = calloc(1, sizeof *s);
    if (!s) return NULL;
    s->p = *p;
    s->max_values_per_line = l1_block_bytes / sizeof(vf);
    s->max_values_per_block = (l2_block_bytes / sizeof(vf)) / 2;
    s->seq_len_threshold = seq_block_bytes / sizeof(vf);
#ifdef _OPENMP
    s->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    s->nthreads = 1;
#endif
    s->ftz = ftz;
    /* Species::new (concentration/mod.rs:36-59): slot 0 = default(), slot 1 = ones/zeros,
     * seed written into slot 1, then flip() = finalize(slot 1); swap. */
    const float init[4] = {0.0f, 1.0f, 0.0f, 0.0f};
    for (int k = 0; k < 4; ++k)
        if (conc_alloc(&s->c[k], rows, cols, init[k])) {
            for (int m = 0; m < k; ++m) free(s->c[m].simd);
            free(s);
            return NULL;
        }
    size_t r0 = rows * 7 / 16, r1 = rows * 8 / 16;
    r0 = r0 > 4 ? r0 - 4 : 0;
    r1 = r1 > 4 ? r1 - 4 : 0;
    conc_fill_slice(&s->c[1], r0, r1, cols * 7 / 16, cols * 8 / 16, 0.0f);
    conc_fill_slice(&s->c[3], r0, r1, cols * 7 / 16, cols * 8 / 16, 1.0f);
    conc_finalize(&s->c[1], 0, cols);
    conc_finalize(&s->c[3], 0, cols);
    s->in = 1;
    leaf_t *list = NULL;
    size_t n = 0, cap = 0;
    const leaf_t whole = {0, s->c[0].L, 0, cols};
    if (collect_leaves(s, whole, &list, &n, &cap)) {
        free(list);
        gs_par_destroy(s);
        return NULL;
    }
    s->leaves = list;
    s->n_leaves = n;
    return s;
}

void gs_par_destroy(gs_par_sim *s)
{
    if (!s) return;
    for (int k = 0; k < 4; ++k) free(s->c[k].simd);
    free(s->leaves);
    free(s);
}

/* perform_steps (compute/shared/src/cpu.rs:30-42): step; flip (= finalize output; swap).  One
 * team for the whole call; per step every thread runs its share of the leaves through the block
 * recursion, then (barrier) rebuilds its share of the halo rows, then (barrier) the slots swap. */
void gs_par_perform_steps(gs_par_sim *s, size_t steps)
{
    const leaf_t *leaves = s->leaves;
    const int in0 = s->in;
#pragma omp parallel num_threads(s->nthreads)
    {
        const int was = set_ftz(s->ftz); /* every worker, see the header */
#ifdef _OPENMP
        const size_t t = (size_t)omp_get_thread_num(), T = (size_t)omp_get_num_threads();
#else
        const size_t t = 0, T = 1;
#endif
        const size_t l0 = s->n_leaves * t / T, l1 = s->n_leaves * (t + 1) / T;
        const size_t j0 = s->c[0].cols * t / T, j1 = s->c[0].cols * (t + 1) / T;
        int in = in0;
        for (size_t n = 0; n < steps; ++n) {
            gs_par_conc *iu = &s->c[in], *ou = &s->c[1 - in];
            gs_par_conc *iv = &s->c[2 + in], *ov = &s->c[3 - in];
            const size_t st = iu->stride;
            for (size_t k = l0; k < l1; ++k) {
                const leaf_t *l = &leaves[k];
                const size_t off = l->r0 * st + l->c0;
                const grid_t g = {iu->simd + off, iv->simd + off, ou->simd + st + 1 + off, ov->simd + st + 1 + off,
                                  l->rows, l->cols, st};
                block_step(s, &g);
            }
#pragma omp barrier
            conc_finalize(ou, j0, j1);
            conc_finalize(ov, j0, j1);
#pragma omp barrier
            in = 1 - in;
        }
        set_ftz(was);
    }
    s->in = (int)((in0 + steps) & 1);
}

/* write_scalar_view of species k (0 = U, 1 = V) input slot into a dense [rows, cols] array
 * (simd/mod.rs:132-190, without the SIMD transpose trick -- not timed). */
void gs_par_read(const gs_par_sim *s, int species, float *out)
{
    const gs_par_conc *c = &s->c[2 * species + s->in];
    for (size_t i = 0; i < c->L; ++i)
        for (size_t j = 0; j < c->cols; ++j) {
            vlanes t;
            t.v = c->simd[(i + 1) * c->stride + 1 + j];
            for (int l = 0; l < GSW; ++l) out[(l * c->L + i) * c->cols + j] = t.f[l];
        }
}
