/*
 * glabc_oracle.c -- CPU restatement of the reference's hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; nothing under
 * gl-abc-mcmc_amd/ links, imports or calls it.
 *
 * What it is: a scalar, one-chain-at-a-time, plain-C restatement of the
 * reference's sampler loops, following the reference's float32 operation order
 * line by line (citations on each function, paths relative to /root/reference).
 * It is pinned against the reference itself: tests/golden/make_golden.py
 * imports the reference's Python modules in the build container, replays a
 * recorded random-number tape through the UNMODIFIED reference loops, and
 * stores tape + resulting chains under tests/golden/; tests/test_oracle_golden.py
 * replays the same tapes through this file and requires bit-identical chains.
 *
 * Random numbers come either from a tape (glabc_run.tape, the pinning mode)
 * or from the Philox stream specified in include/glabc_numerics.h (the mode in
 * which the gfx950 kernels are compared against this file, bit for bit).
 *
 * The elementary functions exp/log are those of include/glabc_numerics.h (the
 * numerical specification shared with the device code), not libm and not
 * ATen's vectorised ones; they differ from ATen's in the last bit for ~1 % of
 * arguments, which can only change a chain through an accept/resample
 * near-tie (|log u - log_acc| within an ulp): none occurs in the goldens.
 *
 * Build: oracle/Makefile  ->  oracle/libglabc_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/glabc.h"
#include "../include/glabc_numerics.h"

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* ATen's float32 sum over a contiguous row of n elements, as this torch build
 * (2.10, CPU) associates it -- probed in the build container for n up to 20 000, see
 * DESIGN.md "row-sum order"; aten/src/ATen/native/cpu/SumKernel.cpp is the published
 * algorithm.  n < 8: four scalar lanes, leftovers into lane 0, lanes combined left to
 * right.  n >= 8: 8-wide vectors dealt to four accumulators (row_sum, ilp_factor 4: the
 * vectors of the full groups of four go to accumulator v % 4, leftover vectors to
 * accumulator 0), each accumulator a four-level cascade (multi_row_sum: every 16 additions
 * level 0 is folded into level 1, every 256 level 1 into level 2, ...), accumulators
 * combined left to right, then a scalar accumulator takes the n%8 tail in order and finally
 * the 8 vector partials in order.  Matters because the reference normalises the N+1 iSIR
 * weights with torch.sum (GLMCMC.py:82). */
static int ceil_log2_i64(int64_t x)                 /* c10 utils::CeilLog2 */
{
    if (x <= 2) return 1;
    int b = 0;
    for (uint64_t v = (uint64_t)(x - 1); v; v >>= 1) ++b;
    return b;
}

/* multi_row_sum<float, 4>: rows = groups of four 8-wide vectors; result in out[4][8] */
static void aten_multi_row_sum(const float* x, int64_t groups, float out[4][8])
{
    float acc[4][4][8];
    memset(acc, 0, sizeof acc);
    int64_t level_power = ceil_log2_i64(groups) / 4;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power, level_mask = level_step - 1;
    int64_t i = 0;
    while (i + level_step <= groups) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int q = 0; q < 4; ++q)
                for (int k = 0; k < 8; ++k) acc[0][q][k] += x[8 * (4 * i + q) + k];
        for (int j = 1; j < 4; ++j) {
            for (int q = 0; q < 4; ++q)
                for (int k = 0; k < 8; ++k) {
                    acc[j][q][k] += acc[j - 1][q][k];
                    acc[j - 1][q][k] = 0.0f;
                }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < groups; ++i)
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < 8; ++k) acc[0][q][k] += x[8 * (4 * i + q) + k];
    for (int j = 1; j < 4; ++j)
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < 8; ++k) acc[0][q][k] += acc[j][q][k];
    memcpy(out, acc[0], sizeof acc[0]);
}

static float aten_rowsum_f32(const float* x, int64_t n)
{
    if (n <= 0) return 0.0f;
    if (n < 8) {
        int g = (int)n / 4;
        if (g == 0) {
            float s = x[0];
            for (int i = 1; i < n; ++i) s = s + x[i];
            return s;
        }
        float l0 = x[0], l1 = x[1], l2 = x[2], l3 = x[3];
        for (int i = 4; i < n; ++i) l0 = l0 + x[i];     /* g == 1 here: all leftovers go to lane 0 */
        return ((l0 + l1) + l2) + l3;
    }
    int64_t nv = n / 8, g = nv / 4;
    float l[4][8], acc[8];
    aten_multi_row_sum(x, g, l);
    for (int64_t v = 4 * g; v < nv; ++v)
        for (int k = 0; k < 8; ++k) l[0][k] = l[0][k] + x[8 * v + k];
    for (int k = 0; k < 8; ++k) acc[k] = ((l[0][k] + l[1][k]) + l[2][k]) + l[3][k];
    float fa = 0.0f;
    for (int64_t i = 8 * nv; i < n; ++i) fa = fa + x[i];
    for (int k = 0; k < 8; ++k) fa = fa + acc[k];
    return fa;
}

ORACLE_API float oracle_aten_rowsum_f32(const float* x, int n) { return aten_rowsum_f32(x, (int64_t)n); }

/* ------------------------------------------------------------------------- */
/* distribution.py */

/* DiagGaussian.log_prob, distribution.py:176-181:
 *   log_p = -0.5*d*log(2pi) - sum_j( log_scale_j + 0.5 * ((z_j - loc_j)/exp(log_scale_j))^2 ) */
static float diag_gauss_log_prob(const glabc_dist* g, const float* z)
{
    float t[GLABC_MAX_DIM];
    for (int j = 0; j < g->dim; ++j) {
        float e = (z[j] - g->p0[j]) / g->p2[j];
        t[j] = g->p1[j] + 0.5f * (e * e);
    }
    return g->c0 - aten_rowsum_f32(t, g->dim);
}

/* DiagGaussian.forward given its noise, distribution.py:166-174:
 *   z = loc + exp(log_scale)*eps ;  log_p = C - sum_j( log_scale_j + 0.5*eps_j^2 ) */
static float diag_gauss_forward(const glabc_dist* g, const float* eps, float* z)
{
    float t[GLABC_MAX_DIM];
    for (int j = 0; j < g->dim; ++j) {
        z[j] = g->p0[j] + g->p2[j] * eps[j];
        t[j] = g->p1[j] + 0.5f * (eps[j] * eps[j]);
    }
    return g->c0 - aten_rowsum_f32(t, g->dim);
}

/* Uniform.log_prob, distribution.py:81-86: constant, -inf outside the CLOSED box. */
static float uniform_log_prob(const glabc_dist* g, const float* z)
{
    for (int j = 0; j < g->dim; ++j)
        if (z[j] < g->p0[j] || z[j] > g->p1[j]) return -INFINITY;
    return g->c0;
}

/* Uniform.forward given its [0,1) draws, distribution.py:73-79: z = low + (high-low)*u. */
static float uniform_forward(const glabc_dist* g, const float* u, float* z)
{
    for (int j = 0; j < g->dim; ++j) z[j] = g->p0[j] + g->p2[j] * u[j];
    return g->c0;
}

static double aten_rowsum_f64(const double* x, int n);

/* Gamma.log_prob, distribution.py:123-137, at a float32 point (include/glabc.h, GLABC_DIST_GAMMA inside the samplers): float64 per
 * coordinate -- log(scipy.stats.gamma.pdf), -inf where the pdf is 0 -- summed as torch.sum sums a float64 row, rounded once */
static float gamma_dist_log_prob(const glabc_dist* g, const float* z)
{
    double t[GLABC_MAX_DIM];
    for (int j = 0; j < g->dim; ++j) t[j] = glabc_gamma_log_pdf((double)g->p0[j], (double)g->p2[j], (double)g->p3[j], (double)z[j]);
    return (float)aten_rowsum_f64(t, g->dim);
}

/* Gamma.forward, distribution.py:106-121, for candidate j of (chain, step): the double variates from the chain's Gamma slots
 * (glabc_gamma_draw_candidate), theta' = (float) z, log_p = (float) log_prob of the DOUBLE variate (:120) */
static float gamma_dist_forward(const glabc_dist* g, uint64_t seed, uint64_t chain, uint32_t step, int j, float* z)
{
    double t[GLABC_MAX_DIM];
    for (int q = 0; q < g->dim; ++q) {
        const double v = glabc_gamma_draw_candidate((double)g->p0[q], (uint32_t)chain, (uint32_t)(chain >> 32), step, j, q,
                                                    (uint32_t)seed, (uint32_t)(seed >> 32)) * (double)g->p2[q];   /* gamma.rvs(shape, scale=1/rate) */
        z[q] = (float)v;
        t[q] = glabc_gamma_log_pdf((double)g->p0[q], (double)g->p2[q], (double)g->p3[q], v);
    }
    return (float)aten_rowsum_f64(t, g->dim);
}

/* test hook: the DOUBLE variates and log-densities behind gamma_dist_forward for candidates 0 .. n_prop-1 of (chain, step) --
 * what the reference's Gamma.forward returns when its scipy draw is replaced by these variates (tests/golden/make_golden.py
 * gamma_candidates_fixture).  z[n_prop][dim], log_p[n_prop]. */
ORACLE_API int oracle_gamma_candidates(const glabc_dist* g, uint64_t seed, uint64_t chain, uint32_t step, int n_prop, double* z,
                                       double* log_p)
{
    if (!g || !z || !log_p) return GLABC_ERR_NULL;
    if (g->kind != GLABC_DIST_GAMMA) return GLABC_ERR_KIND;
    if (g->dim < 1 || g->dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    for (int j = 0; j < n_prop; ++j) {
        double t[GLABC_MAX_DIM];
        for (int q = 0; q < g->dim; ++q) {
            const double v = glabc_gamma_draw_candidate((double)g->p0[q], (uint32_t)chain, (uint32_t)(chain >> 32), step, j, q,
                                                        (uint32_t)seed, (uint32_t)(seed >> 32)) * (double)g->p2[q];
            z[j * g->dim + q] = v;
            t[q] = glabc_gamma_log_pdf((double)g->p0[q], (double)g->p2[q], (double)g->p3[q], v);
        }
        log_p[j] = aten_rowsum_f64(t, g->dim);
    }
    return 0;
}

static int dist_log_prob(const glabc_dist* g, const float* z, float* out)
{
    switch (g->kind) {
    case GLABC_DIST_DIAG_GAUSS: *out = diag_gauss_log_prob(g, z); return 0;
    case GLABC_DIST_UNIFORM: *out = uniform_log_prob(g, z); return 0;
    case GLABC_DIST_GAMMA: *out = gamma_dist_log_prob(g, z); return 0;
    default: return GLABC_ERR_KIND;
    }
}

ORACLE_API int oracle_dist_log_prob(const glabc_dist* dist, const float* z, int64_t n, float* out)
{
    if (!dist || !z || !out) return GLABC_ERR_NULL;
    if (dist->dim < 1 || dist->dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    for (int64_t i = 0; i < n; ++i) {
        int rc = dist_log_prob(dist, z + i * dist->dim, out + i);
        if (rc) return rc;
    }
    return 0;
}

/* Gamma.log_prob, distribution.py:123-137 (float64; scipy.stats.gamma.pdf restated: exp(xlogy(a-1, x) - x - gammaln(a)) / scale) */
ORACLE_API int oracle_gamma_log_prob(const glabc_gamma* g, const double* z, int64_t n, double* out)
{
    if (!g || !z || !out) return GLABC_ERR_NULL;
    if (g->dim < 1 || g->dim > 3) return GLABC_ERR_DIM;
    for (int64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < g->dim; ++j) {
            double x = z[i * g->dim + j] / g->scale[j], lp;
            if (x >= 0.0) {                                /* scipy's support of gamma is closed at 0 */
                double am1 = g->shape[j] - 1.0;
                double xl = am1 == 0.0 ? 0.0 : am1 * glabc_log(x);
                double p = glabc_exp((xl - x) - g->gammaln[j]) / g->scale[j];
                lp = p > 0.0 ? glabc_log(p) : -INFINITY;
            } else {
                lp = -INFINITY;
            }
            acc = j == 0 ? lp : acc + lp;
        }
        out[i] = acc;
    }
    return 0;
}

/* Gamma.forward, distribution.py:106-121, with the device entry point's draws (include/glabc.h, glabc_gamma_forward) */
ORACLE_API int oracle_gamma_forward(const glabc_gamma* g, int64_t n, uint64_t seed, int64_t row0, double* z, double* log_p)
{
    if (!g || !z || !log_p) return GLABC_ERR_NULL;
    if (g->dim < 1 || g->dim > 3) return GLABC_ERR_DIM;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t gid = (uint64_t)(row0 + i);
        for (int j = 0; j < g->dim; ++j)
            z[i * g->dim + j] = glabc_gamma_draw(g->shape[j], (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)j, (uint32_t)seed,
                                                 (uint32_t)(seed >> 32)) * g->scale[j];       /* gamma.rvs(shape, scale=1/rate), :118 */
    }
    return oracle_gamma_log_prob(g, z, n, log_p);                                               /* :120 */
}

/* forward() with the noise supplied by the caller: noise[n][dim] -> z[n][dim], log_p[n]. */
ORACLE_API int oracle_dist_forward(const glabc_dist* dist, const float* noise, int64_t n, float* z, float* log_p)
{
    if (!dist || !noise || !z || !log_p) return GLABC_ERR_NULL;
    if (dist->dim < 1 || dist->dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    for (int64_t i = 0; i < n; ++i) {
        if (dist->kind == GLABC_DIST_DIAG_GAUSS)
            log_p[i] = diag_gauss_forward(dist, noise + i * dist->dim, z + i * dist->dim);
        else if (dist->kind == GLABC_DIST_UNIFORM)
            log_p[i] = uniform_forward(dist, noise + i * dist->dim, z + i * dist->dim);
        else
            return GLABC_ERR_KIND;
    }
    return 0;
}

/* forward() with the device entry point's Philox draws (include/glabc.h, glabc_dist_forward); z is [dim][n]. */
ORACLE_API int oracle_dist_forward_philox(const glabc_dist* dist, int64_t n, uint64_t seed, int64_t row0, float* z, float* log_p)
{
    if (!dist || !z || !log_p) return GLABC_ERR_NULL;
    if (dist->dim < 1 || dist->dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (dist->kind != GLABC_DIST_DIAG_GAUSS && dist->kind != GLABC_DIST_UNIFORM) return GLABC_ERR_KIND;
    int D = dist->dim;
    for (int64_t r = 0; r < n; ++r) {
        uint64_t gid = (uint64_t)(row0 + r);
        float e[GLABC_MAX_DIM + 4], zz[GLABC_MAX_DIM];
        for (int b = 0; b < (D + 3) / 4; ++b) {
            glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, (uint32_t)seed,
                                                (uint32_t)(seed >> 32));
            if (dist->kind == GLABC_DIST_UNIFORM) {
                for (int q = 0; q < 4; ++q) e[4 * b + q] = glabc_uniform_f32(w.v[q]);
            } else {
                glabc_normal_pair(w.v[0], w.v[1], &e[4 * b], &e[4 * b + 1]);
                glabc_normal_pair(w.v[2], w.v[3], &e[4 * b + 2], &e[4 * b + 3]);
            }
        }
        log_p[r] = dist->kind == GLABC_DIST_UNIFORM ? uniform_forward(dist, e, zz) : diag_gauss_forward(dist, e, zz);
        for (int j = 0; j < D; ++j) z[j * n + r] = zz[j];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* examples/Mixture.py -- the Model callbacks */

/* tanh through the specified exp (the g-and-k quantile function below) */
static float gk_tanhf(float x)
{
    float e = glabc_expf(-2.0f * fabsf(x));
    float r = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -r : r;
}

/* generate_samples for one theta and one simulation.
 *   GLABC_SIM_ABS_GAUSS, Mixture.py:19-23:  y = |theta| + (loc + exp(log_scale)*eps)
 *   GLABC_SIM_GK (the build's own Model, no counterpart in the reference tree; glabcmcmc_amd/examples/GK.py):
 *       y_j = A + B (1 + c tanh(g z_j/2)) (1 + z_j^2)^k z_j,  theta = (A, B, g, k), then sorted ascending */
/* GLABC_SIM_USER: the caller's simulator (include/glabc.h, glabc_rtc_compile) built by the host compiler from the same
 * source and registered here by the tests */
typedef void (*oracle_user_sim_fn)(const float* theta, const float* eps, float* y);
static oracle_user_sim_fn g_user_sim = NULL;
ORACLE_API void oracle_set_user_simulator(oracle_user_sim_fn fn) { g_user_sim = fn; }
/* ... and the Model's other callbacks as user source (include/glabc.h, "The Model's OTHER callbacks as user source"): host builds
 * of glabc_user_prior_log_prob / glabc_user_discrepancy / glabc_user_log_kernel, NULL = the descriptor's.  Used for
 * sim_kind == GLABC_SIM_USER only. */
typedef float (*oracle_user_prior_fn)(const float* theta);
typedef float (*oracle_user_dis_fn)(const float* y, const float* y_obs);
typedef float (*oracle_user_kern_fn)(float dis, float scale);
static oracle_user_prior_fn g_user_prior = NULL;
static oracle_user_dis_fn g_user_dis = NULL;
static oracle_user_kern_fn g_user_kern = NULL;
ORACLE_API void oracle_set_user_model(oracle_user_prior_fn prior, oracle_user_dis_fn dis, oracle_user_kern_fn kern)
{
    g_user_prior = prior;
    g_user_dis = dis;
    g_user_kern = kern;
}

/* standard normals one simulation consumes: y_dim for the built-in simulators, noise.dim for a user simulator */
static int model_noise_dim(const glabc_model* m) { return m->sim_kind == GLABC_SIM_USER ? m->noise.dim : m->y_dim; }

static void model_simulate(const glabc_model* m, const float* theta, const float* eps, float* y)
{
    if (m->sim_kind == GLABC_SIM_USER) {
        g_user_sim(theta, eps, y);
        return;
    }
    if (m->sim_kind == GLABC_SIM_GK) {
        for (int j = 0; j < m->y_dim; ++j) {
            float z = eps[j];
            float t = gk_tanhf((theta[2] * z) * 0.5f);
            float pw = glabc_expf(theta[3] * glabc_logf(1.0f + z * z));
            y[j] = theta[0] + ((theta[1] * (1.0f + m->gk_c * t)) * pw) * z;
        }
        for (int a = 1; a < m->y_dim; ++a) {                    /* insertion sort: any correct sort gives the same array */
            float v = y[a];
            int b = a - 1;
            while (b >= 0 && y[b] > v) { y[b + 1] = y[b]; --b; }
            y[b + 1] = v;
        }
        return;
    }
    for (int j = 0; j < m->y_dim; ++j) {
        float noise = m->noise.p0[j] + m->noise.p2[j] * eps[j];
        y[j] = fabsf(theta[j]) + noise;
    }
}

/* prior_log_prob, Mixture.py:28-31 */
static float model_prior(const glabc_model* m, const float* theta)
{
    if (m->sim_kind == GLABC_SIM_USER && g_user_prior) return g_user_prior(theta);
    float v = 0.0f;
    dist_log_prob(&m->prior, theta, &v);
    return v;
}

/* discrepancy, Mixture.py:33-36: sqrt(sum_j (y_j - y_obs_j)^2) */
static float model_discrepancy(const glabc_model* m, const float* y)
{
    if (m->sim_kind == GLABC_SIM_USER && g_user_dis) return g_user_dis(y, m->y_obs);
    float t[GLABC_MAX_DIM];
    for (int j = 0; j < m->y_dim; ++j) {
        float d = y[j] - m->y_obs[j];
        t[j] = d * d;
    }
    return sqrtf(aten_rowsum_f32(t, m->y_dim));
}

/* calculate_log_kernel, Mixture.py:38-45: DiagGaussian(1, 0, log eps).log_prob(dis) */
static float model_log_kernel_dis(const glabc_model* m, float dis)
{
    if (m->sim_kind == GLABC_SIM_USER && g_user_kern) return g_user_kern(dis, m->kern_scale);
    float e = (dis - 0.0f) / m->kern_scale;
    return m->kern_c0 - (m->kern_log_scale + 0.5f * (e * e));
}

static float model_log_kernel(const glabc_model* m, const float* y)
{
    return model_log_kernel_dis(m, model_discrepancy(m, y));
}

static int model_check(const glabc_model* m)
{
    if (!m) return GLABC_ERR_NULL;
    if (m->sim_kind != GLABC_SIM_ABS_GAUSS && m->sim_kind != GLABC_SIM_GK && m->sim_kind != GLABC_SIM_USER) return GLABC_ERR_KIND;
    if (m->theta_dim < 1 || m->theta_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (m->y_dim < 1 || m->y_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (m->prior.dim != m->theta_dim) return GLABC_ERR_DIM;
    if (m->sim_kind == GLABC_SIM_USER)
        return (g_user_sim && m->noise.dim >= 1 && m->noise.dim <= GLABC_MAX_DIM) ? 0 : GLABC_ERR_KIND;
    if (m->sim_kind == GLABC_SIM_GK) return m->theta_dim == 4 ? 0 : GLABC_ERR_DIM;
    if (m->y_dim != m->theta_dim) return GLABC_ERR_DIM;        /* |theta| + noise is elementwise */
    if (m->noise.dim != m->y_dim) return GLABC_ERR_DIM;
    if (m->noise.kind != GLABC_DIST_DIAG_GAUSS) return GLABC_ERR_KIND;
    return 0;
}

ORACLE_API int oracle_model_prior_log_prob(const glabc_model* m, const float* theta, int64_t n, float* out)
{
    int rc = model_check(m);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) out[i] = model_prior(m, theta + i * m->theta_dim);
    return 0;
}

ORACLE_API int oracle_model_discrepancy(const glabc_model* m, const float* y, int64_t n, float* out)
{
    int rc = model_check(m);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) out[i] = model_discrepancy(m, y + i * m->y_dim);
    return 0;
}

ORACLE_API int oracle_model_log_kernel(const glabc_model* m, const float* y, int64_t n, float* out)
{
    int rc = model_check(m);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) out[i] = model_log_kernel(m, y + i * m->y_dim);
    return 0;
}

/* generate_samples with the noise supplied: theta[n][d], eps[n][y_dim] -> y[n][y_dim] */
ORACLE_API int oracle_model_simulate(const glabc_model* m, const float* theta, const float* eps, int64_t n, float* y)
{
    int rc = model_check(m);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) model_simulate(m, theta + i * m->theta_dim, eps + i * m->y_dim, y + i * m->y_dim);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* the random draws of one iteration of one chain */

typedef struct step_draws {
    float u_branch;                                 /* torch.rand(1)            GLMCMC.py:59 */
    float u_accept;                                 /* torch.rand(1)            GLMCMC.py:98 */
    double u_resample;                              /* np.random.uniform(0,1)   GLMCMC.py:17 */
    uint64_t seed, chain;                           /* the (chain, step) the draws belong to: a Gamma proposal draws its */
    uint32_t step;                                  /* variates from the chain's own Gamma slots (gamma_dist_forward)     */
    int philox;                                     /* 0: replayed from a tape (no Gamma variates there) */
    float (*z)[2 * GLABC_MAX_DIM];                  /* per proposal: d proposal draws then y_dim simulator draws */
    float zs[GLABC_MAX_BATCH][2 * GLABC_MAX_DIM];   /* storage for up to GLABC_MAX_BATCH proposals; larger batches: heap */
} step_draws;

/* zeroed draws with room for n_prop proposals; draws_release frees what draws_init took from the heap */
static int draws_init(step_draws* s, int n_prop)
{
    memset(s, 0, sizeof *s);
    s->z = s->zs;
    if (n_prop > GLABC_MAX_BATCH) {
        s->z = calloc((size_t)n_prop, sizeof s->zs[0]);
        if (!s->z) return GLABC_ERR_ARG;
    }
    return 0;
}

static void draws_release(step_draws* s)
{
    if (s->z != s->zs) free(s->z);
}

/* Philox slots of one (chain, step): slot 0 = {branch, accept, resample hi, resample lo};
 * proposal j uses slots 1 + j*spp .. , spp = ceil((dp + y_dim)/4) blocks, dp = d rounded up to even:
 * words 0..d-1 are the proposal draws -- consecutive u32 pairs feeding Box-Muller (DiagGaussian
 * proposal) or single u32 -> [0,1) (Uniform proposal) -- and the simulator's normals start at word
 * dp, so that no simulator normal shares a word with a proposal draw (with odd d a Box-Muller pair
 * would otherwise straddle the boundary and y' would depend on a Uniform proposal's theta'). */
static void draws_from_philox(step_draws* s, uint64_t seed, uint64_t chain, uint32_t step, int n_prop, int d, int yd,
                              int prop_uniform)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)chain, c1 = (uint32_t)(chain >> 32);
    s->seed = seed;
    s->chain = chain;
    s->step = step;
    s->philox = 1;
    glabc_u32x4 h = glabc_philox4x32_10(c0, c1, step, 0u, k0, k1);
    s->u_branch = glabc_uniform_f32(h.v[0]);
    s->u_accept = glabc_uniform_f32(h.v[1]);
    s->u_resample = glabc_uniform_f64(h.v[2], h.v[3]);
    int dp = d + (d & 1);
    int m = dp + yd;
    int spp = (m + 3) / 4;
    for (int j = 0; j < n_prop; ++j) {
        uint32_t w[4 * ((2 * GLABC_MAX_DIM + 3) / 4)];
        for (int b = 0; b < spp; ++b) {
            glabc_u32x4 r = glabc_philox4x32_10(c0, c1, step, (uint32_t)(1 + j * spp + b), k0, k1);
            for (int q = 0; q < 4; ++q) w[4 * b + q] = r.v[q];
        }
        /* normals are made in pairs from words (2i, 2i+1) */
        float nrm[4 * ((2 * GLABC_MAX_DIM + 3) / 4)];
        for (int i = 0; 2 * i < 4 * spp; ++i) glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
        for (int i = 0; i < d; ++i) s->z[j][i] = nrm[i];
        for (int i = 0; i < yd; ++i) s->z[j][d + i] = nrm[dp + i];
        if (prop_uniform)
            for (int i = 0; i < d; ++i) s->z[j][i] = glabc_uniform_f32(w[i]);
    }
}

static void draws_from_tape(step_draws* s, const glabc_tape* t, int64_t chain_local, int64_t t_idx, int64_t n_steps,
                            int n_prop, int d, int yd)
{
    int64_t ct = chain_local * n_steps + t_idx;
    s->philox = 0;
    s->u_branch = t->u[2 * ct + 0];
    s->u_accept = t->u[2 * ct + 1];
    s->u_resample = t->r ? t->r[ct] : 0.0;
    int m = d + yd;
    for (int j = 0; j < n_prop && j < t->n_prop; ++j)
        for (int i = 0; i < m; ++i) s->z[j][i] = t->z[(ct * t->n_prop + j) * m + i];
}

/* ------------------------------------------------------------------------- */
/* a proposal (global) or increment (local) distribution applied to its noise */

static int prop_forward(const glabc_dist* g, const float* noise, float* z, float* log_p)
{
    if (g->kind == GLABC_DIST_DIAG_GAUSS) { *log_p = diag_gauss_forward(g, noise, z); return 0; }
    if (g->kind == GLABC_DIST_UNIFORM) { *log_p = uniform_forward(g, noise, z); return 0; }
    return GLABC_ERR_KIND;
}

/* candidate j of the iteration the draws belong to: a Gamma importance / global proposal ignores the noise row */
static int prop_forward_j(const glabc_dist* g, const step_draws* dr, int j, float* z, float* log_p)
{
    if (g->kind == GLABC_DIST_GAMMA) {
        if (!dr->philox) return GLABC_ERR_ARG;
        *log_p = gamma_dist_forward(g, dr->seed, dr->chain, dr->step, j, z);
        return 0;
    }
    return prop_forward(g, dr->z[j], z, log_p);
}

typedef struct chain_state {
    float theta[GLABC_MAX_DIM];
    float y[GLABC_MAX_DIM];
    float log_w;
    uint32_t flags;
    uint32_t n_moves;
} chain_state;

static void load_chain(chain_state* s, const glabc_chains* c, int64_t i, int d, int yd)
{
    for (int j = 0; j < d; ++j) s->theta[j] = c->theta[j * c->stride + i];
    for (int j = 0; j < yd; ++j) s->y[j] = c->y[j * c->stride + i];
    s->log_w = c->log_w ? c->log_w[i] : 0.0f;
    s->flags = c->flags ? c->flags[i] : 0u;
    s->n_moves = c->n_moves ? c->n_moves[i] : 0u;
}

static void store_chain(const chain_state* s, const glabc_chains* c, int64_t i, int d, int yd)
{
    for (int j = 0; j < d; ++j) c->theta[j * c->stride + i] = s->theta[j];
    for (int j = 0; j < yd; ++j) c->y[j * c->stride + i] = s->y[j];
    if (c->log_w) c->log_w[i] = s->log_w;
    if (c->flags) c->flags[i] = s->flags;
    if (c->n_moves) c->n_moves[i] = s->n_moves;
}

static void record(const glabc_run* run, const glabc_chains* c, int64_t i, int64_t t, int d, const float* theta,
                   const float* theta_prev)
{
    if (run->history)
        for (int j = 0; j < d; ++j) run->history[(t * d + j) * run->hist_stride + i] = theta[j];
    if (run->moments) {
        const glabc_moments* m = run->moments;
        int k = 0;
        for (int a = 0; a < d; ++a) {
            m->sum_theta[a * c->stride + i] += (double)theta[a];
            for (int b = a; b < d; ++b, ++k) {
                m->sum_outer[k * c->stride + i] += (double)theta[a] * (double)theta[b];
                double da = (double)theta[a] - (double)theta_prev[a];
                double db = (double)theta[b] - (double)theta_prev[b];
                m->sum_jump[k * c->stride + i] += da * db;
            }
        }
    }
}

static int run_check(const glabc_model* m, const glabc_dist* a, const glabc_dist* b, const glabc_chains* c,
                     const glabc_run* r)
{
    int rc = model_check(m);
    if (rc) return rc;
    if (!a || !b || !c || !r) return GLABC_ERR_NULL;
    if (!c->theta || !c->y) return GLABC_ERR_NULL;
    if (a->dim != m->theta_dim || b->dim != m->theta_dim) return GLABC_ERR_DIM;
    if (c->n_chains < 0 || c->stride < c->n_chains || r->n_steps < 0) return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    return 0;
}

/* the random-walk MH local move shared by GLMCMC.py:90-104 and GlobalMCMC.py:55-68:
 *   theta' = Local_Proposal.sample(1) + theta ; y' = simulate(theta')
 *   log_acc = prior(theta') + K(y') - prior(theta) - K(y)     (left to right)
 *   accept iff log(u) < log_acc */
static int local_move(const glabc_model* m, const glabc_dist* local, chain_state* s, const step_draws* dr)
{
    int d = m->theta_dim, yd = m->y_dim;
    float inc[GLABC_MAX_DIM], th[GLABC_MAX_DIM], y[GLABC_MAX_DIM], lq;
    prop_forward(local, dr->z[0], inc, &lq);
    for (int j = 0; j < d; ++j) th[j] = inc[j] + s->theta[j];
    model_simulate(m, th, dr->z[0] + d, y);
    float log_acc = ((model_prior(m, th) + model_log_kernel(m, y)) - model_prior(m, s->theta)) - model_log_kernel(m, s->y);
    float log_u = glabc_logf(dr->u_accept);
    if (log_u < log_acc) {
        memcpy(s->theta, th, sizeof(float) * d);
        memcpy(s->y, y, sizeof(float) * yd);
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* GLMCMC.py:24-137 */

/* log_weight_old, GLMCMC.py:53-55 / 62-64 */
static float isir_weight_of_state(const glabc_model* m, const glabc_dist* imp, const chain_state* s)
{
    float lq = 0.0f;
    dist_log_prob(imp, s->theta, &lq);
    return (model_prior(m, s->theta) + model_log_kernel(m, s->y)) - lq;
}

/* the iSIR global move, GLMCMC.py:60-88 (same body at GLMALA.py:152-179) */
static int isir_move(const glabc_model* m, const glabc_dist* imp, int N, chain_state* s, const step_draws* dr)
{
    int d = m->theta_dim, yd = m->y_dim;
    float th_s[GLABC_MAX_BATCH + 1][GLABC_MAX_DIM], y_s[GLABC_MAX_BATCH + 1][GLABC_MAX_DIM];
    float lw_s[GLABC_MAX_BATCH + 1], w_s[GLABC_MAX_BATCH + 1];
    float (*th)[GLABC_MAX_DIM] = th_s, (*y)[GLABC_MAX_DIM] = y_s, *lw = lw_s, *w = w_s;
    void* heap = NULL;
    if (N > GLABC_MAX_BATCH) {                      /* batches beyond the small arrays: one heap block */
        const size_t rows = (size_t)N + 1;
        heap = malloc(rows * (2 * sizeof th_s[0] + 2 * sizeof(float)));
        if (!heap) return 0;
        th = heap;
        y = th + rows;
        lw = (float*)(y + rows);
        w = lw + rows;
    }
    if (s->flags & GLABC_FLAG_LOCAL) s->log_w = isir_weight_of_state(m, imp, s);    /* :60-64 */
    s->flags &= ~GLABC_FLAG_LOCAL;                                                  /* :65 */
    memcpy(th[0], s->theta, sizeof(float) * d);
    memcpy(y[0], s->y, sizeof(float) * yd);
    lw[0] = s->log_w;
    int n = 1;
    for (int j = 0; j < N; ++j) {
        float lq;
        prop_forward_j(imp, dr, j, th[n], &lq);                                     /* :66 */
        int has_nan = 0;
        for (int k = 0; k < d; ++k) has_nan |= isnan(th[n][k]);
        if (has_nan) continue;                                                      /* :67-70 */
        model_simulate(m, th[n], dr->z[j] + d, y[n]);                               /* :71 */
        lw[n] = (model_prior(m, th[n]) + model_log_kernel(m, y[n])) - lq;           /* :72-74 */
        ++n;
    }
    for (int k = 0; k < n; ++k) {
        w[k] = glabc_expf(lw[k]);                                                   /* :78 */
        if (isnan(w[k])) w[k] = 0.0f;                                               /* :80-81 */
    }
    float tot = aten_rowsum_f32(w, n);                                              /* :82 */
    for (int k = 0; k < n; ++k) w[k] = w[k] / tot;
    /* weight_sampling, GLMCMC.py:7-22: python-float (double) running sum of the float32
     * weights, first j with ran < s; falls off the end -> None -> the chain stays (:84). */
    int ind = -1;
    double acc = 0.0;
    for (int k = 0; k < n; ++k) {
        acc += (double)w[k];
        if (dr->u_resample < acc) { ind = k; break; }
    }
    if (ind > 0) {                                                                  /* :84-88 */
        memcpy(s->theta, th[ind], sizeof(float) * d);
        memcpy(s->y, y[ind], sizeof(float) * yd);
        s->log_w = lw[ind];
    }
    free(heap);
    return ind > 0;
}

ORACLE_API int oracle_init_weights(const glabc_model* m, const glabc_dist* imp, const glabc_chains* c)
{
    int rc = model_check(m);
    if (rc) return rc;
    if (!imp || !c || !c->theta || !c->y || !c->log_w || !c->flags) return GLABC_ERR_NULL;
    for (int64_t i = 0; i < c->n_chains; ++i) {
        chain_state s;
        load_chain(&s, c, i, m->theta_dim, m->y_dim);
        s.log_w = isir_weight_of_state(m, imp, &s);
        s.flags |= GLABC_FLAG_LOCAL;                                                /* GLMCMC.py:50 */
        store_chain(&s, c, i, m->theta_dim, m->y_dim);
    }
    return 0;
}

ORACLE_API int oracle_glmcmc_steps(const glabc_model* m, const glabc_dist* local, const glabc_dist* imp,
                                   const glabc_chains* c, const glabc_run* run)
{
    int rc = run_check(m, local, imp, c, run);
    if (rc) return rc;
    if (!c->log_w || !c->flags) return GLABC_ERR_NULL;
    int N = run->batch_size;
    if (N < 1 || N > GLABC_MAX_BATCH_WIDE) return GLABC_ERR_ARG;
    if (N > GLABC_MAX_BATCH && run->tape) return GLABC_ERR_ARG;
    int d = m->theta_dim, yd = m->y_dim, nd = model_noise_dim(m);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < c->n_chains; ++i) {
        chain_state s;
        step_draws dr;
        if (draws_init(&dr, N)) continue;
        load_chain(&s, c, i, d, yd);
        for (int64_t t = 0; t < run->n_steps; ++t) {
            float prev[GLABC_MAX_DIM];
            memcpy(prev, s.theta, sizeof(float) * d);
            /* Both proposal kinds draw from the same slots; the global move's noise kind
             * follows the importance proposal, the local move's follows the local one. */
            if (run->tape)
                draws_from_tape(&dr, run->tape, i, t, run->n_steps, N, d, yd);
            else
                draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0 + (uint32_t)t, N, d, nd, 0);
            const float gf = run->global_frequency_per_chain ? run->global_frequency_per_chain[i] : run->global_frequency;
            if (dr.u_branch < gf) {                                                 /* GLMCMC.py:59 */
                if (!run->tape && imp->kind == GLABC_DIST_UNIFORM)
                    draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0 + (uint32_t)t, N, d, nd, 1);
                s.n_moves += (uint32_t)isir_move(m, imp, N, &s, &dr);
            } else {
                if (!run->tape && local->kind == GLABC_DIST_UNIFORM)
                    draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0 + (uint32_t)t, 1, d, nd, 1);
                if (local_move(m, local, &s, &dr)) {
                    s.flags |= GLABC_FLAG_LOCAL;                                    /* :100 */
                    s.n_moves += 1u;
                }
            }
            record(run, c, i, t, d, s.theta, prev);                                 /* :89,104 */
        }
        store_chain(&s, c, i, d, yd);
        draws_release(&dr);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* GlobalMCMC.py:6-98 */

/* independence MH global move, GlobalMCMC.py:39-53:
 *   log_acc = prior' + K' + q(theta) - q' - prior - K     (left to right) */
static int independence_move(const glabc_model* m, const glabc_dist* glob, chain_state* s, const step_draws* dr)
{
    int d = m->theta_dim, yd = m->y_dim;
    float th[GLABC_MAX_DIM], y[GLABC_MAX_DIM], lq_new, lq_old = 0.0f;
    prop_forward_j(glob, dr, 0, th, &lq_new);
    model_simulate(m, th, dr->z[0] + d, y);
    dist_log_prob(glob, s->theta, &lq_old);
    float log_acc = ((((model_prior(m, th) + model_log_kernel(m, y)) + lq_old) - lq_new) - model_prior(m, s->theta)) -
                    model_log_kernel(m, s->y);
    float log_u = glabc_logf(dr->u_accept);
    if (log_u < log_acc) {
        memcpy(s->theta, th, sizeof(float) * d);
        memcpy(s->y, y, sizeof(float) * yd);
        return 1;
    }
    return 0;
}

ORACLE_API int oracle_globalmcmc_steps(const glabc_model* m, const glabc_dist* local, const glabc_dist* glob,
                                       const glabc_chains* c, const glabc_run* run)
{
    int rc = run_check(m, local, glob, c, run);
    if (rc) return rc;
    int d = m->theta_dim, yd = m->y_dim, nd = model_noise_dim(m);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < c->n_chains; ++i) {
        chain_state s;
        step_draws dr;
        draws_init(&dr, GLABC_MAX_BATCH);
        load_chain(&s, c, i, d, yd);
        for (int64_t t = 0; t < run->n_steps; ++t) {
            float prev[GLABC_MAX_DIM];
            memcpy(prev, s.theta, sizeof(float) * d);
            if (run->tape)
                draws_from_tape(&dr, run->tape, i, t, run->n_steps, 1, d, yd);
            else
                draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0 + (uint32_t)t, 1, d, nd, 0);
            const float gf = run->global_frequency_per_chain ? run->global_frequency_per_chain[i] : run->global_frequency;
            int is_global = dr.u_branch < gf;                                       /* GlobalMCMC.py:39 */
            const glabc_dist* p = is_global ? glob : local;
            if (!run->tape && p->kind == GLABC_DIST_UNIFORM)
                draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0 + (uint32_t)t, 1, d, nd, 1);
            int moved = is_global ? independence_move(m, glob, &s, &dr) : local_move(m, local, &s, &dr);
            s.n_moves += (uint32_t)moved;
            record(run, c, i, t, d, s.theta, prev);                                 /* :53,68 */
        }
        store_chain(&s, c, i, d, yd);
    }
    return 0;
}


/* ========================================================================= */
/* GLMALA.py:8-230                                                            */
/* ========================================================================= */

/* ATen's float64 sum over a contiguous row (same scheme as float32 with 4-wide vectors;
 * probed on the reference's torch build): n < 4 sequential; else four lanes of 4-wide vectors,
 * leftover vectors into lane 0, lanes combined left to right, then a scalar accumulator takes
 * the n%4 tail in order followed by the 4 vector partials in order. */
static double aten_rowsum_f64(const double* x, int n)
{
    if (n <= 0) return 0.0;
    if (n < 4) {
        double s = x[0];
        for (int i = 1; i < n; ++i) s = s + x[i];
        return s;
    }
    int nv = n / 4, g = nv / 4;
    double acc[4];
    if (g == 0) {
        for (int k = 0; k < 4; ++k) acc[k] = x[k];
        for (int v = 1; v < nv; ++v)
            for (int k = 0; k < 4; ++k) acc[k] = acc[k] + x[4 * v + k];
    } else {
        double l[4][4];
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < 4; ++k) l[q][k] = x[4 * q + k];
        for (int i = 1; i < g; ++i)
            for (int q = 0; q < 4; ++q)
                for (int k = 0; k < 4; ++k) l[q][k] = l[q][k] + x[4 * (4 * i + q) + k];
        for (int v = 4 * g; v < nv; ++v)
            for (int k = 0; k < 4; ++k) l[0][k] = l[0][k] + x[4 * v + k];
        for (int k = 0; k < 4; ++k) acc[k] = ((l[0][k] + l[1][k]) + l[2][k]) + l[3][k];
    }
    double fa = 0.0;
    for (int i = 4 * nv; i < n; ++i) fa = fa + x[i];
    for (int k = 0; k < 4; ++k) fa = fa + acc[k];
    return fa;
}

ORACLE_API double oracle_aten_rowsum_f64(const double* x, int n) { return aten_rowsum_f64(x, n); }

/* distribution.py:176-181 / 81-86 evaluated on a float64 tensor: the float32 parameters are
 * promoted, the constant -0.5*d*log(2 pi) stays a float64 scalar; Uniform.log_prob builds its
 * result from float32 ones (distribution.py:82), so it stays a float32 value. */
static double dist_log_prob_f64(const glabc_dist* g, const double* z)
{
    if (g->kind == GLABC_DIST_DIAG_GAUSS) {
        double t[GLABC_MAX_DIM];
        for (int j = 0; j < g->dim; ++j) {
            double e = (z[j] - (double)g->p0[j]) / (double)g->p2[j];
            t[j] = (double)g->p1[j] + 0.5 * (e * e);
        }
        return (-0.5 * (double)g->dim * GLABC_LOG_2PI) - aten_rowsum_f64(t, g->dim);
    }
    for (int j = 0; j < g->dim; ++j)
        if (z[j] < (double)g->p0[j] || z[j] > (double)g->p1[j]) return -INFINITY;
    return (double)g->c0;
}

/* Mixture.py:33-45 on a float64 y */
static double model_log_kernel_f64(const glabc_model* m, const double* y)
{
    double t[GLABC_MAX_DIM];
    for (int j = 0; j < m->y_dim; ++j) {
        double d = y[j] - (double)m->y_obs[j];
        t[j] = d * d;
    }
    double dis = sqrt(aten_rowsum_f64(t, m->y_dim));
    double e = (dis - 0.0) / (double)m->kern_scale;
    return (-0.5 * 1.0 * GLABC_LOG_2PI) - ((double)m->kern_log_scale + 0.5 * (e * e));
}

typedef struct mala_state {
    double theta[GLABC_MAX_DIM];   /* float32-exact while !(flags & TH64) */
    double y[GLABC_MAX_DIM];
    double log_w;                  /* float32-exact while !(flags & LW64) */
    double grad[GLABC_MAX_DIM];
    uint32_t flags, n_moves;
} mala_state;

/* prior / kernel / importance density of the CURRENT state in the precision the reference's
 * tensors have at this point */
static double state_prior(const glabc_model* m, const mala_state* s)
{
    if (s->flags & GLABC_FLAG_TH64) return dist_log_prob_f64(&m->prior, s->theta);
    float th[GLABC_MAX_DIM];
    for (int j = 0; j < m->theta_dim; ++j) th[j] = (float)s->theta[j];
    return (double)model_prior(m, th);
}

static double state_kernel(const glabc_model* m, const mala_state* s)
{
    if (s->flags & GLABC_FLAG_TH64) return model_log_kernel_f64(m, s->y);
    float y[GLABC_MAX_DIM];
    for (int j = 0; j < m->y_dim; ++j) y[j] = (float)s->y[j];
    return (double)model_log_kernel(m, y);
}

/* the standard-normal noise of one gradient simulation: Philox slots beyond the candidates' --
 * GRAD_BASE + g*GRAD_STRIDE + block, normal index n = (k*num + s)*y_dim + j lives in block n/4,
 * Box-Muller pair (n%4)/2, element n%2.  g = 0: gradient at Theta_old (first local move only),
 * g = 1: gradient at the proposal.  The +d and -d simulations of a coordinate share their noise
 * (the reference reseeds torch with the same seed, GLMALA.py:76,80). */
#define GRAD_BASE 0x100000u
#define GRAD_STRIDE 0x80000u

static void grad_noise(uint64_t seed, uint64_t chain, uint32_t step, int g, int k, int s, int num, int yd, float* eps)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)chain, c1 = (uint32_t)(chain >> 32);
    for (int j = 0; j < yd; ++j) {
        int64_t n = ((int64_t)k * num + s) * yd + j;
        uint32_t slot = GRAD_BASE + (uint32_t)g * GRAD_STRIDE + (uint32_t)(n / 4);
        glabc_u32x4 r = glabc_philox4x32_10(c0, c1, step, slot, k0, k1);
        int p = (int)((n % 4) / 2);
        float z0, z1;
        glabc_normal_pair(r.v[2 * p], r.v[2 * p + 1], &z0, &z1);
        eps[j] = (n % 2) ? z1 : z0;
    }
}

ORACLE_API void oracle_grad_noise(uint64_t seed, uint64_t chain, uint32_t step, int g, int k, int num, int yd, float* out)
{
    for (int s = 0; s < num; ++s) grad_noise(seed, chain, step, g, k, s, num, yd, out + (int64_t)s * yd);
}

/* numberical_gradient_logABC, GLMALA.py:46-95, for one theta (already cast to float32, :62).
 * mean / unbiased variance of the num discrepancies come from exact fixed-point sums of the data shifted
 * by the noise-free discrepancy (glabc_fxsum, include/glabc_numerics.h; torch.mean / torch.var use their own
 * cascades; the results agree to ~1e-16 relative, far below anything a float32 chain value can see). */
static void numerical_gradient(const glabc_model* m, const glabc_mala* p, const float* theta, uint64_t seed,
                               uint64_t chain, uint32_t step, int g, double* grad)
{
    int d = m->theta_dim, yd = m->y_dim, num = p->num_grad;
    const float h = 0.1f;                                   /* d = 1e-1 as a float32 eye entry, :65 */
    const float hp = 0.00001f;                              /* :84 */
    for (int k = 0; k < d; ++k) {
        float tp[GLABC_MAX_DIM], tm[GLABC_MAX_DIM];
        for (int j = 0; j < d; ++j) {
            tp[j] = theta[j] + (j == k ? h : 0.0f);         /* :67 */
            tm[j] = theta[j] - (j == k ? h : 0.0f);         /* :68 */
        }
        /* centres of the shifted sums: the discrepancy of the noise-free simulation (eps = 0) */
        float zero[GLABC_MAX_DIM] = {0}, y0p[GLABC_MAX_DIM], y0m[GLABC_MAX_DIM];
        model_simulate(m, tp, zero, y0p);
        model_simulate(m, tm, zero, y0m);
        const double c_p = (double)model_discrepancy(m, y0p), c_m = (double)model_discrepancy(m, y0m);
        glabc_fxsum ap = {0, 0, 0}, am = {0, 0, 0};
        for (int s = 0; s < num; ++s) {
            float eps[GLABC_MAX_DIM], yp[GLABC_MAX_DIM], ym[GLABC_MAX_DIM];
            grad_noise(seed, chain, step, g, k, s, num, yd, eps);
            model_simulate(m, tp, eps, yp);                 /* :78 */
            model_simulate(m, tm, eps, ym);                 /* :82 (same noise) */
            glabc_fx_add(&ap, glabc_fx_quantize((double)model_discrepancy(m, yp) - c_p));
            glabc_fx_add(&am, glabc_fx_quantize((double)model_discrepancy(m, ym) - c_m));
        }
        double n = (double)num;
        double s1p = glabc_fx_sum1(&ap), s2p = glabc_fx_sum2(&ap), s1m = glabc_fx_sum1(&am), s2m = glabc_fx_sum2(&am);
        double mu_p = c_p + s1p / n, mu_m = c_m + s1m / n;                              /* :86-87 */
        double var_p = (s2p - (s1p * s1p) / n) / (n - 1.0), var_m = (s2m - (s1m * s1m) / n) / (n - 1.0);   /* :88-89 */
        double lp = (-0.5 * glabc_log(var_p + p->eps_sq)) - ((0.5 * (mu_p * mu_p)) / (var_p + p->eps_sq));   /* :90-91 */
        double lm = (-0.5 * glabc_log(var_m + p->eps_sq)) - ((0.5 * (mu_m * mu_m)) / (var_m + p->eps_sq));   /* :92-93 */
        double gll = (lp - lm) / 0.2;                                                   /* :94, 2*d */
        for (int j = 0; j < d; ++j) {
            tp[j] = theta[j] + (j == k ? hp : 0.0f);                                    /* :84 */
            tm[j] = theta[j] - (j == k ? hp : 0.0f);
        }
        float gp = (model_prior(m, tp) - model_prior(m, tm)) / 0.00002f;                /* :84-85, float32 */
        grad[k] = gll + (double)gp;                                                     /* :95 */
    }
}

ORACLE_API int oracle_numerical_gradient(const glabc_model* m, const glabc_mala* p, const float* theta, uint64_t seed,
                                         uint64_t chain, uint32_t step, int g, double* grad)
{
    int rc = model_check(m);
    if (rc) return rc;
    numerical_gradient(m, p, theta, seed, chain, step, g, grad);
    return 0;
}

/* the MALA local move, GLMALA.py:182-200 */
static int mala_move(const glabc_model* m, const glabc_mala* p, mala_state* s, const step_draws* dr, uint64_t seed,
                     uint64_t chain, uint32_t step)
{
    int d = m->theta_dim, yd = m->y_dim;
    float thf[GLABC_MAX_DIM];
    if (!(s->flags & GLABC_FLAG_HAS_GRAD)) {                                            /* :183-184 */
        for (int j = 0; j < d; ++j) thf[j] = (float)s->theta[j];
        numerical_gradient(m, p, thf, seed, chain, step, 0, s->grad);
        s->flags |= GLABC_FLAG_HAS_GRAD;
    }
    /* Local_proposal_forward, :25-44: z ~ DiagGaussian(d, [0.0], [0.0]); x = z*tau + theta + grad*tau**2/2 */
    float z[GLABC_MAX_DIM], t[GLABC_MAX_DIM];
    double x[GLABC_MAX_DIM];
    const float tauf = (float)p->tau;
    for (int j = 0; j < d; ++j) {
        z[j] = 0.0f + 1.0f * dr->z[0][j];                                               /* distribution.py:170 */
        t[j] = 0.0f + 0.5f * (dr->z[0][j] * dr->z[0][j]);                               /* :172 */
    }
    float log_pro = (float)(-0.5 * (double)d * GLABC_LOG_2PI) - aten_rowsum_f32(t, d);       /* distribution.py:171 */
    for (int j = 0; j < d; ++j) {
        float a = z[j] * tauf;
        double b = (s->flags & GLABC_FLAG_TH64) ? ((double)a + s->theta[j]) : (double)(a + (float)s->theta[j]);
        x[j] = b + (s->grad[j] * p->tau_sq) / 2.0;                                      /* :43 */
    }
    double gprop[GLABC_MAX_DIM];
    for (int j = 0; j < d; ++j) thf[j] = (float)x[j];                                   /* :62 */
    numerical_gradient(m, p, thf, seed, chain, step, 1, gprop);                         /* :187 */
    /* y = generate_samples(Theta_prop, 1)[0,], :188-189: |x| (float64) + float32 noise */
    double y[GLABC_MAX_DIM];
    for (int j = 0; j < yd; ++j) {
        float noise = m->noise.p0[j] + m->noise.p2[j] * dr->z[0][d + j];
        y[j] = fabs(x[j]) + (double)noise;
    }
    /* log_proposal(Theta_prop, grad_prop, Theta_old, tau), :97-116 */
    double tq[GLABC_MAX_DIM];
    for (int j = 0; j < d; ++j) {
        double arg = ((s->theta[j] - x[j]) - (gprop[j] * p->tau_sq) / 2.0) / p->tau;
        double e = (arg - 0.0) / 1.0;
        tq[j] = 0.0 + 0.5 * (e * e);
    }
    double lq_rev = (-0.5 * (double)d * GLABC_LOG_2PI) - aten_rowsum_f64(tq, d);
    double log_acc = dist_log_prob_f64(&m->prior, x) + model_log_kernel_f64(m, y);     /* :190 */
    log_acc = log_acc + lq_rev;                                                         /* :191 */
    log_acc = log_acc - state_prior(m, s);                                              /* :192 */
    log_acc = log_acc - state_kernel(m, s);
    log_acc = log_acc - (double)log_pro;                                                /* :193 */
    double log_u = (double)glabc_logf(dr->u_accept);                                    /* :194 */
    if (log_u < log_acc) {                                                              /* :195-199 */
        for (int j = 0; j < d; ++j) s->theta[j] = x[j];
        for (int j = 0; j < yd; ++j) s->y[j] = y[j];
        for (int j = 0; j < d; ++j) s->grad[j] = gprop[j];
        s->flags |= GLABC_FLAG_TH64;
        return 1;                                      /* NB: `local` is NOT set (GLMALA.py:195-199 vs GLMCMC.py:100) */
    }
    return 0;
}

/* the iSIR global move of GLMALA.py:151-180: GLMCMC's, except that the current state's slot and
 * the weights take the dtypes described at GLABC_FLAG_TH64 / GLABC_FLAG_LW64 */
static int mala_isir_move(const glabc_model* m, const glabc_dist* imp, int N, mala_state* s, const step_draws* dr)
{
    int d = m->theta_dim, yd = m->y_dim;
    float th[GLABC_MAX_BATCH][GLABC_MAX_DIM], y[GLABC_MAX_BATCH][GLABC_MAX_DIM], lw0[GLABC_MAX_BATCH];
    if (s->flags & GLABC_FLAG_LOCAL) {                                                  /* :152-156 */
        if (s->flags & GLABC_FLAG_TH64) {
            s->log_w = (state_prior(m, s) + state_kernel(m, s)) - dist_log_prob_f64(imp, s->theta);
            s->flags |= GLABC_FLAG_LW64;
        } else {
            float thf[GLABC_MAX_DIM], q = 0.0f;
            for (int j = 0; j < d; ++j) thf[j] = (float)s->theta[j];
            dist_log_prob(imp, thf, &q);
            s->log_w = (double)(((float)state_prior(m, s) + (float)state_kernel(m, s)) - q);
        }
    }
    s->flags &= ~GLABC_FLAG_LOCAL;                                                      /* :157 */
    for (int j = 0; j < N; ++j) {
        float lq;
        prop_forward(imp, dr->z[j], th[j], &lq);                                        /* :158 */
        model_simulate(m, th[j], dr->z[j] + d, y[j]);                                   /* :163 */
        lw0[j] = (model_prior(m, th[j]) + model_log_kernel(m, y[j])) - lq;              /* :164-165 */
    }
    int ind = -1;
    if (s->flags & GLABC_FLAG_LW64) {                                                   /* float64 weights */
        double w[GLABC_MAX_BATCH + 1];
        w[0] = glabc_exp(s->log_w);
        for (int j = 0; j < N; ++j) w[j + 1] = glabc_exp((double)lw0[j]);               /* :169 */
        for (int k = 0; k <= N; ++k)
            if (isnan(w[k])) w[k] = 0.0;                                                /* :171-172 */
        double tot = aten_rowsum_f64(w, N + 1);                                         /* :173 */
        double acc = 0.0;
        for (int k = 0; k <= N; ++k) {
            acc += w[k] / tot;
            if (dr->u_resample < acc) { ind = k; break; }                               /* :174 */
        }
    } else {
        float w[GLABC_MAX_BATCH + 1];
        w[0] = glabc_expf((float)s->log_w);
        for (int j = 0; j < N; ++j) w[j + 1] = glabc_expf(lw0[j]);
        for (int k = 0; k <= N; ++k)
            if (isnan(w[k])) w[k] = 0.0f;
        float tot = aten_rowsum_f32(w, N + 1);
        double acc = 0.0;
        for (int k = 0; k <= N; ++k) {
            acc += (double)(w[k] / tot);
            if (dr->u_resample < acc) { ind = k; break; }
        }
    }
    if (ind > 0) {                                                                      /* :175-179 */
        for (int j = 0; j < d; ++j) s->theta[j] = (double)th[ind - 1][j];
        for (int j = 0; j < yd; ++j) s->y[j] = (double)y[ind - 1][j];
        s->log_w = (double)lw0[ind - 1];
        return 1;
    }
    return 0;
}

static int mala_check(const glabc_model* m, const glabc_dist* imp, const glabc_mala* p, const glabc_chains* c,
                      const glabc_run* r)
{
    int rc = model_check(m);
    if (rc) return rc;
    if (!imp || !p || !c || !r) return GLABC_ERR_NULL;
    if (!c->theta || !c->y || !c->flags || !c->theta64 || !c->y64 || !c->log_w64 || !c->grad) return GLABC_ERR_NULL;
    if (imp->dim != m->theta_dim) return GLABC_ERR_DIM;
    if (c->n_chains < 0 || c->stride < c->n_chains || r->n_steps < 0) return GLABC_ERR_ARG;
    if (r->batch_size < 1 || r->batch_size > GLABC_MAX_BATCH) return GLABC_ERR_ARG;
    if (p->num_grad < 2 || !(p->tau > 0.0)) return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->tape) return GLABC_ERR_ARG;
    return 0;
}

ORACLE_API int oracle_glmala_init(const glabc_model* m, const glabc_chains* c)
{
    if (!m || !c || !c->theta || !c->y || !c->flags || !c->theta64 || !c->y64 || !c->log_w64 || !c->grad)
        return GLABC_ERR_NULL;
    for (int64_t i = 0; i < c->n_chains; ++i) {
        for (int j = 0; j < m->theta_dim; ++j) {
            c->theta64[j * c->stride + i] = (double)c->theta[j * c->stride + i];
            c->grad[j * c->stride + i] = 0.0;
        }
        for (int j = 0; j < m->y_dim; ++j) c->y64[j * c->stride + i] = (double)c->y[j * c->stride + i];
        c->log_w64[i] = 0.0;
        c->flags[i] = GLABC_FLAG_LOCAL;                                                 /* GLMALA.py:146-147 */
    }
    return 0;
}

ORACLE_API int oracle_glmala_steps(const glabc_model* m, const glabc_dist* imp, const glabc_mala* p,
                                   const glabc_chains* c, const glabc_run* run)
{
    int rc = mala_check(m, imp, p, c, run);
    if (rc) return rc;
    int N = run->batch_size, d = m->theta_dim, yd = m->y_dim;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < c->n_chains; ++i) {
        mala_state s;
        step_draws dr;
        draws_init(&dr, GLABC_MAX_BATCH);
        for (int j = 0; j < d; ++j) {
            s.theta[j] = c->theta64[j * c->stride + i];
            s.grad[j] = c->grad[j * c->stride + i];
        }
        for (int j = 0; j < yd; ++j) s.y[j] = c->y64[j * c->stride + i];
        s.log_w = c->log_w64[i];
        s.flags = c->flags[i];
        s.n_moves = c->n_moves ? c->n_moves[i] : 0u;
        const uint64_t chain = (uint64_t)(c->chain0 + i);
        for (int64_t t = 0; t < run->n_steps; ++t) {
            const uint32_t step = run->step0 + (uint32_t)t;
            float prev[GLABC_MAX_DIM], cur[GLABC_MAX_DIM];
            for (int j = 0; j < d; ++j) prev[j] = (float)s.theta[j];
            draws_from_philox(&dr, run->seed, chain, step, N, d, yd, 0);
            if (dr.u_branch < run->global_frequency) {                                  /* GLMALA.py:151 */
                if (imp->kind == GLABC_DIST_UNIFORM) draws_from_philox(&dr, run->seed, chain, step, N, d, yd, 1);
                s.n_moves += (uint32_t)mala_isir_move(m, imp, N, &s, &dr);
            } else {
                s.n_moves += (uint32_t)mala_move(m, p, &s, &dr, run->seed, chain, step);
            }
            for (int j = 0; j < d; ++j) cur[j] = (float)s.theta[j];                     /* Theta_Re is float32, :148,180,200 */
            record(run, c, i, t, d, cur, prev);
        }
        for (int j = 0; j < d; ++j) {
            c->theta64[j * c->stride + i] = s.theta[j];
            c->grad[j * c->stride + i] = s.grad[j];
            c->theta[j * c->stride + i] = (float)s.theta[j];
        }
        for (int j = 0; j < yd; ++j) {
            c->y64[j * c->stride + i] = s.y[j];
            c->y[j * c->stride + i] = (float)s.y[j];
        }
        c->log_w64[i] = s.log_w;
        if (c->log_w) c->log_w[i] = (float)s.log_w;
        c->flags[i] = s.flags;
        if (c->n_moves) c->n_moves[i] = s.n_moves;
    }
    return 0;
}


/* ========================================================================= */
/* RealNVP coupling stack of GLMCMC_NF (GLMCMC_NFs.py:51-61,70-72,96-98).     */
/* normflows is a third-party dependency that is not in the reference tree    */
/* (setup.py:12 `normflows>=1.7.2`, unpinned) and cannot be installed here, so */
/* its published forward / inverse semantics are restated: this   */
/* part of the oracle is NOT pinned by reference outputs ("parity unpinned").  */
/* The arithmetic order is spelled out so that the matrix-core kernel can be   */
/* checked bit for bit: the 128x128 layer is a k-ascending fmaf chain from b2  */
/* (what v_mfma_f32_32x32x2_f32 computes), the 128->2 layer is two 64-term     */
/* fmaf chains over the hidden units with bit 2 of their index clear / set, in */
/* the order i = 32t + (r&3) + 8(r>>2) + 4h, t = 0..3, r = 0..15, then added.  */
/* ========================================================================= */
#define NF_H 128
#define NF_W1_OFF (NF_H * NF_H)
#define NF_B1_OFF (NF_W1_OFF + NF_H)
#define NF_V4_OFF (NF_B1_OFF + NF_H)
#define NF_B3_OFF (NF_V4_OFF + 4 * NF_H)

static void nf_coupling_params(const float* blk, float z0, float* shift, float* log_s)
{
    float h1[NF_H], part[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int k = 0; k < NF_H; ++k) h1[k] = fmaxf(__builtin_fmaf(blk[NF_W1_OFF + k], z0, blk[NF_B1_OFF + k]), 0.0f);
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) {
                int i = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float* v = blk + NF_V4_OFF + 4 * i;                                                 /* (b2, W3[0], W3[1], 0) */
                float acc = v[0];                                                                         /* chain starts at b2 */
                for (int k = 0; k < NF_H; ++k) acc = __builtin_fmaf(blk[k * NF_H + i], h1[k], acc);      /* W2^T[k][i] */
                float h2 = fmaxf(acc, 0.0f);
                part[h][0] = __builtin_fmaf(v[1], h2, part[h][0]);
                part[h][1] = __builtin_fmaf(v[2], h2, part[h][1]);
            }
    *shift = (part[0][0] + part[1][0]) + blk[NF_B3_OFF + 0];
    *log_s = (part[0][1] + part[1][1]) + blk[NF_B3_OFF + 1];
}

/* NF_model.sample: eps[2][n] (or NULL = Philox) -> z[2][n], log_q[n]; params are HOST pointers here */
ORACLE_API int oracle_nf_sample(const glabc_flow* f, const float* eps, uint64_t seed, int64_t row0, int64_t n, float* z,
                                float* log_q)
{
    if (!f || !f->params || !z || !log_q) return GLABC_ERR_NULL;
    if (f->hidden != NF_H || f->n_couplings < 1) return GLABC_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float e0, e1;
        if (eps) {
            e0 = eps[r];
            e1 = eps[n + r];
        } else {
            uint64_t gid = (uint64_t)(row0 + r);
            glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
            glabc_normal_pair(w.v[0], w.v[1], &e0, &e1);
        }
        float z0 = f->base_loc[0] + f->base_scale[0] * e0, z1 = f->base_loc[1] + f->base_scale[1] * e1;
        float lq = f->base_c0 - ((f->base_log_scale[0] + 0.5f * (e0 * e0)) + (f->base_log_scale[1] + 0.5f * (e1 * e1)));
        for (int c = 0; c < f->n_couplings; ++c) {
            float shift, log_s;
            nf_coupling_params(f->params + (int64_t)c * GLABC_NF_COUPLING_FLOATS, z0, &shift, &log_s);
            float nz = z1 * glabc_expf(log_s) + shift;
            lq = lq - log_s;
            z1 = z0;
            z0 = nz;
        }
        z[r] = z0;
        z[n + r] = z1;
        log_q[r] = lq;
    }
    return 0;
}

ORACLE_API int oracle_nf_log_prob(const glabc_flow* f, const float* x, int64_t n, float* log_q)
{
    if (!f || !f->params || !x || !log_q) return GLABC_ERR_NULL;
    if (f->hidden != NF_H || f->n_couplings < 1) return GLABC_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float z0 = x[r], z1 = x[n + r], lq = 0.0f;
        for (int c = f->n_couplings - 1; c >= 0; --c) {
            float t0 = z1, t1 = z0, shift, log_s;
            nf_coupling_params(f->params + (int64_t)c * GLABC_NF_COUPLING_FLOATS, t0, &shift, &log_s);
            z0 = t0;
            z1 = (t1 - shift) * glabc_expf(-log_s);
            lq = lq + (-log_s);
        }
        float e0 = (z0 - f->base_loc[0]) / f->base_scale[0], e1 = (z1 - f->base_loc[1]) / f->base_scale[1];
        float lp = f->base_c0 - ((f->base_log_scale[0] + 0.5f * (e0 * e0)) + (f->base_log_scale[1] + 0.5f * (e1 * e1)));
        log_q[r] = lq + lp;
    }
    return 0;
}


/* ---- the training step of GLMCMC_NFs.py:63,112-124: loss = forward_kld(x) = -mean(log_prob(x)), its gradient, Adam --------
 * The reference differentiates with autograd; this is the same derivative written out (chain rule through
 * base.log_prob and, coupling by coupling, through z1' = (z1 - shift(z0)) exp(-log_s(z0)), log_q -= log_s with the
 * MLP 1 -> 128 -> 128 -> 2, ReLU'(x) = [x > 0] as torch has it).
 * What is differentiated is the FLOAT32 evaluation, as autograd does: the downward pass is oracle_nf_log_prob's arithmetic
 * (so the loss and every ReLU gate are the float32 ones -- a gate whose pre-activation is within rounding of zero would
 * otherwise open in one evaluation and not in another, and the gradient jumps there), and the chain rule is then
 * evaluated in DOUBLE at those float32 states with those gates: the exact gradient of the active set up to 1e-15, against
 * which the float32 kernels are held with a tolerance (tests/test_nf_train.py; this function itself is checked against torch
 * autograd in float64 on cases where no gate is near zero). grad_params: block layout, doubles -> float at the end. */
static void nf_coupling_gates(const float* blk, float z0, unsigned char* gate1, unsigned char* gate2, float* shift, float* log_s)
{
    /* nf_coupling_params with the gates kept: gate1[k] = [W1 z0 + b1 > 0], gate2[i] = [a2_i > 0] */
    float h1[NF_H], part[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int k = 0; k < NF_H; ++k) {
        const float pre = __builtin_fmaf(blk[NF_W1_OFF + k], z0, blk[NF_B1_OFF + k]);
        gate1[k] = pre > 0.0f;
        h1[k] = fmaxf(pre, 0.0f);
    }
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) {
                int i = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float* v = blk + NF_V4_OFF + 4 * i;
                float acc = v[0];
                for (int k = 0; k < NF_H; ++k) acc = __builtin_fmaf(blk[k * NF_H + i], h1[k], acc);
                gate2[i] = acc > 0.0f;
                float h2 = fmaxf(acc, 0.0f);
                part[h][0] = __builtin_fmaf(v[1], h2, part[h][0]);
                part[h][1] = __builtin_fmaf(v[2], h2, part[h][1]);
            }
    *shift = (part[0][0] + part[1][0]) + blk[NF_B3_OFF + 0];
    *log_s = (part[0][1] + part[1][1]) + blk[NF_B3_OFF + 1];
}

ORACLE_API int oracle_nf_grad(const glabc_flow* f, const float* x, int64_t n, float* grad_params, float* grad_base, float* loss)
{
    if (!f || !f->params || !x || !grad_params || !grad_base || !loss) return GLABC_ERR_NULL;
    if (f->hidden != NF_H || f->n_couplings < 1 || n < 1) return GLABC_ERR_ARG;
    const int nc = f->n_couplings;
    const int64_t total = (int64_t)nc * GLABC_NF_COUPLING_FLOATS;
    double* acc = (double*)calloc((size_t)total + 5, sizeof(double));
    if (!acc) return GLABC_ERR_ARG;
    const double gl = -1.0 / (double)n;
    int failed = 0;
#pragma omp parallel
    {
        double* mine = (double*)calloc((size_t)total + 5, sizeof(double));
        float* st = (float*)malloc(sizeof(float) * 3 * (size_t)nc);                     /* per coupling: t0, z1', (unused) */
        unsigned char* gates = (unsigned char*)malloc((size_t)nc * 2 * NF_H);
        if (!mine || !st || !gates) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(static)
            for (int64_t r = 0; r < n; ++r) {
                /* down, in float32 (oracle_nf_log_prob's arithmetic): the states, the gates, log_prob */
                float z0 = x[r], z1 = x[n + r], lqf = 0.0f;
                for (int c = nc - 1; c >= 0; --c) {
                    const float* blk = f->params + (int64_t)c * GLABC_NF_COUPLING_FLOATS;
                    float t0 = z1, t1 = z0, shift, log_s;
                    nf_coupling_gates(blk, t0, gates + (size_t)c * 2 * NF_H, gates + (size_t)c * 2 * NF_H + NF_H, &shift, &log_s);
                    z0 = t0;
                    z1 = (t1 - shift) * glabc_expf(-log_s);
                    lqf = lqf + (-log_s);
                    st[3 * c] = t0;
                    st[3 * c + 1] = z1;
                }
                {
                    float e0 = (z0 - f->base_loc[0]) / f->base_scale[0], e1 = (z1 - f->base_loc[1]) / f->base_scale[1];
                    float lp = f->base_c0 - ((f->base_log_scale[0] + 0.5f * (e0 * e0)) + (f->base_log_scale[1] + 0.5f * (e1 * e1)));
                    mine[total] += (double)(lqf + lp);
                }
                const double e0 = ((double)z0 - f->base_loc[0]) / f->base_scale[0], e1 = ((double)z1 - f->base_loc[1]) / f->base_scale[1];
                mine[total + 1] += gl * (e0 / f->base_scale[0]);
                mine[total + 2] += gl * (e1 / f->base_scale[1]);
                mine[total + 3] += gl * (e0 * e0 - 1.0);
                mine[total + 4] += gl * (e1 * e1 - 1.0);
                double g0 = gl * (-(e0 / f->base_scale[0])), g1 = gl * (-(e1 / f->base_scale[1]));   /* dL/d(z0, z1) */
                /* back up in double at the float32 states, with the float32 gates: coupling 0 was applied last */
                for (int c = 0; c < nc; ++c) {
                    const float* blk = f->params + (int64_t)c * GLABC_NF_COUPLING_FLOATS;
                    double* gb = mine + (int64_t)c * GLABC_NF_COUPLING_FLOATS;
                    const unsigned char* gate1 = gates + (size_t)c * 2 * NF_H;
                    const unsigned char* gate2 = gate1 + NF_H;
                    const double t0 = st[3 * c], z1p = st[3 * c + 1];
                    double h1[NF_H], h2[NF_H], ls = blk[NF_B3_OFF + 1];
                    for (int k = 0; k < NF_H; ++k)
                        h1[k] = gate1[k] ? (double)blk[NF_W1_OFF + k] * t0 + (double)blk[NF_B1_OFF + k] : 0.0;
                    for (int i = 0; i < NF_H; ++i) {
                        double s2 = 0.0;
                        if (gate2[i]) {
                            s2 = blk[NF_V4_OFF + 4 * i];
                            for (int k = 0; k < NF_H; ++k) s2 += (double)blk[k * NF_H + i] * h1[k];
                        }
                        h2[i] = s2;
                        ls += (double)blk[NF_V4_OFF + 4 * i + 2] * s2;
                    }
                    const double dz1 = g1 * exp(-ls), dsh = -dz1, dls = -(g1 * z1p) - gl;
                    gb[NF_B3_OFF] += dsh;
                    gb[NF_B3_OFF + 1] += dls;
                    double dh1[NF_H];
                    for (int k = 0; k < NF_H; ++k) dh1[k] = 0.0;
                    for (int i = 0; i < NF_H; ++i) {
                        gb[NF_V4_OFF + 4 * i + 1] += dsh * h2[i];
                        gb[NF_V4_OFF + 4 * i + 2] += dls * h2[i];
                        if (!gate2[i]) continue;
                        const double da2 = (double)blk[NF_V4_OFF + 4 * i + 1] * dsh + (double)blk[NF_V4_OFF + 4 * i + 2] * dls;
                        gb[NF_V4_OFF + 4 * i] += da2;
                        for (int k = 0; k < NF_H; ++k) {
                            gb[k * NF_H + i] += da2 * h1[k];
                            dh1[k] += (double)blk[k * NF_H + i] * da2;
                        }
                    }
                    double dt0 = g0;
                    for (int k = 0; k < NF_H; ++k) {
                        if (!gate1[k]) continue;
                        gb[NF_W1_OFF + k] += dh1[k] * t0;
                        gb[NF_B1_OFF + k] += dh1[k];
                        dt0 += (double)blk[NF_W1_OFF + k] * dh1[k];
                    }
                    g0 = dz1;                                               /* dL/d(state in front of the coupling) */
                    g1 = dt0;
                }
            }
#pragma omp critical
            for (int64_t j = 0; j < total + 5; ++j) acc[j] += mine[j];
        }
        free(mine);
        free(st);
        free(gates);
    }
    if (!failed) {
        for (int64_t j = 0; j < total; ++j) grad_params[j] = (float)acc[j];
        *loss = (float)(-acc[total] / (double)n);
        for (int j = 0; j < 4; ++j) grad_base[j] = (float)acc[total + 1 + j];
    }
    free(acc);
    return failed ? GLABC_ERR_ARG : 0;
}

/* torch.optim.Adam.step() element by element, the operation order of glabc_adam_step (bit for bit) */
ORACLE_API int oracle_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                double eps, double weight_decay, int32_t step)
{
    if (!p || !g || !m || !v) return GLABC_ERR_NULL;
    if (n < 0 || step < 1) return GLABC_ERR_ARG;
    const float b1 = (float)beta1, omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2), e = (float)eps,
                wd = (float)weight_decay, step_size = (float)(lr / (1.0 - pow(beta1, (double)step))),
                bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    (void)b1;
    for (int64_t j = 0; j < n; ++j) {
        const float pj = p[j];
        const float gj = wd != 0.0f ? g[j] + wd * pj : g[j];
        const float mj = m[j] + (gj - m[j]) * omb1;
        const float vj = v[j] * b2 + omb2 * (gj * gj);
        const float denom = sqrtf(vj) / bias2_sqrt + e;
        m[j] = mj;
        v[j] = vj;
        p[j] = pj - step_size * (mj / denom);
    }
    return 0;
}


/* ---- GLMCMC_NF pool weights and one iteration against the pool (GLMCMC_NFs.py:73-111,141-152) ---- */
ORACLE_API int oracle_pool_weights(const glabc_model* m, const float* theta, const float* log_q, int64_t n, uint64_t seed,
                                   int64_t row_id0, float* x_out, float* w_out)
{
    int rc = model_check(m);
    if (rc) return rc;
    int d = m->theta_dim;
    for (int64_t r = 0; r < n; ++r) {
        float th[GLABC_MAX_DIM], y[GLABC_MAX_DIM], eps[GLABC_MAX_DIM + 4];
        for (int j = 0; j < d; ++j) th[j] = theta[j * n + r];
        uint64_t gid = (uint64_t)(row_id0 + r);
        for (int b = 0; b < (d + 3) / 4; ++b) {
            glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, (uint32_t)seed,
                                                (uint32_t)(seed >> 32));
            glabc_normal_pair(w.v[0], w.v[1], &eps[4 * b], &eps[4 * b + 1]);
            glabc_normal_pair(w.v[2], w.v[3], &eps[4 * b + 2], &eps[4 * b + 3]);
        }
        model_simulate(m, th, eps, y);                                                      /* :79 */
        float lw = (model_prior(m, th) + model_log_kernel(m, y)) - log_q[r];                /* :80-81 */
        float v = glabc_expf(lw);                                                           /* :82 */
        for (int j = 0; j < d; ++j) x_out[j * n + r] = y[j];
        w_out[r] = isnan(v) ? 0.0f : v;                                                     /* :83-85 */
    }
    return 0;
}

/* ---- KernelDensity (kernel_density.py:4-177) and AGLMCMC's pool weights (AGLMCMC.py:104-109,199-204) ---------- */

/* float64 sum of 256 strided partials combined by a binary tree: the fixed order glabc_kde_fit documents */
static double strided_tree_sum(const double* term, int64_t n)
{
    double part[256];
    for (int t = 0; t < 256; ++t) {
        double p = 0.0;
        for (int64_t i = t; i < n; i += 256) p = p + term[i];
        part[t] = p;
    }
    for (int s = 128; s >= 1; s >>= 1)
        for (int t = 0; t < s; ++t) part[t] = part[t] + part[t + s];
    return part[0];
}

ORACLE_API int oracle_kde_fit(const float* x, const float* w_raw, int64_t n, int32_t dim, double h, const float* bw_fixed,
                              float* weights, float* log_w, int64_t* wq, float* consts)
{
    if (!x || !weights || !log_w || !wq || !consts) return GLABC_ERR_NULL;
    if (dim < 1 || dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (n < 1) return GLABC_ERR_ARG;
    double* term = (double*)malloc(sizeof(double) * (size_t)n);
    float wsum32 = 0.0f;
    if (w_raw) {
        for (int64_t i = 0; i < n; ++i) term[i] = (double)w_raw[i];
        wsum32 = (float)strided_tree_sum(term, n);
    }
    for (int64_t i = 0; i < n; ++i) {                                            /* kernel_density.py:83-87 */
        float w = w_raw ? w_raw[i] / wsum32 : 1.0f / (float)n;
        weights[i] = w;
        log_w[i] = glabc_logf(w + 1e-10f);                                       /* :125 */
        wq[i] = glabc_fx_quantize((double)w);
    }
    float bw[GLABC_MAX_DIM];
    if (bw_fixed) {
        for (int d = 0; d < dim; ++d) bw[d] = bw_fixed[d];
    } else {                                                                     /* weighted_std, :40-68 */
        float hf = (float)h;
        for (int64_t i = 0; i < n; ++i) term[i] = (double)weights[i];
        float w2 = (float)strided_tree_sum(term, n);                             /* :54 */
        for (int64_t i = 0; i < n; ++i) {
            float w = weights[i] / w2;
            term[i] = (double)(w * w);
        }
        float corr = 1.0f - (float)strided_tree_sum(term, n);                    /* :64 */
        corr = corr < 1e-10f ? 1e-10f : corr;                                    /* :65 */
        for (int d = 0; d < dim; ++d) {
            for (int64_t i = 0; i < n; ++i) term[i] = (double)((weights[i] / w2) * x[d * n + i]);
            float mean = (float)strided_tree_sum(term, n);                       /* :57 */
            for (int64_t i = 0; i < n; ++i) {
                float diff = x[d * n + i] - mean;                                /* :60 */
                term[i] = (double)((weights[i] / w2) * (diff * diff));
            }
            float var = (float)strided_tree_sum(term, n) / corr;                 /* :62-65 */
            bw[d] = hf * sqrtf(var);                                             /* :67, :36 */
        }
    }
    float lb[GLABC_MAX_DIM];
    for (int d = 0; d < dim; ++d) {
        consts[d] = bw[d];
        lb[d] = glabc_logf(bw[d]);
    }
    consts[dim] = aten_rowsum_f32(lb, dim);                                      /* :122 */
    consts[dim + 1] = (float)(0.5 * dim) * 1.8378770351409912f;                  /* :121, float32 log(2 pi) */
    free(term);
    return 0;
}

/* the difference is multiplied by the correctly rounded 1/bandwidth (the reference divides, :117; <= 1 ulp per term) */
static float kde_log_term(const glabc_kde* k, const float* pt, int64_t s)
{
    float t[GLABC_MAX_DIM];
    for (int d = 0; d < k->dim; ++d) {
        float inv = 1.0f / k->bandwidth[d];
        float e = (pt[d] - k->x[d * k->n_samples + s]) * inv;                    /* :117 */
        t[d] = e * e;
    }
    float lk = -0.5f * aten_rowsum_f32(t, k->dim);                               /* :118 */
    lk = lk - k->c_2pi;                                                          /* :121 */
    lk = lk - k->sum_log_bw;                                                     /* :122 */
    return lk + k->log_w[s];                                                     /* :125 */
}

ORACLE_API int oracle_kde_log_prob(const glabc_kde* k, const float* pts, int64_t n_points, float* out)
{
    if (!k || !pts || !out) return GLABC_ERR_NULL;
    for (int64_t p = 0; p < n_points; ++p) {
        float pt[GLABC_MAX_DIM];
        for (int d = 0; d < k->dim; ++d) pt[d] = pts[d * n_points + p];
        float m = -INFINITY;
        int nan = 0;
        for (int64_t s = 0; s < k->n_samples; ++s) {                             /* torch.logsumexp: amax, :126 */
            float lk = kde_log_term(k, pt, s);
            nan |= isnan(lk);
            m = lk > m ? lk : m;
        }
        float m0 = isinf(m) ? 0.0f : m;
        if (nan) { out[p] = NAN; continue; }
        if (m == INFINITY) { out[p] = m; continue; }
        int64_t acc = 0;
        for (int64_t s = 0; s < k->n_samples; ++s)
            acc += glabc_fx_quantize((double)glabc_expf(kde_log_term(k, pt, s) - m0));
        out[p] = glabc_logf((float)((double)acc * 0x1p-40)) + m0;
    }
    return 0;
}

ORACLE_API int oracle_kde_sample(const glabc_kde* k, int64_t n, uint64_t seed, int64_t row0, float* out)
{
    if (!k || !out || !k->cum_q) return GLABC_ERR_NULL;
    int D = k->dim;
    int64_t S = k->n_samples;
    for (int64_t r = 0; r < n; ++r) {
        uint64_t gid = (uint64_t)(row0 + r);
        float nrm[GLABC_MAX_DIM + 6];
        double u = 0.0;
        int nb = (D + 2 + 3) / 4;
        for (int b = 0; b < nb; ++b) {
            glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, (uint32_t)seed,
                                                (uint32_t)(seed >> 32));
            if (b == 0) {
                u = glabc_uniform_f64(w.v[0], w.v[1]);
                glabc_normal_pair(w.v[2], w.v[3], &nrm[0], &nrm[1]);
            } else {
                glabc_normal_pair(w.v[0], w.v[1], &nrm[4 * b - 2], &nrm[4 * b - 1]);
                glabc_normal_pair(w.v[2], w.v[3], &nrm[4 * b], &nrm[4 * b + 1]);
            }
        }
        int64_t target = (int64_t)(u * (double)k->cum_q[S - 1]);
        int64_t j = 0;
        while (j < S - 1 && !(k->cum_q[j] > target)) ++j;                        /* torch.multinomial: inverse CDF, :141 */
        for (int d = 0; d < D; ++d) out[d * n + r] = k->x[d * S + j] + nrm[d] * k->bandwidth[d];   /* :147-148 */
    }
    return 0;
}

/* the raw draws of oracle_kde_sample (test hook for replaying the reference's AGLMCMC): u[n] float64, nrm[n][dim] */
ORACLE_API void oracle_kde_draws(uint64_t seed, int64_t row0, int64_t n, int dim, double* u, float* nrm_out)
{
    for (int64_t r = 0; r < n; ++r) {
        uint64_t gid = (uint64_t)(row0 + r);
        float nrm[GLABC_MAX_DIM + 6];
        int nb = (dim + 2 + 3) / 4;
        for (int b = 0; b < nb; ++b) {
            glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, (uint32_t)seed,
                                                (uint32_t)(seed >> 32));
            if (b == 0) {
                u[r] = glabc_uniform_f64(w.v[0], w.v[1]);
                glabc_normal_pair(w.v[2], w.v[3], &nrm[0], &nrm[1]);
            } else {
                glabc_normal_pair(w.v[0], w.v[1], &nrm[4 * b - 2], &nrm[4 * b - 1]);
                glabc_normal_pair(w.v[2], w.v[3], &nrm[4 * b], &nrm[4 * b + 1]);
            }
        }
        for (int d = 0; d < dim; ++d) nrm_out[r * dim + d] = nrm[d];
    }
}

ORACLE_API int oracle_kde_train_weights(const glabc_model* m, const float* theta, const float* dis, const float* log_q, int64_t n,
                                        float* w_out)
{
    int rc = model_check(m);
    if (rc) return rc;
    int d = m->theta_dim;
    for (int64_t r = 0; r < n; ++r) {
        float th[GLABC_MAX_DIM];
        for (int j = 0; j < d; ++j) th[j] = theta[j * n + r];
        float v = glabc_expf((model_prior(m, th) + model_log_kernel_dis(m, dis[r])) - log_q[r]);   /* AGLMCMC.py:200-201 */
        w_out[r] = isnan(v) ? 0.0f : v;
    }
    return 0;
}

ORACLE_API int oracle_glmcmc_nf_step(const glabc_model* m, const glabc_dist* local, const glabc_pool* pool,
                                     const glabc_chains* c, const glabc_run* run)
{
    int rc = model_check(m);
    if (rc) return rc;
    if (!local || !pool || !c || !run) return GLABC_ERR_NULL;
    if (run->n_steps != 1) return GLABC_ERR_ARG;
    int d = m->theta_dim, yd = m->y_dim, N = run->batch_size;
    int64_t C = c->n_chains, rows = (int64_t)pool->step_size * N * C;
    for (int64_t i = 0; i < C; ++i) {
        chain_state s;
        step_draws dr;
        draws_init(&dr, GLABC_MAX_BATCH);
        load_chain(&s, c, i, d, yd);
        draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0, 1, d, yd, 0);
        int kk = pool->kk[i], moved = 0;
        if (dr.u_branch < run->global_frequency) {                                          /* :91-92 */
            float w[GLABC_MAX_BATCH + 1];
            w[0] = glabc_expf((model_prior(m, s.theta) + model_log_kernel(m, s.y)) - pool->log_q_old[i]);   /* :99-101 */
            int have = kk < pool->step_size;
            int64_t base = ((int64_t)kk * N) * C + i;
            for (int j = 0; j < N; ++j) w[j + 1] = have ? pool->w[base + (int64_t)j * C] : 0.0f;
            float tot = aten_rowsum_f32(w, N + 1);                                          /* :103 */
            int ind = -1;
            double acc = 0.0;
            for (int k = 0; k <= N; ++k) {
                acc += (double)(w[k] / tot);
                if (dr.u_resample < acc) { ind = k; break; }                                /* :104 */
            }
            if (ind > 0 && have) {                                                          /* :105-107 */
                int64_t r = base + (int64_t)(ind - 1) * C;
                for (int j = 0; j < d; ++j) s.theta[j] = pool->theta[j * rows + r];
                for (int j = 0; j < yd; ++j) s.y[j] = pool->x[j * rows + r];
                moved = 1;
            }
            kk += 1;                                                                        /* :111 */
        } else {
            if (local->kind == GLABC_DIST_UNIFORM)
                draws_from_philox(&dr, run->seed, (uint64_t)(c->chain0 + i), run->step0, 1, d, yd, 1);
            moved = local_move(m, local, &s, &dr);                                          /* :141-152 */
        }
        s.n_moves += (uint32_t)moved;
        for (int j = 0; j < d; ++j) {
            c->theta[j * c->stride + i] = s.theta[j];
            if (run->history) run->history[j * run->hist_stride + i] = s.theta[j];
        }
        for (int j = 0; j < yd; ++j) c->y[j * c->stride + i] = s.y[j];
        if (c->n_moves) c->n_moves[i] = s.n_moves;
        pool->kk[i] = kk;
        if (pool->moved_idx && moved) pool->moved_idx[(*pool->n_moved)++] = (int32_t)i;
    }
    if (pool->n_moved_reset) *pool->n_moved_reset = 0;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* ESJD.py:2-25 : det( D^T D / (n-1) )^(1/d), D = consecutive differences, all float32.
 * torch.det is an LU with partial pivoting; restated for d <= GLABC_MAX_DIM.
 * history is chain-major [n_rows][d][stride] as the samplers write it. */
ORACLE_API int oracle_esjd(const float* history, int64_t n_rows, int32_t d, int64_t n_chains, int64_t stride,
                           float* out)
{
    if (!history || !out) return GLABC_ERR_NULL;
    if (d < 1 || d > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (n_rows < 2) return GLABC_ERR_ARG;
    for (int64_t c = 0; c < n_chains; ++c) {
        float a[GLABC_MAX_DIM][GLABC_MAX_DIM];
        memset(a, 0, sizeof a);
        for (int64_t t = 1; t < n_rows; ++t) {
            float dl[GLABC_MAX_DIM];
            for (int j = 0; j < d; ++j)
                dl[j] = history[(t * d + j) * stride + c] - history[((t - 1) * d + j) * stride + c];
            for (int p = 0; p < d; ++p)
                for (int q = 0; q < d; ++q) a[p][q] += dl[p] * dl[q];
        }
        float nd = (float)(n_rows - 1);
        for (int p = 0; p < d; ++p)
            for (int q = 0; q < d; ++q) a[p][q] = a[p][q] / nd;
        float det = 1.0f;
        for (int k = 0; k < d; ++k) {
            int piv = k;
            for (int r = k + 1; r < d; ++r)
                if (fabsf(a[r][k]) > fabsf(a[piv][k])) piv = r;
            if (piv != k) {
                for (int q = 0; q < d; ++q) { float tmp = a[k][q]; a[k][q] = a[piv][q]; a[piv][q] = tmp; }
                det = -det;
            }
            det *= a[k][k];
            if (a[k][k] == 0.0f) break;
            for (int r = k + 1; r < d; ++r) {
                float f = a[r][k] / a[k][k];
                for (int q = k; q < d; ++q) a[r][q] -= f * a[k][q];
            }
        }
        out[c] = powf(det, 1.0f / (float)d);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* numerics hooks for tests/test_numerics.py */

ORACLE_API void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    glabc_u32x4 r = glabc_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}

ORACLE_API void oracle_expf_v(const float* x, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_expf(x[i]);
}

ORACLE_API void oracle_fx_quantize_v(const double* d, int64_t n, int64_t* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_fx_quantize(d[i]);
}

/* test hook: both fixed-point accumulators over the same stream of quantised terms -> (s1, s2_lo, s2_hi) twice */
ORACLE_API void oracle_fx_both(const int64_t* q, int64_t n, uint64_t* out6)
{
    glabc_fxsum a = {0, 0, 0};
    glabc_fxsplit b = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        glabc_fx_add(&a, q[i]);
        glabc_fxs_add(&b, q[i]);
    }
    glabc_fxsum c = glabc_fxs_finish(&b);
    out6[0] = (uint64_t)a.s1; out6[1] = a.s2_lo; out6[2] = a.s2_hi;
    out6[3] = (uint64_t)c.s1; out6[4] = c.s2_lo; out6[5] = c.s2_hi;
}

ORACLE_API void oracle_expf_b_v(const float* x, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_expf_b(x[i]);
}

ORACLE_API void oracle_logf_v(const float* x, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_logf(x[i]);
}

ORACLE_API void oracle_sincos2pi_v(const float* u, int64_t n, float* s, float* c)
{
    for (int64_t i = 0; i < n; ++i) glabc_sincos2pi(u[i], s + i, c + i);
}

ORACLE_API void oracle_normal_pair_v(const uint32_t* a, const uint32_t* b, int64_t n, float* z0, float* z1)
{
    for (int64_t i = 0; i < n; ++i) glabc_normal_pair(a[i], b[i], z0 + i, z1 + i);
}

ORACLE_API void oracle_uniforms_v(const uint32_t* a, const uint32_t* b, int64_t n, float* u, float* upos, double* u64)
{
    for (int64_t i = 0; i < n; ++i) {
        u[i] = glabc_uniform_f32(a[i]);
        upos[i] = glabc_uniform_pos_f32(a[i]);
        u64[i] = glabc_uniform_f64(a[i], b[i]);
    }
}

ORACLE_API void oracle_exp_v(const double* x, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_exp(x[i]);
}

ORACLE_API void oracle_log_v(const double* x, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = glabc_log(x[i]);
}

/* the draws of one (chain, step) exactly as the samplers consume them */
ORACLE_API void oracle_step_draws(uint64_t seed, uint64_t chain, uint32_t step, int n_prop, int d, int yd,
                                  float* u2, double* r, float* z)
{
    step_draws dr;
    if (draws_init(&dr, n_prop)) return;
    draws_from_philox(&dr, seed, chain, step, n_prop, d, yd, 0);
    u2[0] = dr.u_branch;
    u2[1] = dr.u_accept;
    *r = dr.u_resample;
    for (int j = 0; j < n_prop; ++j)
        for (int i = 0; i < d + yd; ++i) z[j * (d + yd) + i] = dr.z[j][i];
    draws_release(&dr);
}

/* ========================================================================= */
/* Split-phase iteration (include/glabc.h: glabc_propose / glabc_select): the  */
/* same iteration of GLMCMC.py:58-104 / GlobalMCMC.py:37-68 cut where the loop  */
/* calls the Model, so that the Model can be arbitrary code.                    */
/* ========================================================================= */

/* the draws of candidate j of (chain, step): e[d] proposal noise (normals, or [0,1) uniforms for a Uniform
 * proposal), noise[nd] simulator normals from word dp = d rounded up to even */
static void candidate_draws(uint64_t seed, uint64_t chain, uint32_t step, int j, int d, int nd, int uniform, float* e,
                            float* noise)
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), c0 = (uint32_t)chain, c1 = (uint32_t)(chain >> 32);
    const int dp = d + (d & 1), spp = (dp + nd + 3) / 4;
    for (int b = 0; b < spp; ++b) {
        glabc_u32x4 o = glabc_philox4x32_10(c0, c1, step, (uint32_t)(1 + j * spp + b), k0, k1);
        float nrm[4];
        glabc_normal_pair(o.v[0], o.v[1], &nrm[0], &nrm[1]);
        glabc_normal_pair(o.v[2], o.v[3], &nrm[2], &nrm[3]);
        for (int t = 0; t < 4; ++t) {
            int g = 4 * b + t;
            if (g < d && g < GLABC_MAX_DIM) e[g] = uniform ? glabc_uniform_f32(o.v[t]) : nrm[t];   /* d > 8: callback proposals, e unused */
            if (noise && g >= dp && g - dp < nd) noise[g - dp] = nrm[t];
        }
    }
}

static int step_io_check(int algo, const glabc_chains* c, const glabc_run* r, const glabc_step_io* io)
{
    if (!c || !r || !io) return GLABC_ERR_NULL;
    if (algo != GLABC_ALGO_GLMCMC && algo != GLABC_ALGO_GLOBALMCMC && algo != GLABC_ALGO_GLMALA) return GLABC_ERR_KIND;
    if (io->theta_dim < 1 || io->y_dim < 1 || io->noise_dim < 0) return GLABC_ERR_DIM;
    if (io->n_prop < 1 || (algo == GLABC_ALGO_GLOBALMCMC && io->n_prop != 1)) return GLABC_ERR_ARG;
    if (r->n_steps != 1 || r->tape) return GLABC_ERR_ARG;
    if (!c->theta || !c->y) return GLABC_ERR_NULL;
    return 0;
}

ORACLE_API int oracle_propose(int algo, const glabc_dist* local, const glabc_dist* global, const glabc_chains* c,
                              const glabc_run* run, const glabc_step_io* io)
{
    int rc = step_io_check(algo, c, run, io);
    if (rc) return rc;
    const int d = io->theta_dim, nd = io->noise_dim, N = io->n_prop;
    if ((local || global) && d > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    const int64_t C = c->n_chains;
    const uint32_t step = run->step0_device ? *run->step0_device : run->step0;     /* glabc_run.step0_device */
    for (int64_t i = 0; i < C; ++i) {
        const uint64_t chain = (uint64_t)(c->chain0 + i);
        glabc_u32x4 h = glabc_philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), step, 0u, (uint32_t)run->seed,
                                            (uint32_t)(run->seed >> 32));
        const float gf = run->global_frequency_per_chain ? run->global_frequency_per_chain[i] : run->global_frequency;
        const int is_global = glabc_uniform_f32(h.v[0]) < gf;                   /* GLMCMC.py:59 */
        io->is_global[i] = is_global;
        io->log_u[i] = glabc_logf(glabc_uniform_f32(h.v[1]));                   /* GLMCMC.py:98 */
        io->u_res[i] = glabc_uniform_f64(h.v[2], h.v[3]);                       /* GLMCMC.py:17 */
        for (int j = 0; j < N; ++j) {
            const int64_t r = (int64_t)j * C + i;
            const glabc_dist* g = (j == 0 && !is_global) ? local : global;
            float e[GLABC_MAX_DIM] = {0}, z[GLABC_MAX_DIM], lq = 0.0f;
            candidate_draws(run->seed, chain, step, j, d, nd, g && g->kind == GLABC_DIST_UNIFORM, e,
                            io->sim_noise ? io->sim_noise + r * nd : NULL);
            if (!g) continue;                                                   /* the caller fills these rows */
            if (g->kind == GLABC_DIST_GAMMA && !(j == 0 && !is_global)) {       /* a Gamma importance / global proposal */
                lq = gamma_dist_forward(g, run->seed, chain, step, j, z);
                rc = 0;
            } else {
                rc = prop_forward(g, e, z, &lq);                                /* GLMCMC.py:66 / :91 */
            }
            if (rc) return rc;
            if (j == 0 && !is_global) {
                for (int k = 0; k < d; ++k) z[k] = z[k] + c->theta[k * c->stride + i];    /* GLMCMC.py:91 */
                lq = 0.0f;
            }
            memcpy(io->theta_prop + r * d, z, sizeof(float) * d);
            io->log_q[r] = lq;
        }
    }
    return 0;
}

/* GLMCMC.py:92-93 */
ORACLE_API int oracle_propose_redraw(const glabc_dist* local, const glabc_chains* c, const glabc_run* run,
                                     const glabc_step_io* io, int32_t round, int32_t* n_redrawn)
{
    int rc = step_io_check(GLABC_ALGO_GLMCMC, c, run, io);
    if (rc) return rc;
    if (!local || !n_redrawn) return GLABC_ERR_NULL;
    const int d = io->theta_dim;
    const float sentinel = (float)(7.0 * log(1e-10));
    for (int64_t i = 0; i < c->n_chains; ++i) {
        if ((io->is_global[i] & 1) || io->prior_prop[i] != sentinel) continue;
        const uint64_t chain = (uint64_t)(c->chain0 + i);
        float e[GLABC_MAX_DIM], z[GLABC_MAX_DIM], lq;
        for (int b = 0; b < 2; ++b) {
            glabc_u32x4 o = glabc_philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32),
                                                run->step0_device ? *run->step0_device : run->step0,
                                                GLABC_SLOT_REDRAW + (uint32_t)(2 * round + b), (uint32_t)run->seed,
                                                (uint32_t)(run->seed >> 32));
            float nrm[4];
            glabc_normal_pair(o.v[0], o.v[1], &nrm[0], &nrm[1]);
            glabc_normal_pair(o.v[2], o.v[3], &nrm[2], &nrm[3]);
            for (int t = 0; t < 4; ++t) e[4 * b + t] = local->kind == GLABC_DIST_UNIFORM ? glabc_uniform_f32(o.v[t]) : nrm[t];
        }
        prop_forward(local, e, z, &lq);
        for (int k = 0; k < d; ++k) io->theta_prop[i * d + k] = z[k] + c->theta[k * c->stride + i];
        *n_redrawn += 1;
    }
    return 0;
}

ORACLE_API int oracle_select(int algo, const glabc_dist* global, const glabc_chains* c, const glabc_run* run,
                             const glabc_step_io* io)
{
    int rc = step_io_check(algo, c, run, io);
    if (rc) return rc;
    if (!global && !io->q_cur) return GLABC_ERR_NULL;
    if (algo != GLABC_ALGO_GLOBALMCMC && (!c->log_w || !c->flags)) return GLABC_ERR_NULL;
    const int isir = algo != GLABC_ALGO_GLOBALMCMC;
    const int d = io->theta_dim, yd = io->y_dim, N_all = io->n_prop;
    const int64_t C = c->n_chains;
    float* lw = (float*)malloc(sizeof(float) * (size_t)(N_all + 1));
    float* w = (float*)malloc(sizeof(float) * (size_t)(N_all + 1));
    float* th_old = (float*)malloc(sizeof(float) * (size_t)d);
    if (!lw || !w || !th_old) { free(lw); free(w); free(th_old); return GLABC_ERR_ARG; }
    for (int64_t i = 0; i < C; ++i) {
        const int is_global = io->is_global[i] & 1;
        /* GLMCMC.py:67-70: rows with a NaN coordinate are gone before generate_samples; the caller has compacted the chain's
         * candidates and says how many survive */
        int N = N_all;
        if (io->n_valid && is_global) N = io->n_valid[i] < 0 ? 0 : (io->n_valid[i] < N_all ? io->n_valid[i] : N_all);
        const float prior_c = io->prior_cur[i], kern_c = io->kern_cur[i];
        for (int k = 0; k < d; ++k) th_old[k] = c->theta[k * c->stride + i];
        float q_state = 0.0f;
        if (io->q_cur) q_state = io->q_cur[i];
        else dist_log_prob(global, th_old, &q_state);
        int ind = 0;
        if (isir && is_global) {
            if (c->flags[i] & GLABC_FLAG_LOCAL) c->log_w[i] = (prior_c + kern_c) - q_state;       /* GLMCMC.py:60-64 */
            c->flags[i] &= ~GLABC_FLAG_LOCAL;                                                     /* :65 */
            lw[0] = c->log_w[i];
            for (int j = 0; j < N; ++j) {
                const int64_t r = (int64_t)j * C + i;
                lw[j + 1] = (io->prior_prop[r] + io->kern_prop[r]) - io->log_q[r];                /* :74 */
            }
            for (int k = 0; k <= N; ++k) {
                w[k] = glabc_expf(lw[k]);                                                         /* :78 */
                if (isnan(w[k])) w[k] = 0.0f;                                                     /* :80-81 */
            }
            const float tot = aten_rowsum_f32(w, N + 1);                                          /* :82 */
            double acc = 0.0;
            ind = -1;
            for (int k = 0; k <= N; ++k) {                                                        /* :17-22 */
                acc += (double)(w[k] / tot);
                if (io->u_res[i] < acc) { ind = k; break; }
            }
            if (ind < 0) ind = 0;                                                                 /* :84 */
        } else {
            const float pk = io->prior_prop[i] + io->kern_prop[i];
            float log_acc;
            if (algo == GLABC_ALGO_GLOBALMCMC && is_global)
                log_acc = (((pk + q_state) - io->log_q[i]) - prior_c) - kern_c;                   /* GlobalMCMC.py:44-46 */
            else if (algo == GLABC_ALGO_GLMALA)
                log_acc = ((pk + io->log_q[i]) - prior_c) - kern_c;                               /* GLMALA.py:190-193 */
            else
                log_acc = (pk - prior_c) - kern_c;                                                /* GLMCMC.py:96-97 */
            ind = io->log_u[i] < log_acc ? 1 : 0;
        }
        if (ind > 0) {
            const int64_t r = (int64_t)(ind - 1) * C + i;
            for (int k = 0; k < d; ++k) c->theta[k * c->stride + i] = io->theta_prop[r * d + k];
            for (int k = 0; k < yd; ++k) c->y[k * c->stride + i] = io->y_prop[r * yd + k];
            io->prior_cur[i] = io->prior_prop[r];
            io->kern_cur[i] = io->kern_prop[r];
            if (isir) {
                if (is_global) c->log_w[i] = lw[ind];                                             /* GLMCMC.py:86 */
                else if (algo == GLABC_ALGO_GLMCMC) c->flags[i] |= GLABC_FLAG_LOCAL;              /* GLMCMC.py:100, absent from GLMALA.py:195-199 */
            }
            if (c->n_moves) c->n_moves[i] += 1u;
            io->is_global[i] |= 2;
        }
        /* Theta_Re row and streaming sums, any theta_dim */
        if (run->history) {
            float* row = run->history + (run->step0_device ? (int64_t)(*run->step0_device - run->step0) * d * run->hist_stride : 0);
            for (int k = 0; k < d; ++k) row[k * run->hist_stride + i] = c->theta[k * c->stride + i];
        }
        if (run->moments) {
            const glabc_moments* m = run->moments;
            int k = 0;
            for (int a = 0; a < d; ++a) {
                const float ta = c->theta[a * c->stride + i];
                m->sum_theta[a * c->stride + i] += (double)ta;
                for (int b = a; b < d; ++b, ++k) {
                    const float tb = c->theta[b * c->stride + i];
                    m->sum_outer[k * c->stride + i] += (double)ta * (double)tb;
                    m->sum_jump[k * c->stride + i] += ((double)ta - (double)th_old[a]) * ((double)tb - (double)th_old[b]);
                }
            }
        }
    }
    free(lw);
    free(w);
    free(th_old);
    return 0;
}
