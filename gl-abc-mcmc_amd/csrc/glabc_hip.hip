// glabc_hip.hip -- gfx950 kernels and the C ABI of include/glabc.h.
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared glabc_hip.hip -o libglabc_hip.so
//
// Kernels
//   sampler_kernel<ALGO, D, N>   fused K-iteration GLMCMC / GlobalMCMC step, one work-item per chain
//   init_weights_kernel<D>       GLMCMC.py:52-55
//   rowwise_kernel<OP>           distribution / Model callbacks on row-major points
//   esjd_kernel<D>               ESJD.py:2-25 per chain from a chain-major history
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "glabc_device.h"

namespace glabc {

enum Algo { ALGO_GLMCMC = 0, ALGO_GLOBAL = 1 };

constexpr int BLOCK = 64;      // one wavefront per workgroup: 65 536 chains -> 1024 workgroups over 256 CUs x 4 SIMDs

// ---- the fused sampler ------------------------------------------------------------------
template <int ALGO, int D, int N>
__global__ void __launch_bounds__(BLOCK) sampler_kernel(const StepArgs<D> a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_chains) return;

    Chain<D> c;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        c.theta[j] = a.theta[j * a.stride + i];
        c.y[j] = a.y[j * a.stride + i];
    }
    c.log_w = (ALGO == ALGO_GLMCMC) ? a.log_w[i] : 0.0f;
    c.flags = (ALGO == ALGO_GLMCMC) ? a.flags[i] : 0u;
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;

    constexpr int TRI = D * (D + 1) / 2;
    const bool mom = a.sum_theta != nullptr;
    double s1[D], s2[TRI], sj[TRI];
    if (mom) {
#pragma unroll
        for (int j = 0; j < D; ++j) s1[j] = a.sum_theta[j * a.stride + i];
#pragma unroll
        for (int k = 0; k < TRI; ++k) {
            s2[k] = a.sum_outer[k * a.stride + i];
            sj[k] = a.sum_jump[k * a.stride + i];
        }
    }

    const uint64_t gid = (uint64_t)(a.chain0 + i);
    Rng rng;
    rng.c0 = (uint32_t)gid;
    rng.c1 = (uint32_t)(gid >> 32);
    rng.k0 = a.seed_lo;
    rng.k1 = a.seed_hi;

    float* hist = a.history ? a.history + i : nullptr;

    for (int t = 0; t < a.n_steps; ++t) {
        const uint32_t step = a.step0 + (uint32_t)t;
        float prev[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = c.theta[j];

        StepHead h = draw_head(rng, step);
        bool moved;
        if (h.u_branch < a.gf) {                                   // GLMCMC.py:59 / GlobalMCMC.py:39
            if constexpr (ALGO == ALGO_GLMCMC)
                moved = isir_move<D, N>(a, rng, step, h.u_resample, c);
            else
                moved = independence_move<D>(a, rng, step, h.u_accept, c);
        } else {
            moved = local_move<D>(a, rng, step, h.u_accept, c);
            if (ALGO == ALGO_GLMCMC && moved) c.flags |= GLABC_FLAG_LOCAL;    // GLMCMC.py:100
        }
        c.n_moves += moved ? 1u : 0u;

        if (hist) {                                                // Theta_Re[i,:] = Theta_old, GLMCMC.py:89,104
#pragma unroll
            for (int j = 0; j < D; ++j) hist[((int64_t)t * D + j) * a.hist_stride] = c.theta[j];
        }
        if (mom) {
            int k = 0;
#pragma unroll
            for (int p = 0; p < D; ++p) {
                s1[p] += (double)c.theta[p];
#pragma unroll
                for (int q = p; q < D; ++q, ++k) {
                    s2[k] += (double)c.theta[p] * (double)c.theta[q];
                    double dp = (double)c.theta[p] - (double)prev[p];
                    double dq = (double)c.theta[q] - (double)prev[q];
                    sj[k] += dp * dq;
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < D; ++j) {
        a.theta[j * a.stride + i] = c.theta[j];
        a.y[j * a.stride + i] = c.y[j];
    }
    if (ALGO == ALGO_GLMCMC) {
        a.log_w[i] = c.log_w;
        a.flags[i] = c.flags;
    }
    if (a.n_moves) a.n_moves[i] = c.n_moves;
    if (mom) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.sum_theta[j * a.stride + i] = s1[j];
#pragma unroll
        for (int k = 0; k < TRI; ++k) {
            a.sum_outer[k * a.stride + i] = s2[k];
            a.sum_jump[k * a.stride + i] = sj[k];
        }
    }
}

template <int D>
__global__ void __launch_bounds__(BLOCK) init_weights_kernel(const StepArgs<D> a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_chains) return;
    Chain<D> c;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        c.theta[j] = a.theta[j * a.stride + i];
        c.y[j] = a.y[j * a.stride + i];
    }
    a.log_w[i] = isir_weight_of_state<D>(a, c);
    a.flags[i] = a.flags[i] | GLABC_FLAG_LOCAL;                    // GLMCMC.py:50
}

// ---- row-major callbacks: one work-item per point ------------------------------------------
enum RowOp { ROW_DIST_LOG_PROB = 0, ROW_PRIOR = 1, ROW_DISCREPANCY = 2, ROW_LOG_KERNEL = 3 };

struct RowArgs {
    glabc_dist dist;              // ROW_DIST_LOG_PROB / ROW_PRIOR
    float y_obs[GLABC_MAX_DIM];
    float kern_log_scale, kern_scale, kern_c0;
    int32_t dim;
    const float* in;
    float* out;
    int64_t n;
};

template <int D>
__device__ __forceinline__ DistArgs<D> narrow(const glabc_dist& g)
{
    DistArgs<D> o;
    o.kind = g.kind;
    o.c0 = g.c0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        o.p0[j] = g.p0[j];
        o.p1[j] = g.p1[j];
        o.p2[j] = g.p2[j];
    }
    return o;
}

template <int OP, int D>
__global__ void __launch_bounds__(256) rowwise_kernel(const RowArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float x[D];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = a.in[i * D + j];
    float r;
    if constexpr (OP == ROW_DIST_LOG_PROB || OP == ROW_PRIOR) {
        DistArgs<D> g = narrow<D>(a.dist);
        r = dist_log_prob<D>(g, x);
    } else {
        float t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float d = x[j] - a.y_obs[j];
            t[j] = d * d;
        }
        float dis = __builtin_sqrtf(aten_rowsum<D>(t));            // Mixture.py:36
        if constexpr (OP == ROW_DISCREPANCY) {
            r = dis;
        } else {
            float e = (dis - 0.0f) / a.kern_scale;                 // Mixture.py:42-44
            r = a.kern_c0 - (a.kern_log_scale + 0.5f * (e * e));
        }
    }
    a.out[i] = r;
}

// det( M / n )^(1/d) in float32 for a symmetric M given by its upper triangle (ESJD.py:21-24):
// Gaussian elimination with row exchanges (torch.det is an LU with partial pivoting).
template <int D>
__device__ __forceinline__ float det_root(const double (&m)[D][D], float nd)
{
    float a[D][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = 0; q < D; ++q) a[p][q] = (float)m[p < q ? p : q][p < q ? q : p] / nd;    // ESJD.py:21
    float det = 1.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) {
#pragma unroll
        for (int r = k + 1; r < D; ++r) {
            bool sw = __builtin_fabsf(a[r][k]) > __builtin_fabsf(a[k][k]);
#pragma unroll
            for (int q = 0; q < D; ++q) {
                float x = a[k][q], y = a[r][q];
                a[k][q] = sw ? y : x;
                a[r][q] = sw ? x : y;
            }
            det = sw ? -det : det;
        }
        det *= a[k][k];
#pragma unroll
        for (int r = k + 1; r < D; ++r) {
            float f = a[k][k] != 0.0f ? a[r][k] / a[k][k] : 0.0f;       // singular: det is already 0
#pragma unroll
            for (int q = k; q < D; ++q) a[r][q] -= f * a[k][q];
        }
    }
    if constexpr (D == 1)                                              // ESJD.py:24: det ** (1/d)
        return det;
    else if constexpr (D == 2)
        return __builtin_sqrtf(det);
    else
        return det > 0.0f ? glabc_expf(glabc_logf(det) / (float)D) : (det == 0.0f ? 0.0f : __builtin_nanf(""));
}

// ---- ESJD.py:2-25 ----------------------------------------------------------------------------
// One work-item per chain walks its column of the chain-major history (coalesced across the
// wavefront), accumulates D^T D in double, then det(.)^(1/d) by Gaussian elimination with
// row exchanges (torch.det is an LU with partial pivoting).
template <int D>
__global__ void __launch_bounds__(BLOCK) esjd_kernel(const float* __restrict__ hist, int64_t n_rows, int64_t n_chains,
                                                     int64_t stride, float* __restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n_chains) return;
    double m[D][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = 0; q < D; ++q) m[p][q] = 0.0;
    float prev[D];
#pragma unroll
    for (int j = 0; j < D; ++j) prev[j] = hist[j * stride + c];
    for (int64_t t = 1; t < n_rows; ++t) {
        float dl[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float v = hist[(t * D + j) * stride + c];
            dl[j] = v - prev[j];                                   // ESJD.py:17 (float32 difference)
            prev[j] = v;
        }
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = p; q < D; ++q) m[p][q] += (double)dl[p] * (double)dl[q];
    }
    out[c] = det_root<D>(m, (float)(n_rows - 1));
}

// ESJD.py:21-24 from the streamed jump sums of glabc_moments (no history needed)
template <int D>
__global__ void __launch_bounds__(BLOCK) moments_esjd_kernel(const double* __restrict__ sum_jump, int64_t n_steps,
                                                             int64_t n_chains, int64_t stride, float* __restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n_chains) return;
    double m[D][D];
    int k = 0;
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = p; q < D; ++q, ++k) m[p][q] = sum_jump[k * stride + c];
    out[c] = det_root<D>(m, (float)n_steps);
}

}  // namespace glabc

// =================================================================================================
// host side: validation, argument marshalling, dispatch
// =================================================================================================
using namespace glabc;

static thread_local int g_last_hip_error = 0;

static bool finite_dist(const glabc_dist* g)
{
    for (int j = 0; j < g->dim; ++j)
        if (!std::isfinite(g->p0[j]) || !std::isfinite(g->p1[j]) || !std::isfinite(g->p2[j])) return false;
    return std::isfinite(g->c0);
}

static int check_dist(const glabc_dist* g, int dim)
{
    if (!g) return GLABC_ERR_NULL;
    if (g->dim < 1 || g->dim > GLABC_MAX_DIM || (dim > 0 && g->dim != dim)) return GLABC_ERR_DIM;
    if (g->kind != GLABC_DIST_DIAG_GAUSS && g->kind != GLABC_DIST_UNIFORM) return GLABC_ERR_KIND;
    if (!finite_dist(g)) return GLABC_ERR_ARG;
    if (g->kind == GLABC_DIST_DIAG_GAUSS)
        for (int j = 0; j < g->dim; ++j)
            if (!(g->p2[j] > 0.0f)) return GLABC_ERR_ARG;
    return GLABC_OK;
}

static int check_model(const glabc_model* m)
{
    if (!m) return GLABC_ERR_NULL;
    if (m->sim_kind != GLABC_SIM_ABS_GAUSS) return GLABC_ERR_KIND;
    if (m->theta_dim < 1 || m->theta_dim > GLABC_MAX_DIM || m->y_dim != m->theta_dim) return GLABC_ERR_DIM;
    int rc = check_dist(&m->prior, m->theta_dim);
    if (rc) return rc;
    rc = check_dist(&m->noise, m->y_dim);
    if (rc) return rc;
    if (m->noise.kind != GLABC_DIST_DIAG_GAUSS) return GLABC_ERR_KIND;
    if (!std::isfinite(m->kern_log_scale) || !(m->kern_scale > 0.0f) || !std::isfinite(m->kern_scale) ||
        !std::isfinite(m->kern_c0))
        return GLABC_ERR_ARG;
    for (int j = 0; j < m->y_dim; ++j)
        if (!std::isfinite(m->y_obs[j])) return GLABC_ERR_ARG;
    return GLABC_OK;
}

template <int D>
static DistArgs<D> pack_dist(const glabc_dist* g)
{
    DistArgs<D> o;
    o.kind = g->kind;
    o.c0 = g->c0;
    for (int j = 0; j < D; ++j) {
        o.p0[j] = g->p0[j];
        o.p1[j] = g->p1[j];
        o.p2[j] = g->p2[j];
    }
    return o;
}

template <int D>
static StepArgs<D> pack_args(const glabc_model* m, const glabc_dist* local, const glabc_dist* global,
                             const glabc_chains* c, const glabc_run* r)
{
    StepArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.prior = pack_dist<D>(&m->prior);
    for (int j = 0; j < D; ++j) {
        a.noise_loc[j] = m->noise.p0[j];
        a.noise_scale[j] = m->noise.p2[j];
        a.y_obs[j] = m->y_obs[j];
    }
    a.kern_log_scale = m->kern_log_scale;
    a.kern_scale = m->kern_scale;
    a.kern_c0 = m->kern_c0;
    a.local = pack_dist<D>(local ? local : global);
    a.global = pack_dist<D>(global);
    a.theta = c->theta;
    a.y = c->y;
    a.log_w = c->log_w;
    a.flags = c->flags;
    a.n_moves = c->n_moves;
    a.n_chains = c->n_chains;
    a.chain0 = c->chain0;
    a.stride = c->stride;
    if (r) {
        a.seed_lo = (uint32_t)r->seed;
        a.seed_hi = (uint32_t)(r->seed >> 32);
        a.step0 = r->step0;
        a.n_steps = r->n_steps;
        a.gf = r->global_frequency;
        a.history = r->history;
        a.hist_stride = r->hist_stride;
        if (r->moments) {
            a.sum_theta = r->moments->sum_theta;
            a.sum_outer = r->moments->sum_outer;
            a.sum_jump = r->moments->sum_jump;
        }
    }
    return a;
}

static int finish_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return GLABC_ERR_LAUNCH;
    }
    return GLABC_OK;
}

static unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

template <int ALGO, int D, int N>
static int launch_sampler(const StepArgs<D>& a, hipStream_t s)
{
    hipLaunchKernelGGL((sampler_kernel<ALGO, D, N>), dim3(grid_for(a.n_chains, BLOCK)), dim3(BLOCK), 0, s, a);
    return finish_launch();
}

template <int ALGO, int D>
static int dispatch_batch(const StepArgs<D>& a, int N, hipStream_t s)
{
    if constexpr (ALGO == ALGO_GLOBAL) {
        return launch_sampler<ALGO, D, 1>(a, s);
    } else {
        switch (N) {
#define GLABC_CASE(n) case n: return launch_sampler<ALGO, D, n>(a, s);
            GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
            GLABC_CASE(9) GLABC_CASE(10) GLABC_CASE(11) GLABC_CASE(12) GLABC_CASE(13) GLABC_CASE(14) GLABC_CASE(15) GLABC_CASE(16)
#undef GLABC_CASE
        default: return GLABC_ERR_ARG;
        }
    }
}

static int check_run(const glabc_model* m, const glabc_dist* local, const glabc_dist* global, const glabc_chains* c,
                     const glabc_run* r, bool isir)
{
    int rc = check_model(m);
    if (rc) return rc;
    rc = check_dist(local, m->theta_dim);
    if (rc) return rc;
    rc = check_dist(global, m->theta_dim);
    if (rc) return rc;
    if (!c || !r) return GLABC_ERR_NULL;
    if (!c->theta || !c->y) return GLABC_ERR_NULL;
    if (isir && (!c->log_w || !c->flags)) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains || c->chain0 < 0) return GLABC_ERR_ARG;
    if (r->n_steps < 0) return GLABC_ERR_ARG;
    if (!(r->global_frequency >= 0.0f) && !(r->global_frequency < 0.0f)) return GLABC_ERR_ARG;    // NaN
    if (isir && (r->batch_size < 1 || r->batch_size > GLABC_MAX_BATCH)) return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->moments && (!r->moments->sum_theta || !r->moments->sum_outer || !r->moments->sum_jump)) return GLABC_ERR_NULL;
    if (r->tape) return GLABC_ERR_ARG;       // tape replay is implemented by the CPU checker only (for now)
    if ((uint64_t)r->step0 + (uint64_t)r->n_steps > 0xFFFFFFFFull) return GLABC_ERR_ARG;
    return GLABC_OK;
}

template <int ALGO>
static int run_sampler(const glabc_model* m, const glabc_dist* local, const glabc_dist* global, const glabc_chains* c,
                       const glabc_run* r, void* stream)
{
    int rc = check_run(m, local, global, c, r, ALGO == ALGO_GLMCMC);
    if (rc) return rc;
    if (c->n_chains == 0 || r->n_steps == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (m->theta_dim) {
    case 1: return dispatch_batch<ALGO, 1>(pack_args<1>(m, local, global, c, r), r->batch_size, s);
    case 2: return dispatch_batch<ALGO, 2>(pack_args<2>(m, local, global, c, r), r->batch_size, s);
    case 3: return dispatch_batch<ALGO, 3>(pack_args<3>(m, local, global, c, r), r->batch_size, s);
    case 4: return dispatch_batch<ALGO, 4>(pack_args<4>(m, local, global, c, r), r->batch_size, s);
    default: return GLABC_ERR_DIM;
    }
}

template <int OP>
static int launch_rowwise(const RowArgs& a, hipStream_t s)
{
    dim3 grid(grid_for(a.n, 256)), block(256);
    switch (a.dim) {
#define GLABC_CASE(d) case d: hipLaunchKernelGGL((rowwise_kernel<OP, d>), grid, block, 0, s, a); break;
        GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
#undef GLABC_CASE
    default: return GLABC_ERR_DIM;
    }
    return finish_launch();
}

extern "C" {

__attribute__((visibility("default"))) int glabc_glmcmc_steps(const glabc_model* model, const glabc_dist* local,
                                                              const glabc_dist* importance, const glabc_chains* chains,
                                                              const glabc_run* run, void* stream)
{
    return run_sampler<ALGO_GLMCMC>(model, local, importance, chains, run, stream);
}

__attribute__((visibility("default"))) int glabc_globalmcmc_steps(const glabc_model* model, const glabc_dist* local,
                                                                  const glabc_dist* global, const glabc_chains* chains,
                                                                  const glabc_run* run, void* stream)
{
    return run_sampler<ALGO_GLOBAL>(model, local, global, chains, run, stream);
}

__attribute__((visibility("default"))) int glabc_init_weights(const glabc_model* model, const glabc_dist* importance,
                                                              const glabc_chains* c, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    rc = check_dist(importance, model->theta_dim);
    if (rc) return rc;
    if (!c || !c->theta || !c->y || !c->log_w || !c->flags) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains) return GLABC_ERR_ARG;
    if (c->n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(c->n_chains, BLOCK)), block(BLOCK);
    switch (model->theta_dim) {
    case 1: hipLaunchKernelGGL(init_weights_kernel<1>, grid, block, 0, s, pack_args<1>(model, nullptr, importance, c, nullptr)); break;
    case 2: hipLaunchKernelGGL(init_weights_kernel<2>, grid, block, 0, s, pack_args<2>(model, nullptr, importance, c, nullptr)); break;
    case 3: hipLaunchKernelGGL(init_weights_kernel<3>, grid, block, 0, s, pack_args<3>(model, nullptr, importance, c, nullptr)); break;
    case 4: hipLaunchKernelGGL(init_weights_kernel<4>, grid, block, 0, s, pack_args<4>(model, nullptr, importance, c, nullptr)); break;
    default: return GLABC_ERR_DIM;
    }
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_dist_log_prob(const glabc_dist* dist, const float* z, int64_t n, float* out,
                                                               void* stream)
{
    int rc = check_dist(dist, 0);
    if (rc) return rc;
    if (!z || !out) return GLABC_ERR_NULL;
    if (n < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    RowArgs a;
    std::memset(&a, 0, sizeof a);
    a.dist = *dist;
    a.dim = dist->dim;
    a.in = z;
    a.out = out;
    a.n = n;
    return launch_rowwise<ROW_DIST_LOG_PROB>(a, (hipStream_t)stream);
}

static int model_rowwise(const glabc_model* m, const float* in, int64_t n, float* out, void* stream, int op)
{
    int rc = check_model(m);
    if (rc) return rc;
    if (!in || !out) return GLABC_ERR_NULL;
    if (n < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    RowArgs a;
    std::memset(&a, 0, sizeof a);
    a.dist = m->prior;
    for (int j = 0; j < GLABC_MAX_DIM; ++j) a.y_obs[j] = m->y_obs[j];
    a.kern_log_scale = m->kern_log_scale;
    a.kern_scale = m->kern_scale;
    a.kern_c0 = m->kern_c0;
    a.dim = (op == ROW_PRIOR) ? m->theta_dim : m->y_dim;
    a.in = in;
    a.out = out;
    a.n = n;
    hipStream_t s = (hipStream_t)stream;
    if (op == ROW_PRIOR) return launch_rowwise<ROW_PRIOR>(a, s);
    if (op == ROW_DISCREPANCY) return launch_rowwise<ROW_DISCREPANCY>(a, s);
    return launch_rowwise<ROW_LOG_KERNEL>(a, s);
}

__attribute__((visibility("default"))) int glabc_model_prior_log_prob(const glabc_model* model, const float* theta, int64_t n,
                                                                      float* out, void* stream)
{
    return model_rowwise(model, theta, n, out, stream, ROW_PRIOR);
}

__attribute__((visibility("default"))) int glabc_model_discrepancy(const glabc_model* model, const float* y, int64_t n,
                                                                   float* out, void* stream)
{
    return model_rowwise(model, y, n, out, stream, ROW_DISCREPANCY);
}

__attribute__((visibility("default"))) int glabc_model_log_kernel(const glabc_model* model, const float* y, int64_t n,
                                                                  float* out, void* stream)
{
    return model_rowwise(model, y, n, out, stream, ROW_LOG_KERNEL);
}

__attribute__((visibility("default"))) int glabc_esjd(const float* history, int64_t n_rows, int32_t theta_dim,
                                                      int64_t n_chains, int64_t stride, float* esjd_out, void* stream)
{
    if (!history || !esjd_out) return GLABC_ERR_NULL;
    if (theta_dim < 1 || theta_dim > 4) return GLABC_ERR_DIM;
    if (n_rows < 2 || n_chains < 0 || stride < n_chains) return GLABC_ERR_ARG;
    if (n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n_chains, BLOCK)), block(BLOCK);
    switch (theta_dim) {
    case 1: hipLaunchKernelGGL(esjd_kernel<1>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 2: hipLaunchKernelGGL(esjd_kernel<2>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 3: hipLaunchKernelGGL(esjd_kernel<3>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 4: hipLaunchKernelGGL(esjd_kernel<4>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    }
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_moments_esjd(const glabc_moments* moments, int64_t n_steps, int32_t theta_dim,
                                                              int64_t n_chains, int64_t stride, float* esjd_out, void* stream)
{
    if (!moments || !moments->sum_jump || !esjd_out) return GLABC_ERR_NULL;
    if (theta_dim < 1 || theta_dim > 4) return GLABC_ERR_DIM;
    if (n_steps < 1 || n_chains < 0 || stride < n_chains) return GLABC_ERR_ARG;
    if (n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n_chains, BLOCK)), block(BLOCK);
    switch (theta_dim) {
    case 1: hipLaunchKernelGGL(moments_esjd_kernel<1>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 2: hipLaunchKernelGGL(moments_esjd_kernel<2>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 3: hipLaunchKernelGGL(moments_esjd_kernel<3>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 4: hipLaunchKernelGGL(moments_esjd_kernel<4>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    }
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_version(void) { return GLABC_VERSION; }

__attribute__((visibility("default"))) const char* glabc_status_string(int status)
{
    switch (status) {
    case GLABC_OK: return "ok";
    case GLABC_ERR_NULL: return "required pointer is NULL";
    case GLABC_ERR_DIM: return "dimension out of range or not compiled in";
    case GLABC_ERR_KIND: return "distribution / simulator kind not supported by this entry point";
    case GLABC_ERR_ARG: return "bad argument";
    case GLABC_ERR_LAUNCH: return "kernel launch failed";
    case GLABC_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown status";
    }
}

__attribute__((visibility("default"))) int glabc_last_hip_error(void) { return g_last_hip_error; }

}  // extern "C"
