// glabc_hip.hip -- gfx950 kernels and the C ABI of include/glabc.h.
//
// Build: csrc/Makefile (driven by __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c   (this file + glabc_sampler_dim.hip x 4 dims)
//   hipcc -shared *.o -o libglabc_hip.so
//
// Kernels
//   sampler_kernel<ALGO, D, N, L> fused K-iteration GLMCMC / GlobalMCMC step (glabc_sampler.h, one TU per D)
//   init_weights_kernel<D>       GLMCMC.py:52-55
//   rowwise_kernel<OP>           distribution / Model callbacks on row-major points
//   esjd_kernel<D>               ESJD.py:2-25 per chain from a chain-major history
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "glabc_mala.h"
#include "glabc_pack.h"
#include "glabc_sampler.h"
#include "glabc_team.h"

namespace glabc {

template <int D, int YD>
__global__ void __launch_bounds__(BLOCK) init_weights_kernel(const StepArgs<D, YD> a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_chains) return;
    Chain<D, YD> c;
#pragma unroll
    for (int j = 0; j < D; ++j) c.theta[j] = a.theta[j * a.stride + i];
#pragma unroll
    for (int j = 0; j < YD; ++j) c.y[j] = a.y[j * a.stride + i];
    refresh_cache<D, YD, true>(a, c);                              // Gamma importance proposal / prior included
    a.log_w[i] = (c.prior + c.kern) - c.q;                         // GLMCMC.py:52-55
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
    bool unit = g.kind == GLABC_DIST_DIAG_GAUSS;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        o.p0[j] = g.p0[j];
        o.p1[j] = g.p1[j];
        o.p2[j] = g.p2[j];
        o.p3[j] = g.p3[j];
        unit = unit && (g.p2[j] == 1.0f) && (g.p1[j] == 0.0f);
    }
    o.unit_scale = unit ? 1 : 0;
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
        r = dist_log_prob<D, false, true>(g, x);                   // DiagGaussian / Uniform / Gamma
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


// ---- numerics self-test hooks (device evaluation of include/glabc_numerics.h) -------------------
// op 0: expf  1: logf  2: sin(2 pi u)  3: cos(2 pi u)  4: normal_pair(a,b).z0  5: .z1 (in = 2 u32 words per item)
__global__ void __launch_bounds__(256) numerics_kernel(int op, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r;
    if (op == 0) r = glabc_expf(glabc_u2f(in[i]));
    else if (op == 1) r = glabc_logf(glabc_u2f(in[i]));
    else if (op == 2 || op == 3) {
        float sn, cs;
        glabc_sincos2pi(glabc_u2f(in[i]), &sn, &cs);
        r = op == 2 ? sn : cs;
    } else {
        float z0, z1;
        glabc_normal_pair(in[2 * i], in[2 * i + 1], &z0, &z1);
        r = op == 4 ? z0 : z1;
    }
    out[i] = glabc_f2u(r);
}

// glabc_sqrtf_normal against the exactly rounded (float)sqrt((double)x) for every float with bit
// pattern in [first, last]; counts mismatches (both are expected to be IEEE-correct on 0 and normals)
__global__ void __launch_bounds__(256) sqrt_check_kernel(uint32_t first, uint32_t last, unsigned long long* __restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long local = 0;
    for (uint64_t b = (uint64_t)first + (uint64_t)blockIdx.x * 256 + threadIdx.x; b <= (uint64_t)last; b += stride) {
        const float x = glabc_u2f((uint32_t)b);
        const float want = (float)__builtin_sqrt((double)x);
        local += (glabc_f2u(glabc_sqrtf_normal(x)) != glabc_f2u(want)) ? 1ull : 0ull;
    }
    if (local) atomicAdd(bad, local);
}

// ---- GLMCMC_NF: pool weights and one iteration against the pool (GLMCMC_NFs.py:73-111,141-152) ---------------
template <int D>
struct PoolArgs {
    StepArgs<D> s;
    const float* theta;       // forward: [D][n] proposals ; step: pool theta
    const float* x;
    const float* w;
    const float* log_q;
    int32_t* kk;
    float* x_out;
    float* w_out;
    int64_t n_rows, row_id0;
    int32_t step_size;
    int32_t* moved_idx;
    int32_t* n_moved;
    int32_t* n_moved_reset;
};

template <int D>
__global__ void __launch_bounds__(256) pool_weights_kernel(const PoolArgs<D> p)
{
    const StepArgs<D>& a = p.s;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= p.n_rows) return;
    float th[D], y[D], eps[D];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = p.theta[j * p.n_rows + r];
    // simulator noise: D normals from Philox blocks (row id, 0, b), pairs (2i, 2i+1)
    const uint64_t gid = (uint64_t)(p.row_id0 + r);
    constexpr int NB = (D + 3) / 4;
    float nrm[4 * NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, a.seed_lo, a.seed_hi);
        glabc_normal_pair(w.v[0], w.v[1], &nrm[4 * b], &nrm[4 * b + 1]);
        glabc_normal_pair(w.v[2], w.v[3], &nrm[4 * b + 2], &nrm[4 * b + 3]);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) eps[j] = nrm[j];
    model_simulate<D>(a, th, eps, y);                                                   // GLMCMC_NFs.py:79
    const float lw = (dist_log_prob<D>(a.prior, th) + model_log_kernel<D>(a, y)) - p.log_q[r];   // :80-81
    const float v = glabc_expf(lw);                                                     // :82
#pragma unroll
    for (int j = 0; j < D; ++j) p.x_out[j * p.n_rows + r] = y[j];
    p.w_out[r] = (v != v) ? 0.0f : v;                                                   // :83-85
}

// BaseDistribution.forward on the device (distribution.py:165-173, 73-78)
template <int D>
struct ForwardArgs {
    DistArgs<D> g;
    int64_t n, row0;
    uint32_t seed_lo, seed_hi;
    float* z;
    float* log_p;
};

template <int D>
__global__ void __launch_bounds__(256) dist_forward_kernel(const ForwardArgs<D> a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n) return;
    const uint64_t gid = (uint64_t)(a.row0 + r);
    constexpr int NB = (D + 3) / 4;
    float e[4 * NB], noise[D];
    const bool uni = a.g.kind == GLABC_DIST_UNIFORM;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, a.seed_lo, a.seed_hi);
        float n0, n1, n2, n3;
        glabc_normal_pair(w.v[0], w.v[1], &n0, &n1);
        glabc_normal_pair(w.v[2], w.v[3], &n2, &n3);
        e[4 * b] = uni ? glabc_uniform_f32(w.v[0]) : n0;
        e[4 * b + 1] = uni ? glabc_uniform_f32(w.v[1]) : n1;
        e[4 * b + 2] = uni ? glabc_uniform_f32(w.v[2]) : n2;
        e[4 * b + 3] = uni ? glabc_uniform_f32(w.v[3]) : n3;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        noise[j] = e[j];
        a.z[j * a.n + r] = a.g.p0[j] + a.g.p2[j] * e[j];                                // distribution.py:170 / :77
    }
    a.log_p[r] = dist_forward_log_p<D>(a.g, noise);
}

// AGLMCMC.py:104-109 / 199-204: pool weights from stored discrepancies under the threshold in a.kern_*
template <int D>
__global__ void __launch_bounds__(256) train_weights_kernel(const PoolArgs<D> p)
{
    const StepArgs<D>& a = p.s;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= p.n_rows) return;
    float th[D];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = p.theta[j * p.n_rows + r];
    const float e = (p.x[r] - 0.0f) / a.kern_scale;                                     // Mixture.py:50-52
    const float k = a.kern_c0 - (a.kern_log_scale + 0.5f * (e * e));
    const float v = glabc_expf((dist_log_prob<D>(a.prior, th) + k) - p.log_q[r]);       // AGLMCMC.py:200-201
    p.w_out[r] = (v != v) ? 0.0f : v;
}

template <int D, int N>
__global__ void __launch_bounds__(64) nf_step_kernel(const PoolArgs<D> p)
{
    const StepArgs<D>& a = p.s;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_chains) return;
    float th[D], y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        th[j] = a.theta[j * a.stride + i];
        y[j] = a.y[j * a.stride + i];
    }
    uint32_t n_moves = a.n_moves ? a.n_moves[i] : 0u;
    int32_t kk = p.kk[i];
    const uint64_t gid = (uint64_t)(a.chain0 + i);
    Rng rng;
    rng.c0 = (uint32_t)gid;
    rng.c1 = (uint32_t)(gid >> 32);
    rng.k0 = a.seed_lo;
    rng.k1 = a.seed_hi;
    const uint32_t step = a.step0;
    glabc_u32x4 h = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
    const float prior_old = dist_log_prob<D>(a.prior, th), kern_old = model_log_kernel<D>(a, y);
    bool moved = false;
    if (glabc_uniform_f32(h.v[0]) < a.gf) {                                             // GLMCMC_NFs.py:91-92
        float w[N + 1];
        w[0] = glabc_expf((prior_old + kern_old) - p.log_q[i]);                         // :99-101 (NaN is NOT zeroed here)
        const int64_t rows = (int64_t)p.step_size * N * a.n_chains;
        const int64_t base = ((int64_t)kk * N) * a.n_chains + i;                        // row (kk*N + j)*C + c
        const bool have = kk < p.step_size;
#pragma unroll
        for (int j = 0; j < N; ++j) w[j + 1] = have ? p.w[base + (int64_t)j * a.n_chains] : 0.0f;
        const float tot = aten_rowsum<N + 1>(w);                                        // :103
        const double u_res = glabc_uniform_f64(h.v[2], h.v[3]);
        int ind = -1;
        double run = 0.0;
#pragma unroll
        for (int k = 0; k <= N; ++k) {
            run += (double)(w[k] / tot);
            ind = (ind < 0 && u_res < run) ? k : ind;                                   // :104, :11-26
        }
        if (ind > 0 && have) {                                                          // :105-107
            const int64_t r = base + (int64_t)(ind - 1) * a.n_chains;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                th[j] = p.theta[j * rows + r];
                y[j] = p.x[j * rows + r];
            }
            moved = true;
        }
        kk += 1;                                                                        // :111
    } else {                                                                            // :141-152
        float e[D], s[D], tn[D], yn[D];
        constexpr int DP = D + (D & 1);                       // simulator normals start at an even word (glabc_device.h)
        constexpr int SPP = (DP + D + 3) / 4;
        uint32_t wd[4 * SPP];
#pragma unroll
        for (int b = 0; b < SPP; ++b) {
            glabc_u32x4 o = glabc_philox4x32_10(rng.c0, rng.c1, step, (uint32_t)(1 + b), rng.k0, rng.k1);
#pragma unroll
            for (int q = 0; q < 4; ++q) wd[4 * b + q] = o.v[q];
        }
        float nrm[4 * SPP];
#pragma unroll
        for (int q = 0; q < 2 * SPP; ++q) glabc_normal_pair(wd[2 * q], wd[2 * q + 1], &nrm[2 * q], &nrm[2 * q + 1]);
        const bool uni = a.local.kind == GLABC_DIST_UNIFORM;
#pragma unroll
        for (int q = 0; q < D; ++q) {
            e[q] = uni ? glabc_uniform_f32(wd[q]) : nrm[q];
            s[q] = nrm[DP + q];
            tn[q] = (a.local.p0[q] + a.local.p2[q] * e[q]) + th[q];                     // :142
        }
        model_simulate<D>(a, tn, s, yn);
        const float log_acc = ((dist_log_prob<D>(a.prior, tn) + model_log_kernel<D>(a, yn)) - prior_old) - kern_old;   // :145-146
        if (glabc_logf(glabc_uniform_f32(h.v[1])) < log_acc) {                          // :147-148
#pragma unroll
            for (int q = 0; q < D; ++q) {
                th[q] = tn[q];
                y[q] = yn[q];
            }
            moved = true;
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        a.theta[j * a.stride + i] = th[j];
        a.y[j * a.stride + i] = y[j];
        if (a.history) a.history[j * a.hist_stride + i] = th[j];
    }
    p.kk[i] = kk;
    if (a.n_moves) a.n_moves[i] = n_moves + (moved ? 1u : 0u);
    if (p.moved_idx && moved) p.moved_idx[atomicAdd(p.n_moved, 1)] = (int32_t)i;
    if (p.n_moved_reset && i == 0) *p.n_moved_reset = 0;
}

// ---- Gamma.log_prob, distribution.py:123-137 (float64) ----------------------------------------------------
struct GammaArgs {
    glabc_gamma g;
    const double* z;
    double* out;
    int64_t n;
};

// one coordinate's log(pdf), distribution.py:133-136
__device__ __forceinline__ double gamma_log_pdf(const glabc_gamma& g, int j, double z)
{
    const double x = z / g.scale[j];
    if (x >= 0.0) {                                        // scipy's support of gamma is closed at 0
        const double am1 = g.shape[j] - 1.0;
        const double xl = am1 == 0.0 ? 0.0 : am1 * glabc_log(x);                       // scipy.special.xlogy
        const double p = glabc_exp((xl - x) - g.gammaln[j]) / g.scale[j];               // gamma.pdf
        return p > 0.0 ? glabc_log(p) : -__builtin_inf();                               // distribution.py:136
    }
    return -__builtin_inf();
}

struct GammaFwdArgs {
    glabc_gamma g;
    double* z;
    double* log_p;
    int64_t n, row0;
    uint32_t seed_lo, seed_hi;
};

// Gamma.forward, distribution.py:106-121: one work-item per row
__global__ void __launch_bounds__(256) gamma_forward_kernel(const GammaFwdArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t gid = (uint64_t)(a.row0 + i);
    double acc = 0.0;
    for (int j = 0; j < a.g.dim; ++j) {
        const double z = glabc_gamma_draw(a.g.shape[j], (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)j, a.seed_lo, a.seed_hi) *
                         a.g.scale[j];
        a.z[i * a.g.dim + j] = z;
        const double lp = gamma_log_pdf(a.g, j, z);
        acc = j == 0 ? lp : acc + lp;
    }
    a.log_p[i] = acc;
}

__global__ void __launch_bounds__(256) gamma_log_prob_kernel(const GammaArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    double acc = 0.0;
    for (int j = 0; j < a.g.dim; ++j) {
        const double lp = gamma_log_pdf(a.g, j, a.z[i * a.g.dim + j]);
        acc = j == 0 ? lp : acc + lp;                                               // torch.sum(dim=1), distribution.py:137
    }
    a.out[i] = acc;
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

// allow_gamma: GLABC_DIST_GAMMA is known where include/glabc.h says so (importance / global proposal and prior of
// glabc_glmcmc_steps / glabc_globalmcmc_steps / glabc_init_weights, glabc_dist_log_prob, the row-wise Model callbacks)
static int check_dist(const glabc_dist* g, int dim, bool allow_gamma = false)
{
    if (!g) return GLABC_ERR_NULL;
    if (g->dim < 1 || g->dim > GLABC_MAX_DIM || (dim > 0 && g->dim != dim)) return GLABC_ERR_DIM;
    if (g->kind == GLABC_DIST_GAMMA) {
        if (!allow_gamma) return GLABC_ERR_KIND;
        for (int j = 0; j < g->dim; ++j)            // shape, rate, scale = 1/rate > 0 and finite; gammaln(shape) finite
            if (!(g->p0[j] > 0.0f) || !std::isfinite(g->p0[j]) || !(g->p1[j] > 0.0f) || !std::isfinite(g->p1[j]) ||
                !(g->p2[j] > 0.0f) || !std::isfinite(g->p2[j]) || !std::isfinite(g->p3[j]))
                return GLABC_ERR_ARG;
        return GLABC_OK;
    }
    if (g->kind != GLABC_DIST_DIAG_GAUSS && g->kind != GLABC_DIST_UNIFORM) return GLABC_ERR_KIND;
    if (!finite_dist(g)) return GLABC_ERR_ARG;
    if (g->kind == GLABC_DIST_DIAG_GAUSS)
        for (int j = 0; j < g->dim; ++j)
            if (!(g->p2[j] > 0.0f)) return GLABC_ERR_ARG;
    return GLABC_OK;
}

static int check_model(const glabc_model* m, bool allow_user_sim = false, bool allow_gamma_prior = false)
{
    if (!m) return GLABC_ERR_NULL;
    if (allow_user_sim && m->sim_kind == GLABC_SIM_USER) {            // row-wise callbacks: the simulator is not involved
        if (m->theta_dim < 1 || m->theta_dim > GLABC_MAX_DIM || m->y_dim < 1 || m->y_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
        int rc0 = check_dist(&m->prior, m->theta_dim, allow_gamma_prior);
        if (rc0) return rc0;
        if (!std::isfinite(m->kern_log_scale) || !(m->kern_scale > 0.0f) || !std::isfinite(m->kern_scale) || !std::isfinite(m->kern_c0))
            return GLABC_ERR_ARG;
        for (int j = 0; j < m->y_dim; ++j)
            if (!std::isfinite(m->y_obs[j])) return GLABC_ERR_ARG;
        return GLABC_OK;
    }
    if (m->sim_kind != GLABC_SIM_ABS_GAUSS && m->sim_kind != GLABC_SIM_GK) return GLABC_ERR_KIND;
    if (m->theta_dim < 1 || m->theta_dim > GLABC_MAX_DIM || m->y_dim < 1 || m->y_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    int rc = check_dist(&m->prior, m->theta_dim, allow_gamma_prior);
    if (rc) return rc;
    if (m->sim_kind == GLABC_SIM_GK) {
        if (m->theta_dim != 4 || m->y_dim != 8) return GLABC_ERR_DIM;          // the compiled g-and-k shape
        if (!std::isfinite(m->gk_c)) return GLABC_ERR_ARG;
    } else {
        if (m->y_dim != m->theta_dim) return GLABC_ERR_DIM;
        rc = check_dist(&m->noise, m->y_dim);
        if (rc) return rc;
        if (m->noise.kind != GLABC_DIST_DIAG_GAUSS) return GLABC_ERR_KIND;
    }
    if (!std::isfinite(m->kern_log_scale) || !(m->kern_scale > 0.0f) || !std::isfinite(m->kern_scale) ||
        !std::isfinite(m->kern_c0))
        return GLABC_ERR_ARG;
    for (int j = 0; j < m->y_dim; ++j)
        if (!std::isfinite(m->y_obs[j])) return GLABC_ERR_ARG;
    return GLABC_OK;
}

// RN(1/s) if the three-instruction division of model_log_kernel is exact for this divisor, else 0.  The check runs the
// device's instruction sequence (IEEE mul + two fused multiply-adds) over all 2^23 significands of the dividend and
// compares with the IEEE quotient (~7 ms, remembered per divisor); powers of two scale every step exactly.
__attribute__((target("fma"))) static bool reciprocal_division_exact(float s, float r)
{
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        const uint32_t bits = 0x3f800000u | m;
        float x;
        std::memcpy(&x, &bits, sizeof x);
        const float q = x * r;
        if (__builtin_fmaf(__builtin_fmaf(-q, s, x), r, q) != x / s) return false;
    }
    return true;
}

namespace glabc {
float verified_reciprocal(float s)
{
    if (!(s >= 0x1p-20f && s <= 0x1p20f) || !__builtin_cpu_supports("fma")) return 0.0f;
    static std::mutex lock;
    static float seen_s[16], seen_r[16];
    static int n_seen = 0, next = 0;
    std::lock_guard<std::mutex> guard(lock);
    for (int i = 0; i < n_seen; ++i)
        if (seen_s[i] == s) return seen_r[i];
    const float r = 1.0f / s;
    const float out = reciprocal_division_exact(s, r) ? r : 0.0f;
    seen_s[next] = s;
    seen_r[next] = out;
    next = (next + 1) % 16;
    if (n_seen < 16) ++n_seen;
    return out;
}
}  // namespace glabc

template <int D, int YD = D>
static StepArgs<D, YD> pack_args(const glabc_model* m, const glabc_dist* local, const glabc_dist* global,
                                 const glabc_chains* c, const glabc_run* r)
{
    return pack_args_rinv<D, YD>(m, local, global, c, r,
                                 (local && YD == D) ? verified_reciprocal(m->kern_scale) : 0.0f);      // sampler launches only
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

static int check_run(const glabc_model* m, const glabc_dist* local, const glabc_dist* global, const glabc_chains* c,
                     const glabc_run* r, bool isir)
{
    int rc = check_model(m, false, true);
    if (rc) return rc;
    rc = check_dist(local, m->theta_dim);
    if (rc) return rc;
    rc = check_dist(global, m->theta_dim, true);
    if (rc) return rc;
    const bool gamma = m->prior.kind == GLABC_DIST_GAMMA || global->kind == GLABC_DIST_GAMMA;
    if (gamma && (m->sim_kind != GLABC_SIM_ABS_GAUSS || m->theta_dim > 4)) return GLABC_ERR_KIND;    // the instantiated Gamma variants
    if (gamma && r && r->tape) return GLABC_ERR_ARG;                // a tape has no Gamma variates
    if (!c || !r) return GLABC_ERR_NULL;
    if (!c->theta || !c->y) return GLABC_ERR_NULL;
    if (isir && (!c->log_w || !c->flags)) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains || c->chain0 < 0) return GLABC_ERR_ARG;
    if (r->n_steps < 0) return GLABC_ERR_ARG;
    if (!(r->global_frequency >= 0.0f) && !(r->global_frequency < 0.0f)) return GLABC_ERR_ARG;    // NaN
    if (isir && (r->batch_size < 1 || r->batch_size > GLABC_MAX_BATCH_WIDE)) return GLABC_ERR_ARG;
    if (isir && r->batch_size > GLABC_MAX_BATCH) {               // the wide kernel: 8..64 lanes per chain, Philox only
        if (r->tape) return GLABC_ERR_ARG;
        if (r->lanes_per_chain != 0 && r->lanes_per_chain != 8 && r->lanes_per_chain != 16 && r->lanes_per_chain != 32 &&
            r->lanes_per_chain != 64)
            return GLABC_ERR_ARG;
    } else if (r->lanes_per_chain != 0 && r->lanes_per_chain != 1 && r->lanes_per_chain != 2 && r->lanes_per_chain != 4)
        return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->moments && (!r->moments->sum_theta || !r->moments->sum_outer || !r->moments->sum_jump)) return GLABC_ERR_NULL;
    if (r->tape) {                           // replayed random numbers: device arrays covering exactly this call
        if (!r->tape->u || !r->tape->z || (isir && !r->tape->r)) return GLABC_ERR_NULL;
        if (r->tape->n_prop < 1 || (isir && r->tape->n_prop < r->batch_size)) return GLABC_ERR_ARG;
    }
    if ((uint64_t)r->step0 + (uint64_t)r->n_steps > 0xFFFFFFFFull) return GLABC_ERR_ARG;
    if (r->step0_device) return GLABC_ERR_ARG;                  // the split-phase entry points only
    if (r->math_mode != GLABC_MATH_EXACT && r->math_mode != GLABC_MATH_FAST) return GLABC_ERR_ARG;
    if (r->math_mode == GLABC_MATH_FAST) {                      // the opt-in variant: where a team kernel exists (include/glabc.h)
        if (!isir || r->tape || gamma || m->sim_kind != GLABC_SIM_ABS_GAUSS || m->theta_dim > 4 || r->batch_size < 2 ||
            r->batch_size > GLABC_MAX_BATCH || r->lanes_per_chain != 0)
            return GLABC_ERR_ARG;
        if (r->dump_draws && (!r->dump_draws->u || !r->dump_draws->r || !r->dump_draws->z)) return GLABC_ERR_NULL;
    } else if (r->dump_draws) {
        return GLABC_ERR_ARG;
    }
    return GLABC_OK;
}

// lanes per chain: a launch-geometry choice (results do not depend on it).  Measured on MI355X
// (profiles/): with the branch-free candidate code one wave per SIMD already interleaves its N
// independent candidates, and one work-item per chain is fastest at 65 536 chains for N = 5
// (6.3 ms / 2000 iterations vs 6.8 ms with 2 lanes, 9.2 ms with 4); the split only pays when a
// launch would otherwise leave SIMDs empty (fewer chains than lanes on the chip).
static int pick_lanes(int requested, int n_batch, int64_t n_chains)
{
    int lanes = requested;
    if (lanes <= 0) {
        const int64_t chip_lanes = 64 * 1024;                // one wave on each of the 1024 SIMDs
        lanes = 1;
        while (lanes < 4 && n_chains * lanes < chip_lanes) lanes *= 2;
    }
    if (lanes >= 4 && n_batch >= 3) return 4;
    if (lanes >= 2 && n_batch >= 2) return 2;
    return 1;
}

static int run_sampler(int algo, const glabc_model* m, const glabc_dist* local, const glabc_dist* global,
                       const glabc_chains* c, const glabc_run* r, void* stream)
{
    int rc = check_run(m, local, global, c, r, algo == ALGO_GLMCMC);
    if (rc) return rc;
    if (c->n_chains == 0 || r->n_steps == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (algo == ALGO_GLMCMC && r->batch_size > GLABC_MAX_BATCH) {            // glabc_wide.hip
        if (m->sim_kind == GLABC_SIM_GK) return launch_wide<4, 8>(pack_args<4, 8>(m, local, global, c, r), r->batch_size, r->lanes_per_chain, s);
        switch (m->theta_dim) {
        case 1: return launch_wide<1, 1>(pack_args<1>(m, local, global, c, r), r->batch_size, r->lanes_per_chain, s);
        case 2: return launch_wide<2, 2>(pack_args<2>(m, local, global, c, r), r->batch_size, r->lanes_per_chain, s);
        case 3: return launch_wide<3, 3>(pack_args<3>(m, local, global, c, r), r->batch_size, r->lanes_per_chain, s);
        case 4: return launch_wide<4, 4>(pack_args<4>(m, local, global, c, r), r->batch_size, r->lanes_per_chain, s);
        default: return GLABC_ERR_DIM;
        }
    }
    // Team geometry (glabc_team.h): two wavefronts per 64 chains, for launches that would otherwise leave the SIMDs with at most
    // two wavefronts of sampler_kernel each.  Chosen when the caller leaves the geometry to the library.
    const bool gamma = m->prior.kind == GLABC_DIST_GAMMA || global->kind == GLABC_DIST_GAMMA;      // VAR_GAMMA: one lane per chain, or a team of two / three wavefronts
    const bool fast = r->math_mode == GLABC_MATH_FAST;      // runs the team kernels whatever the launch size
    if (algo == ALGO_GLMCMC && !r->tape && !(gamma && (fast || m->sim_kind != GLABC_SIM_ABS_GAUSS)) && (fast || !(r->debug_flags & GLABC_DEBUG_NO_TEAM)) &&
        (fast || (r->debug_flags & GLABC_DEBUG_TEAM) || (r->lanes_per_chain == 0 && c->n_chains >= 64 * 256 && c->n_chains <= 2 * 1024 * 64))) {
        int prio = 1;                                       // the main wavefront carries the serial part of an iteration
        if (const char* e = std::getenv("GLABC_TEAM_PRIO")) prio = std::max(0, std::min(3, std::atoi(e)));
        // wavefronts per 64 chains: enough for about three wavefronts per SIMD (1024 SIMDs)
        const int64_t groups = (c->n_chains + 63) / 64;
        int nw = groups <= 1024 ? 3 : 2;
        if (const char* e = std::getenv("GLABC_TEAM_WAVES")) nw = std::max(2, std::min(4, std::atoi(e)));
        rc = GLABC_ERR_ARG;
        for (; nw >= 2 && rc == GLABC_ERR_ARG; --nw) {      // fewer wavefronts when the batch is too small to split that far
            if (m->sim_kind == GLABC_SIM_GK) {
                rc = launch_team_dim<4, 8>(r->batch_size, nw, pack_args<4, 8>(m, local, global, c, r), prio, fast, s);
            } else {
                switch (m->theta_dim) {
#define GLABC_TEAM_CASE(d) case d: rc = launch_team_dim<d, d>(r->batch_size, nw, pack_args<d>(m, local, global, c, r), prio, fast, s); break;
                    GLABC_TEAM_CASE(1) GLABC_TEAM_CASE(2) GLABC_TEAM_CASE(3) GLABC_TEAM_CASE(4)
#undef GLABC_TEAM_CASE
                default: nw = 0; break;
                }
            }
        }
        if (rc != GLABC_ERR_ARG || fast) {                  // GLABC_ERR_ARG: no team kernel for this configuration (fast math: refused)
            if (rc == GLABC_ERR_LAUNCH) g_last_hip_error = (int)hipPeekAtLastError();
            return rc;
        }
    }
    // GlobalMCMC: a team of two wavefronts per 64 chains (global_team_kernel, glabc_team.h: the helper draws an iteration's random
    // numbers one iteration ahead), for the same launch sizes and under the same debug bits
    if (algo == ALGO_GLOBAL && !r->tape && !gamma && !(r->debug_flags & GLABC_DEBUG_NO_TEAM) &&
        ((r->debug_flags & GLABC_DEBUG_TEAM) || (r->lanes_per_chain == 0 && c->n_chains >= 64 * 256 && c->n_chains <= 2 * 1024 * 64))) {
        rc = GLABC_ERR_ARG;
        int gnw = 2;                                        // (three -- the helper's work split once more -- measured slower: 1.22 against 1.17 ms)
        if (const char* e = std::getenv("GLABC_TEAM_WAVES")) gnw = std::atoi(e) >= 3 ? 3 : 2;
        if (m->sim_kind == GLABC_SIM_GK) {
            rc = launch_global_team_dim<4, 8>(gnw, pack_args<4, 8>(m, local, global, c, r), 1, s);
        } else {
            switch (m->theta_dim) {
#define GLABC_TEAM_CASE(d) case d: rc = launch_global_team_dim<d, d>(gnw, pack_args<d>(m, local, global, c, r), 1, s); break;
                GLABC_TEAM_CASE(1) GLABC_TEAM_CASE(2) GLABC_TEAM_CASE(3) GLABC_TEAM_CASE(4)
#undef GLABC_TEAM_CASE
            default: break;
            }
        }
        if (rc != GLABC_ERR_ARG) {
            if (rc == GLABC_ERR_LAUNCH) g_last_hip_error = (int)hipPeekAtLastError();
            return rc;
        }
    }
    const int lanes = (algo == ALGO_GLMCMC && !r->tape && !gamma) ? pick_lanes(r->lanes_per_chain, r->batch_size, c->n_chains) : 1;
    // Two builds of the same kernels: up to two waves per SIMD (131 072 lanes on this part) a launch is latency-bound
    // and runs the max-ilp schedule (217 VGPRs, 6 % faster at 65 536 chains); larger launches need the occupancy
    // of the default schedule (126 VGPRs).  The tape variant and lane groups exist in the default objects only.
    const bool ilp = lanes == 1 && !r->tape && !gamma && c->n_chains <= 2 * 1024 * 64 && !(r->debug_flags & GLABC_DEBUG_DEFAULT_SCHEDULE);
    if (m->sim_kind == GLABC_SIM_GK) {
        rc = launch_sampler_dim<4, 8, SCHED_DEFAULT>(algo, r->batch_size, lanes, pack_args<4, 8>(m, local, global, c, r), s);
    } else {
#define GLABC_DIM_CASE(d)                                                                                             \
    case d:                                                                                                           \
        rc = ilp ? launch_sampler_dim<d, d, SCHED_ILP>(algo, r->batch_size, lanes, pack_args<d>(m, local, global, c, r), s) \
                 : launch_sampler_dim<d, d, SCHED_DEFAULT>(algo, r->batch_size, lanes, pack_args<d>(m, local, global, c, r), s); \
        break;
        // theta_dim 5..8: default-schedule objects only; a lane keeps (2 theta_dim + 4) registers per candidate, so the
        // candidates are dealt to 2 / 4 lanes as soon as there are that many (the arrays would leave the registers otherwise)
#define GLABC_HI_CASE(d)                                                                                               \
    case d: {                                                                                                         \
        const int hl = (algo != ALGO_GLMCMC || r->tape) ? 1 : r->lanes_per_chain ? lanes : (r->batch_size >= 3 ? 4 : r->batch_size); \
        rc = launch_sampler_dim<d, d, SCHED_DEFAULT>(algo, r->batch_size, hl, pack_args<d>(m, local, global, c, r), s); \
    } break;
        switch (m->theta_dim) {
            GLABC_DIM_CASE(1) GLABC_DIM_CASE(2) GLABC_DIM_CASE(3) GLABC_DIM_CASE(4)
            GLABC_HI_CASE(5) GLABC_HI_CASE(6) GLABC_HI_CASE(7) GLABC_HI_CASE(8)
        default: return GLABC_ERR_DIM;
        }
#undef GLABC_HI_CASE
#undef GLABC_DIM_CASE
    }
    if (rc == GLABC_ERR_LAUNCH) g_last_hip_error = (int)hipPeekAtLastError();
    return rc;
}

template <int D>
static MalaArgs<D> pack_mala(const glabc_model* m, const glabc_dist* imp, const glabc_mala* p, const glabc_chains* c,
                             const glabc_run* r)
{
    MalaArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.s = pack_args<D>(m, nullptr, imp ? imp : &m->prior, c, r);
    a.theta64 = c->theta64;
    a.y64 = c->y64;
    a.log_w64 = c->log_w64;
    a.grad = c->grad;
    if (p) {
        a.tau = p->tau;
        a.tau_sq = p->tau_sq;
        a.eps_sq = p->eps_sq;
        a.num_grad = p->num_grad;
    }
    a.lanes = r ? r->lanes_per_chain : 0;
    // team kernel (glabc_mala.h): the main wavefront runs at s_setprio 1 -- it carries the serial part of an iteration, the helper
    // fills the issue slots it leaves (37.9 against 43.9 ms per 2000 iterations of 65 536 chains) -- and takes the same share of
    // the gradient items as a helper lane (credit 0; measured optimum, flat between -2 and +2).  GLABC_MALA_PRIO /
    // GLABC_MALA_CREDIT override both for tuning runs -- execution strategy only, never results
    a.credit = 0;
    if (const char* e = std::getenv("GLABC_MALA_CREDIT")) a.credit = std::max(-64, std::min(64, std::atoi(e)));
    a.prio = 1;
    if (const char* e = std::getenv("GLABC_MALA_PRIO")) a.prio = std::max(0, std::min(3, std::atoi(e)));
    return a;
}

static int check_mala_chains(const glabc_chains* c)
{
    if (!c) return GLABC_ERR_NULL;
    if (!c->theta || !c->y || !c->flags || !c->theta64 || !c->y64 || !c->log_w64 || !c->grad) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains || c->chain0 < 0) return GLABC_ERR_ARG;
    return GLABC_OK;
}

extern "C" {

__attribute__((visibility("default"))) int glabc_glmala_steps(const glabc_model* model, const glabc_dist* importance,
                                                              const glabc_mala* mala, const glabc_chains* c,
                                                              const glabc_run* r, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    if (model->sim_kind != GLABC_SIM_ABS_GAUSS) return GLABC_ERR_KIND;
    rc = check_dist(importance, model->theta_dim);
    if (rc) return rc;
    if (!mala || !r) return GLABC_ERR_NULL;
    rc = check_mala_chains(c);
    if (rc) return rc;
    if (r->n_steps < 0 || r->batch_size < 1 || r->batch_size > GLABC_MAX_BATCH) return GLABC_ERR_ARG;
    if (!(r->global_frequency >= 0.0f) && !(r->global_frequency < 0.0f)) return GLABC_ERR_ARG;
    if (r->global_frequency_per_chain) return GLABC_ERR_ARG;        // GLMCMC / GlobalMCMC only
    if (mala->num_grad < 2 || mala->num_grad > (1 << 16) || !(mala->tau > 0.0) || !std::isfinite(mala->tau) ||
        !std::isfinite(mala->tau_sq) || !std::isfinite(mala->eps_sq) || !(mala->eps_sq >= 0.0))
        return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->moments && (!r->moments->sum_theta || !r->moments->sum_outer || !r->moments->sum_jump)) return GLABC_ERR_NULL;
    if (r->tape || r->math_mode != GLABC_MATH_EXACT || r->dump_draws) return GLABC_ERR_ARG;
    if (r->lanes_per_chain < 0 || r->lanes_per_chain > 2) return GLABC_ERR_ARG;      // wavefronts per 64 chains (theta_dim 2)
    if ((uint64_t)r->step0 + (uint64_t)r->n_steps > 0xFFFFFFFFull) return GLABC_ERR_ARG;
    if (c->n_chains == 0 || r->n_steps == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (model->theta_dim) {
    case 1: rc = launch_glmala_dim<1>(r->batch_size, pack_mala<1>(model, importance, mala, c, r), s); break;
    case 2: rc = launch_glmala_dim<2>(r->batch_size, pack_mala<2>(model, importance, mala, c, r), s); break;
    case 3: rc = launch_glmala_dim<3>(r->batch_size, pack_mala<3>(model, importance, mala, c, r), s); break;
    case 4: rc = launch_glmala_dim<4>(r->batch_size, pack_mala<4>(model, importance, mala, c, r), s); break;
    default: return GLABC_ERR_DIM;
    }
    if (rc == GLABC_ERR_LAUNCH) g_last_hip_error = (int)hipPeekAtLastError();
    return rc;
}

__attribute__((visibility("default"))) int glabc_glmala_init(const glabc_model* model, const glabc_chains* c, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    rc = check_mala_chains(c);
    if (rc) return rc;
    if (c->n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (model->theta_dim) {
    case 1: rc = launch_glmala_init_dim<1>(pack_mala<1>(model, nullptr, nullptr, c, nullptr), s); break;
    case 2: rc = launch_glmala_init_dim<2>(pack_mala<2>(model, nullptr, nullptr, c, nullptr), s); break;
    case 3: rc = launch_glmala_init_dim<3>(pack_mala<3>(model, nullptr, nullptr, c, nullptr), s); break;
    case 4: rc = launch_glmala_init_dim<4>(pack_mala<4>(model, nullptr, nullptr, c, nullptr), s); break;
    default: return GLABC_ERR_DIM;
    }
    if (rc == GLABC_ERR_LAUNCH) g_last_hip_error = (int)hipPeekAtLastError();
    return rc;
}

}  // extern "C"

template <int D>
static int launch_pool_weights(const PoolArgs<D>& p, hipStream_t s)
{
    hipLaunchKernelGGL((pool_weights_kernel<D>), dim3(grid_for(p.n_rows, 256)), dim3(256), 0, s, p);
    return finish_launch();
}

template <int D>
static int launch_nf_step(const PoolArgs<D>& p, int N, hipStream_t s)
{
    dim3 grid(grid_for(p.s.n_chains, 64)), block(64);
    switch (N) {
#define GLABC_CASE(n) case n: hipLaunchKernelGGL((nf_step_kernel<D, n>), grid, block, 0, s, p); break;
        GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
        GLABC_CASE(9) GLABC_CASE(10) GLABC_CASE(11) GLABC_CASE(12) GLABC_CASE(13) GLABC_CASE(14) GLABC_CASE(15) GLABC_CASE(16)
#undef GLABC_CASE
    default: return GLABC_ERR_ARG;
    }
    return finish_launch();
}

extern "C" {

__attribute__((visibility("default"))) int glabc_pool_weights(const glabc_model* model, const float* theta, const float* log_q,
                                                              int64_t n_rows, uint64_t seed, int64_t row_id0, float* x_out,
                                                              float* w_out, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    if (model->sim_kind != GLABC_SIM_ABS_GAUSS) return GLABC_ERR_KIND;
    if (!theta || !log_q || !x_out || !w_out) return GLABC_ERR_NULL;
    if (n_rows < 0 || row_id0 < 0) return GLABC_ERR_ARG;
    if (n_rows == 0) return GLABC_OK;
    glabc_chains dummy;
    std::memset(&dummy, 0, sizeof dummy);
    glabc_run r;
    std::memset(&r, 0, sizeof r);
    r.seed = seed;
    hipStream_t s = (hipStream_t)stream;
#define GLABC_POOLW(d)                                                        \
    case d: {                                                                 \
        PoolArgs<d> p;                                                        \
        std::memset(&p, 0, sizeof p);                                         \
        p.s = pack_args<d>(model, nullptr, &model->prior, &dummy, &r);        \
        p.theta = theta; p.log_q = log_q; p.x_out = x_out; p.w_out = w_out;   \
        p.n_rows = n_rows; p.row_id0 = row_id0;                               \
        return launch_pool_weights<d>(p, s);                                  \
    }
    switch (model->theta_dim) {
        GLABC_POOLW(1) GLABC_POOLW(2) GLABC_POOLW(3) GLABC_POOLW(4)
    default: return GLABC_ERR_DIM;
    }
#undef GLABC_POOLW
}

__attribute__((visibility("default"))) int glabc_dist_forward(const glabc_dist* dist, int64_t n, uint64_t seed, int64_t row0,
                                                              float* z_out, float* log_p_out, void* stream)
{
    if (!dist) return GLABC_ERR_NULL;
    int rc = check_dist(dist, dist->dim);
    if (rc) return rc;
    if (n < 0 || row0 < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    if (!z_out || !log_p_out) return GLABC_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
#define GLABC_FWD(d)                                                          \
    case d: {                                                                 \
        ForwardArgs<d> a;                                                     \
        std::memset(&a, 0, sizeof a);                                         \
        a.g = pack_dist<d>(dist);                                             \
        a.n = n; a.row0 = row0; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); \
        a.z = z_out; a.log_p = log_p_out;                                     \
        hipLaunchKernelGGL((dist_forward_kernel<d>), dim3(grid_for(n, 256)), dim3(256), 0, s, a); \
        return finish_launch();                                               \
    }
    switch (dist->dim) {
        GLABC_FWD(1) GLABC_FWD(2) GLABC_FWD(3) GLABC_FWD(4)
    default: return GLABC_ERR_DIM;
    }
#undef GLABC_FWD
}

__attribute__((visibility("default"))) int glabc_kde_train_weights(const glabc_model* model, const float* theta, const float* dis,
                                                                   const float* log_q, int64_t n, float* w_out, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    if (!theta || !dis || !log_q || !w_out) return GLABC_ERR_NULL;
    if (n < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    glabc_chains dummy;
    std::memset(&dummy, 0, sizeof dummy);
    glabc_run r;
    std::memset(&r, 0, sizeof r);
    hipStream_t s = (hipStream_t)stream;
#define GLABC_TRAINW(d)                                                       \
    case d: {                                                                 \
        PoolArgs<d> p;                                                        \
        std::memset(&p, 0, sizeof p);                                         \
        p.s = pack_args<d>(model, nullptr, &model->prior, &dummy, &r);        \
        p.theta = theta; p.x = dis; p.log_q = log_q; p.w_out = w_out;         \
        p.n_rows = n;                                                         \
        hipLaunchKernelGGL((train_weights_kernel<d>), dim3(grid_for(n, 256)), dim3(256), 0, s, p); \
        return finish_launch();                                               \
    }
    switch (model->theta_dim) {
        GLABC_TRAINW(1) GLABC_TRAINW(2) GLABC_TRAINW(3) GLABC_TRAINW(4)
    default: return GLABC_ERR_DIM;
    }
#undef GLABC_TRAINW
}

__attribute__((visibility("default"))) int glabc_glmcmc_nf_step(const glabc_model* model, const glabc_dist* local,
                                                                const glabc_pool* pool, const glabc_chains* c,
                                                                const glabc_run* r, void* stream)
{
    int rc = check_model(model);
    if (rc) return rc;
    rc = check_dist(local, model->theta_dim);
    if (rc) return rc;
    if (model->sim_kind != GLABC_SIM_ABS_GAUSS) return GLABC_ERR_KIND;
    if (!pool || !c || !r) return GLABC_ERR_NULL;
    if (!pool->theta || !pool->x || !pool->w || !pool->log_q_old || !pool->kk || !c->theta || !c->y) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains || c->chain0 < 0 || pool->step_size < 1) return GLABC_ERR_ARG;
    if (r->n_steps != 1 || r->batch_size < 1 || r->batch_size > GLABC_MAX_BATCH) return GLABC_ERR_ARG;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->tape || r->moments || r->global_frequency_per_chain || r->math_mode != GLABC_MATH_EXACT || r->dump_draws) return GLABC_ERR_ARG;
    if ((pool->moved_idx == nullptr) != (pool->n_moved == nullptr)) return GLABC_ERR_NULL;
    if (c->n_chains > 0x7fffffff) return GLABC_ERR_ARG;
    if (c->n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
#define GLABC_NFSTEP(d)                                                       \
    case d: {                                                                 \
        PoolArgs<d> p;                                                        \
        std::memset(&p, 0, sizeof p);                                         \
        p.s = pack_args<d>(model, local, &model->prior, c, r);                \
        p.theta = pool->theta; p.x = pool->x; p.w = pool->w; p.log_q = pool->log_q_old; p.kk = pool->kk; \
        p.step_size = pool->step_size;                                        \
        p.moved_idx = pool->moved_idx; p.n_moved = pool->n_moved; p.n_moved_reset = pool->n_moved_reset; \
        return launch_nf_step<d>(p, r->batch_size, s);                        \
    }
    switch (model->theta_dim) {
        GLABC_NFSTEP(1) GLABC_NFSTEP(2) GLABC_NFSTEP(3) GLABC_NFSTEP(4)
    default: return GLABC_ERR_DIM;
    }
#undef GLABC_NFSTEP
}

}  // extern "C"

extern "C" __attribute__((visibility("default"))) int glabc_gamma_log_prob(const glabc_gamma* dist, const double* z, int64_t n,
                                                                           double* out, void* stream)
{
    if (!dist || !z || !out) return GLABC_ERR_NULL;
    if (dist->dim < 1 || dist->dim > 3) return GLABC_ERR_DIM;        // < 4 terms: torch's float64 row sum is sequential
    if (n < 0) return GLABC_ERR_ARG;
    for (int j = 0; j < dist->dim; ++j)
        if (!(dist->shape[j] > 0.0) || !(dist->scale[j] > 0.0) || !std::isfinite(dist->gammaln[j])) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    GammaArgs a;
    a.g = *dist;
    a.z = z;
    a.out = out;
    a.n = n;
    hipLaunchKernelGGL(gamma_log_prob_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return finish_launch();
}

extern "C" __attribute__((visibility("default"))) int glabc_gamma_forward(const glabc_gamma* dist, int64_t n, uint64_t seed,
                                                                          int64_t row0, double* z_out, double* log_p_out, void* stream)
{
    if (!dist || !z_out || !log_p_out) return GLABC_ERR_NULL;
    if (dist->dim < 1 || dist->dim > 3) return GLABC_ERR_DIM;
    if (n < 0 || row0 < 0) return GLABC_ERR_ARG;
    for (int j = 0; j < dist->dim; ++j)
        if (!(dist->shape[j] > 0.0) || !std::isfinite(dist->shape[j]) || !(dist->scale[j] > 0.0) || !std::isfinite(dist->gammaln[j]))
            return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    GammaFwdArgs a;
    a.g = *dist;
    a.z = z_out;
    a.log_p = log_p_out;
    a.n = n;
    a.row0 = row0;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    hipLaunchKernelGGL(gamma_forward_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return finish_launch();
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
    return run_sampler(ALGO_GLMCMC, model, local, importance, chains, run, stream);
}

__attribute__((visibility("default"))) int glabc_globalmcmc_steps(const glabc_model* model, const glabc_dist* local,
                                                                  const glabc_dist* global, const glabc_chains* chains,
                                                                  const glabc_run* run, void* stream)
{
    return run_sampler(ALGO_GLOBAL, model, local, global, chains, run, stream);
}

__attribute__((visibility("default"))) int glabc_init_weights(const glabc_model* model, const glabc_dist* importance,
                                                              const glabc_chains* c, void* stream)
{
    int rc = check_model(model, false, true);
    if (rc) return rc;
    rc = check_dist(importance, model->theta_dim, true);
    if (rc) return rc;
    if (!c || !c->theta || !c->y || !c->log_w || !c->flags) return GLABC_ERR_NULL;
    if (c->n_chains < 0 || c->stride < c->n_chains) return GLABC_ERR_ARG;
    if (c->n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(c->n_chains, BLOCK)), block(BLOCK);
    if (model->sim_kind == GLABC_SIM_GK) {
        hipLaunchKernelGGL((init_weights_kernel<4, 8>), grid, block, 0, s, pack_args<4, 8>(model, nullptr, importance, c, nullptr));
        return finish_launch();
    }
    switch (model->theta_dim) {
    case 1: hipLaunchKernelGGL((init_weights_kernel<1, 1>), grid, block, 0, s, pack_args<1>(model, nullptr, importance, c, nullptr)); break;
    case 2: hipLaunchKernelGGL((init_weights_kernel<2, 2>), grid, block, 0, s, pack_args<2>(model, nullptr, importance, c, nullptr)); break;
    case 3: hipLaunchKernelGGL((init_weights_kernel<3, 3>), grid, block, 0, s, pack_args<3>(model, nullptr, importance, c, nullptr)); break;
    case 4: hipLaunchKernelGGL((init_weights_kernel<4, 4>), grid, block, 0, s, pack_args<4>(model, nullptr, importance, c, nullptr)); break;
    case 5: hipLaunchKernelGGL((init_weights_kernel<5, 5>), grid, block, 0, s, pack_args<5>(model, nullptr, importance, c, nullptr)); break;
    case 6: hipLaunchKernelGGL((init_weights_kernel<6, 6>), grid, block, 0, s, pack_args<6>(model, nullptr, importance, c, nullptr)); break;
    case 7: hipLaunchKernelGGL((init_weights_kernel<7, 7>), grid, block, 0, s, pack_args<7>(model, nullptr, importance, c, nullptr)); break;
    case 8: hipLaunchKernelGGL((init_weights_kernel<8, 8>), grid, block, 0, s, pack_args<8>(model, nullptr, importance, c, nullptr)); break;
    default: return GLABC_ERR_DIM;
    }
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_dist_log_prob(const glabc_dist* dist, const float* z, int64_t n, float* out,
                                                               void* stream)
{
    int rc = check_dist(dist, 0, true);
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
    int rc = check_model(m, true, true);
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
    if (theta_dim < 1 || theta_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (n_rows < 2 || n_chains < 0 || stride < n_chains) return GLABC_ERR_ARG;
    if (n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n_chains, BLOCK)), block(BLOCK);
    switch (theta_dim) {
    case 1: hipLaunchKernelGGL(esjd_kernel<1>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 2: hipLaunchKernelGGL(esjd_kernel<2>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 3: hipLaunchKernelGGL(esjd_kernel<3>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 4: hipLaunchKernelGGL(esjd_kernel<4>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 5: hipLaunchKernelGGL(esjd_kernel<5>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 6: hipLaunchKernelGGL(esjd_kernel<6>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 7: hipLaunchKernelGGL(esjd_kernel<7>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    case 8: hipLaunchKernelGGL(esjd_kernel<8>, grid, block, 0, s, history, n_rows, n_chains, stride, esjd_out); break;
    }
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_moments_esjd(const glabc_moments* moments, int64_t n_steps, int32_t theta_dim,
                                                              int64_t n_chains, int64_t stride, float* esjd_out, void* stream)
{
    if (!moments || !moments->sum_jump || !esjd_out) return GLABC_ERR_NULL;
    if (theta_dim < 1 || theta_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (n_steps < 1 || n_chains < 0 || stride < n_chains) return GLABC_ERR_ARG;
    if (n_chains == 0) return GLABC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_for(n_chains, BLOCK)), block(BLOCK);
    switch (theta_dim) {
    case 1: hipLaunchKernelGGL(moments_esjd_kernel<1>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 2: hipLaunchKernelGGL(moments_esjd_kernel<2>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 3: hipLaunchKernelGGL(moments_esjd_kernel<3>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 4: hipLaunchKernelGGL(moments_esjd_kernel<4>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 5: hipLaunchKernelGGL(moments_esjd_kernel<5>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 6: hipLaunchKernelGGL(moments_esjd_kernel<6>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 7: hipLaunchKernelGGL(moments_esjd_kernel<7>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    case 8: hipLaunchKernelGGL(moments_esjd_kernel<8>, grid, block, 0, s, moments->sum_jump, n_steps, n_chains, stride, esjd_out); break;
    }
    return finish_launch();
}

/* Test hooks: evaluate include/glabc_numerics.h ON THE DEVICE so the tests can require bit equality
 * with the host evaluation (the premise of every bit-parity claim).  Not part of the sampling API. */
__attribute__((visibility("default"))) int glabc_selftest_numerics(int op, const uint32_t* in, uint32_t* out, int64_t n, void* stream)
{
    if (!in || !out) return GLABC_ERR_NULL;
    if (op < 0 || op > 5 || n < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    hipLaunchKernelGGL(numerics_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, op, in, out, n);
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_selftest_sqrt(uint32_t first_bits, uint32_t last_bits, uint64_t* mismatches, void* stream)
{
    if (!mismatches) return GLABC_ERR_NULL;
    if (last_bits < first_bits) return GLABC_ERR_ARG;
    hipLaunchKernelGGL(sqrt_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, first_bits, last_bits,
                       (unsigned long long*)mismatches);
    return finish_launch();
}

__attribute__((visibility("default"))) int glabc_version(void) { return GLABC_VERSION; }
__attribute__((visibility("default"))) int glabc_stream_layout(void) { return GLABC_STREAM_LAYOUT; }

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
