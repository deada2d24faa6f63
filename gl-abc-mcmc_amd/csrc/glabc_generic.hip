// glabc_generic.hip -- the split-phase iteration for Models whose callbacks are arbitrary code (include/glabc.h,
// "split-phase iteration"): glabc_propose, glabc_propose_redraw, glabc_select, glabc_model_simulate.
//
// The fused sampler kernels (glabc_sampler.h) need the Model as numbers (a glabc_model).  The reference's plug-in API is
// the duck-typed Model protocol (examples/Mixture.py:5-53): any object with generate_samples / prior_log_prob /
// calculate_log_kernel.  For such a Model an iteration is cut where the reference calls the Model (GLMCMC.py:71-74,94-97):
//
//   propose_kernel   one work-item per candidate row r = j*C + c: the Philox blocks of candidate j of chain c -- the SAME
//                    slots, words and Box-Muller pairs as chain_step (glabc_device.h) -- give theta', forward()'s log q and
//                    the simulator's normals; the work-item of candidate 0 also draws the step head (branch, accept,
//                    resampling uniforms) and builds the local move's theta' = Theta_old + increment on the local branch
//   select_kernel    one work-item per chain: iSIR weights exp((prior' + K') - q'), normalisation in torch.sum's order
//                    (aten_rowsum_rt: any n, including ATen's cascade levels from 512 elements on), the reference's
//                    double running sum against the double uniform, or the MH tests; winner row copied into the
//                    chain-major state; Theta_Re row; streaming sums
//
// Everything between the two is the Model's code.  When that code is the build's own row-wise kernels the chains equal the
// fused kernels' bit for bit (tests/test_generic_path.py) -- same draws, same operation order, same exp / log.
// Both kernels are HBM-streaming: a candidate row is 4*(theta_dim + y_dim + noise_dim + 3) bytes written and read once.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "glabc_device.h"

namespace glabc {

constexpr int GD = GLABC_MAX_DIM;

struct GenDist {
    int32_t present;             // 0: the caller fills this proposal's rows itself
    int32_t kind, dim;
    float c0;
    float p0[GD], p1[GD], p2[GD], p3[GD];
};

// torch.sum over a short float64 register row t[0..n), n <= 8: aten_rowsum_f64<N> (glabc_device.h) spelled with predicates
GLABC_DEV double rowsum_small_f64(const double (&t)[GD], int n)
{
    if (n < 4) {
        double s = t[0];
        if (n > 1) s = s + t[1];
        if (n > 2) s = s + t[2];
        return s;
    }
    double acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = (n == 8) ? t[k] + t[4 + k] : t[k];
    double fa = 0.0;
    if (n < 8) {
#pragma unroll
        for (int i = 4; i < 8; ++i)
            if (i < n) fa = fa + t[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) fa = fa + acc[k];
    return fa;
}

// DiagGaussian.log_prob / Uniform.log_prob / Gamma.log_prob (distribution.py:176-181, 81-86, 123-137) at a point held in registers
GLABC_DEV float gen_log_prob(const GenDist& g, const float (&z)[GD])
{
    if (g.kind == GLABC_DIST_GAMMA) {                                  // float64 per coordinate, rounded once (include/glabc.h)
        double t[GD];
#pragma unroll
        for (int j = 0; j < GD; ++j)
            t[j] = j < g.dim ? glabc_gamma_log_pdf((double)g.p0[j], (double)g.p2[j], (double)g.p3[j], (double)z[j]) : 0.0;
        return (float)rowsum_small_f64(t, g.dim);
    }
    if (g.kind == GLABC_DIST_DIAG_GAUSS) {
        float t[GD];
#pragma unroll
        for (int j = 0; j < GD; ++j) {
            const float e = (z[j] - g.p0[j]) / g.p2[j];
            t[j] = j < g.dim ? g.p1[j] + 0.5f * (e * e) : 0.0f;
        }
        return g.c0 - rowsum_small(t, g.dim);
    }
    bool out = false;
#pragma unroll
    for (int j = 0; j < GD; ++j) out = out || (j < g.dim && ((z[j] < g.p0[j]) || (z[j] > g.p1[j])));
    return out ? -__builtin_inff() : g.c0;
}

struct GenArgs {
    int32_t algo, n_prop, theta_dim, y_dim, noise_dim;
    GenDist local, global;
    // chains (chain-major)
    float* theta;
    float* y;
    float* log_w;
    uint32_t* flags;
    uint32_t* n_moves;
    int64_t n_chains, chain0, stride;
    // run
    uint32_t seed_lo, seed_hi, step;
    const uint32_t* step_dev;     // glabc_run.step0_device: the iteration index lives on the device (graph replay)
    float gf;
    const float* gf_chain;
    float* history;
    int64_t hist_stride;
    double* sum_theta;
    double* sum_outer;
    double* sum_jump;
    // step io
    float* theta_prop;
    float* log_q;
    float* sim_noise;
    float* log_u;
    double* u_res;
    int32_t* is_global;
    const float* y_prop;
    const float* prior_prop;
    const float* kern_prop;
    float* prior_cur;
    float* kern_cur;
    const float* q_cur;
    const int32_t* n_valid;
    int32_t redraw_round;
    int32_t* n_redrawn;
};

// theta' and forward()'s log_p of one proposal from its noise e (distribution.py:166-174 / 73-79); local: + Theta_old
GLABC_DEV void apply_proposal(const GenDist& g, const float (&e)[GD], const float* th_old, int64_t stride, float* th_out, float* lq_out)
{
    float t[GD];
#pragma unroll
    for (int q = 0; q < GD; ++q) {
        if (q < g.dim) {
            const float v = g.p0[q] + g.p2[q] * e[q];                          // distribution.py:170 / :77
            th_out[q] = th_old ? v + th_old[q * stride] : v;                   // GLMCMC.py:91
        }
        t[q] = q < g.dim ? g.p1[q] + 0.5f * (e[q] * e[q]) : 0.0f;
    }
    if (lq_out) *lq_out = g.kind == GLABC_DIST_DIAG_GAUSS ? g.c0 - rowsum_small(t, g.dim) : g.c0;
}

__global__ void __launch_bounds__(256) propose_kernel(const GenArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_chains * a.n_prop) return;
    const int j = (int)(r / a.n_chains);
    const int64_t c = r - (int64_t)j * a.n_chains;
    const uint64_t gid = (uint64_t)(a.chain0 + c);
    const uint32_t c0 = (uint32_t)gid, c1 = (uint32_t)(gid >> 32);
    const int D = a.theta_dim, DP = D + (D & 1), ND = a.noise_dim;
    const int SPP = (DP + ND + 3) / 4;
    const uint32_t step = a.step_dev ? *a.step_dev : a.step;

    bool is_global = true;
    if (j == 0) {                                                           // the step head, Philox slot 0
        const glabc_u32x4 h = glabc_philox4x32_10(c0, c1, step, 0u, a.seed_lo, a.seed_hi);
        const float ub = glabc_uniform_f32(h.v[0]), ua = glabc_uniform_f32(h.v[1]);
        const float gf = a.gf_chain ? a.gf_chain[c] : a.gf;
        is_global = ub < gf;                                                // GLMCMC.py:59 / GlobalMCMC.py:39
        a.is_global[c] = is_global ? 1 : 0;
        a.log_u[c] = (ua == 0.0f) ? -__builtin_inff() : glabc_logf_normal(ua);   // GLMCMC.py:98
        a.u_res[c] = glabc_uniform_f64(h.v[2], h.v[3]);                     // GLMCMC.py:17
    }
    const GenDist& g = is_global ? a.global : a.local;
    const bool uni = g.kind == GLABC_DIST_UNIFORM;

    float e[GD];
#pragma unroll
    for (int q = 0; q < GD; ++q) e[q] = 0.0f;
    float* noise_row = a.sim_noise ? a.sim_noise + r * ND : nullptr;
    // blocks 0 and 1 hold every proposal word (theta_dim <= 8): statically indexed
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < SPP) {
            const glabc_u32x4 o = glabc_philox4x32_10(c0, c1, step, (uint32_t)(1 + j * SPP + b), a.seed_lo, a.seed_hi);
            float nrm[4];
            glabc_normal_pair(o.v[0], o.v[1], &nrm[0], &nrm[1]);
            glabc_normal_pair(o.v[2], o.v[3], &nrm[2], &nrm[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gi = 4 * b + t;
                if (gi < D) e[gi] = uni ? glabc_uniform_f32(o.v[t]) : nrm[t];
                if (noise_row && gi >= DP && gi - DP < ND) noise_row[gi - DP] = nrm[t];
            }
        }
    }
    for (int b = 2; b < SPP; ++b) {                                         // simulator normals only
        const glabc_u32x4 o = glabc_philox4x32_10(c0, c1, step, (uint32_t)(1 + j * SPP + b), a.seed_lo, a.seed_hi);
        float nrm[4];
        glabc_normal_pair(o.v[0], o.v[1], &nrm[0], &nrm[1]);
        glabc_normal_pair(o.v[2], o.v[3], &nrm[2], &nrm[3]);
        if (noise_row) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gi = 4 * b + t;
                if (gi >= DP && gi - DP < ND) noise_row[gi - DP] = nrm[t];      // theta_dim > 8: words 8 .. DP-1 are proposal words
            }
        }
    }
    if (!g.present) return;                                                 // the caller fills this proposal's rows
    float* th_out = a.theta_prop + r * D;
    if (is_global && g.kind == GLABC_DIST_GAMMA) {                          // Gamma.forward on the chain's Gamma slots (include/glabc.h)
        double t[GD];
#pragma unroll
        for (int q = 0; q < GD; ++q) t[q] = 0.0;
        for (int q = 0; q < g.dim; ++q) {
            const double z = glabc_gamma_draw_candidate((double)g.p0[q], c0, c1, step, j, q, a.seed_lo, a.seed_hi) * (double)g.p2[q];
            th_out[q] = (float)z;
            const double lp = glabc_gamma_log_pdf((double)g.p0[q], (double)g.p2[q], (double)g.p3[q], z);
#pragma unroll
            for (int k = 0; k < GD; ++k)
                if (k == q) t[k] = lp;
        }
        a.log_q[r] = (float)rowsum_small_f64(t, g.dim);
        return;
    }
    if (is_global) {
        apply_proposal(g, e, nullptr, 0, th_out, a.log_q + r);              // GLMCMC.py:66 / GlobalMCMC.py:40
    } else {
        apply_proposal(g, e, a.theta + c, a.stride, th_out, nullptr);       // GLMCMC.py:91 / GlobalMCMC.py:56
        a.log_q[r] = 0.0f;                                                  // unused by the local move
    }
}

// GLMCMC.py:92-93: a chain on the local branch whose proposal's prior is the sentinel 7*log(1e-10) redraws its increment
__global__ void __launch_bounds__(256) redraw_kernel(const GenArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n_chains) return;
    if (a.is_global[c] & 1) return;
    // the reference compares a float32 tensor with the Python float 7*log(1e-10): torch promotes the double to float32
    if (a.prior_prop[c] != (float)(7.0 * -23.025850929940457)) return;
    const uint64_t gid = (uint64_t)(a.chain0 + c);
    const uint32_t c0 = (uint32_t)gid, c1 = (uint32_t)(gid >> 32);
    const GenDist& g = a.local;
    const bool uni = g.kind == GLABC_DIST_UNIFORM;
    float e[GD];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const glabc_u32x4 o = glabc_philox4x32_10(c0, c1, a.step_dev ? *a.step_dev : a.step,
                                                  GLABC_SLOT_REDRAW + (uint32_t)(2 * a.redraw_round + b), a.seed_lo, a.seed_hi);
        float nrm[4];
        glabc_normal_pair(o.v[0], o.v[1], &nrm[0], &nrm[1]);
        glabc_normal_pair(o.v[2], o.v[3], &nrm[2], &nrm[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) e[4 * b + t] = uni ? glabc_uniform_f32(o.v[t]) : nrm[t];
    }
    apply_proposal(g, e, a.theta + c, a.stride, a.theta_prop + c * a.theta_dim, nullptr);
    atomicAdd(a.n_redrawn, 1);
}

__global__ void __launch_bounds__(256) select_kernel(const GenArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n_chains) return;
    const int D = a.theta_dim, YD = a.y_dim;
    const int64_t C = a.n_chains;
    const bool is_global = (a.is_global[c] & 1) != 0;
    // GLMCMC.py:67-70: proposals with a NaN coordinate were removed before the Model saw them (glabc_step_io.n_valid)
    int N = a.n_prop;
    if (a.n_valid && is_global) N = a.n_valid[c] < 0 ? 0 : (a.n_valid[c] < N ? a.n_valid[c] : N);
    const float prior_c = a.prior_cur[c], kern_c = a.kern_cur[c];
    uint32_t flags = a.flags ? a.flags[c] : 0u;
    float log_w = a.log_w ? a.log_w[c] : 0.0f;

    auto q_of_state = [&]() -> float {                                      // Importance_Proposal.log_prob(Theta_old)
        if (a.q_cur) return a.q_cur[c];
        float z[GD];
#pragma unroll
        for (int j = 0; j < GD; ++j) z[j] = j < D ? a.theta[j * a.stride + c] : 0.0f;
        return gen_log_prob(a.global, z);
    };

    int ind = 0;
    const bool isir = a.algo == GLABC_ALGO_GLMCMC || a.algo == GLABC_ALGO_GLMALA;
    if (isir && is_global) {
        if (flags & GLABC_FLAG_LOCAL) log_w = (prior_c + kern_c) - q_of_state();          // GLMCMC.py:60-64
        flags &= ~GLABC_FLAG_LOCAL;                                                       // GLMCMC.py:65
        auto weight = [&](int k) -> float {                                               // GLMCMC.py:74-81
            const float lw = k == 0 ? log_w
                                    : (a.prior_prop[(int64_t)(k - 1) * C + c] + a.kern_prop[(int64_t)(k - 1) * C + c]) -
                                          a.log_q[(int64_t)(k - 1) * C + c];
            const float v = glabc_expf(lw);
            return (v != v) ? 0.0f : v;
        };
        const float tot = aten_rowsum_rt(weight, N + 1);                                  // GLMCMC.py:82
        const double u = a.u_res[c];
        double run = 0.0;
        ind = -1;
        for (int k = 0; k <= N; ++k) {                                                    // GLMCMC.py:17-22
            run += (double)(weight(k) / tot);
            if (u < run) {
                ind = k;
                break;
            }
        }
        if (ind < 0) ind = 0;                                                             // None -> stay, GLMCMC.py:84
    } else {
        const float pk = a.prior_prop[c] + a.kern_prop[c];
        float log_acc;
        if (a.algo == GLABC_ALGO_GLOBALMCMC && is_global)
            log_acc = (((pk + q_of_state()) - a.log_q[c]) - prior_c) - kern_c;            // GlobalMCMC.py:44-46
        else if (a.algo == GLABC_ALGO_GLMALA)
            log_acc = ((pk + a.log_q[c]) - prior_c) - kern_c;                             // GLMALA.py:190-193
        else
            log_acc = (pk - prior_c) - kern_c;                                            // GLMCMC.py:96-97, GlobalMCMC.py:60-61
        ind = a.log_u[c] < log_acc ? 1 : 0;                                               // GLMCMC.py:98-99
    }

    const bool moved = ind > 0;
    const int64_t r = moved ? (int64_t)(ind - 1) * C + c : 0;
    const float* th_new = a.theta_prop + r * D;
    // Theta_Re row and streaming sums (the same accumulation as sampler_kernel)
    if (a.sum_theta) {
        int k = 0;
        for (int p = 0; p < D; ++p) {
            const float op = a.theta[p * a.stride + c], np_ = moved ? th_new[p] : op;
            a.sum_theta[p * a.stride + c] += (double)np_;
            for (int q = p; q < D; ++q, ++k) {
                const float oq = a.theta[q * a.stride + c], nq = moved ? th_new[q] : oq;
                a.sum_outer[k * a.stride + c] += (double)np_ * (double)nq;
                a.sum_jump[k * a.stride + c] += ((double)np_ - (double)op) * ((double)nq - (double)oq);
            }
        }
    }
    if (moved) {
        for (int j = 0; j < D; ++j) a.theta[j * a.stride + c] = th_new[j];                // GLMCMC.py:85,102
        const float* y_new = a.y_prop + r * YD;
        for (int j = 0; j < YD; ++j) a.y[j * a.stride + c] = y_new[j];                    // GLMCMC.py:87,103
        a.prior_cur[c] = a.prior_prop[r];
        a.kern_cur[c] = a.kern_prop[r];
        if (isir) {
            if (is_global)
                log_w = (a.prior_prop[r] + a.kern_prop[r]) - a.log_q[r];                  // GLMCMC.py:86
            else if (a.algo == GLABC_ALGO_GLMCMC)
                flags |= GLABC_FLAG_LOCAL;                                                // GLMCMC.py:100; not in GLMALA.py:195-199
        }
        if (a.n_moves) a.n_moves[c] += 1u;
        a.is_global[c] |= 2;
    }
    if (a.history) {                                                                      // GLMCMC.py:89,104
        // row (index - step0) when the iteration index is read from the device (one captured graph, many replays)
        float* row = a.history + (a.step_dev ? (int64_t)(*a.step_dev - a.step) * D * a.hist_stride : 0);
        for (int j = 0; j < D; ++j) row[j * a.hist_stride + c] = a.theta[j * a.stride + c];
    }
    if (isir) {
        a.log_w[c] = log_w;
        a.flags[c] = flags;
    }
}

// ---- generate_samples of a descriptor Model on row-major points (Mixture.py:13-26, examples/GK.py) ------------------------
struct SimArgs {
    int32_t sim_kind, theta_dim, y_dim;
    float gk_c;
    float noise_loc[GD], noise_scale[GD];
    const float* theta;
    const float* eps;
    float* y;
    int64_t n, row0;
    uint32_t seed_lo, seed_hi;
};

__global__ void __launch_bounds__(256) simulate_kernel(const SimArgs s)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= s.n) return;
    const int D = s.theta_dim, YD = s.y_dim;
    float eps[GD];
    if (s.eps) {
#pragma unroll
        for (int j = 0; j < GD; ++j) eps[j] = j < YD ? s.eps[r * YD + j] : 0.0f;
    } else {
        const uint64_t gid = (uint64_t)(s.row0 + r);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const glabc_u32x4 o = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, s.seed_lo, s.seed_hi);
            glabc_normal_pair(o.v[0], o.v[1], &eps[4 * b], &eps[4 * b + 1]);
            glabc_normal_pair(o.v[2], o.v[3], &eps[4 * b + 2], &eps[4 * b + 3]);
        }
    }
    if (s.sim_kind == GLABC_SIM_GK) {
        StepArgs<4, 8> a;
        a.sim_kind = GLABC_SIM_GK;
        a.gk_c = s.gk_c;
        float th[4], y[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) th[j] = s.theta[r * 4 + j];
        model_simulate<4, 8>(a, th, eps, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) s.y[r * 8 + j] = y[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < GD; ++j) {
        if (j < D) {
            const float noise = s.noise_loc[j] + s.noise_scale[j] * eps[j];               // Mixture.py:19-23
            s.y[r * YD + j] = __builtin_fabsf(s.theta[r * D + j]) + noise;
        }
    }
}

// test hook: aten_rowsum_rt of each row of x[n_rows][n]
__global__ void __launch_bounds__(64) rowsum_kernel(const float* __restrict__ x, int n_rows, int n, float* __restrict__ out)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rows) return;
    const float* row = x + (int64_t)r * n;
    out[r] = aten_rowsum_rt([&](int i) { return row[i]; }, n);
}

}  // namespace glabc

// =================================================================================================
using namespace glabc;

static int finish() { return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH; }

// allow_gamma: the global / importance proposal may be a Gamma (include/glabc.h); the local increment may not
static int pack_gen_dist(const glabc_dist* g, int dim, GenDist* o, bool allow_gamma)
{
    std::memset(o, 0, sizeof *o);
    if (!g) return GLABC_OK;
    if (g->dim != dim || dim < 1 || dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    const bool gamma = g->kind == GLABC_DIST_GAMMA;
    if (g->kind != GLABC_DIST_DIAG_GAUSS && g->kind != GLABC_DIST_UNIFORM && !(gamma && allow_gamma)) return GLABC_ERR_KIND;
    o->present = 1;
    o->kind = g->kind;
    o->dim = g->dim;
    o->c0 = gamma ? 0.0f : g->c0;
    if (!gamma && !std::isfinite(g->c0)) return GLABC_ERR_ARG;
    for (int j = 0; j < GLABC_MAX_DIM; ++j) {
        const bool in = j < dim;
        o->p0[j] = in ? g->p0[j] : 0.0f;
        o->p1[j] = in ? g->p1[j] : 0.0f;
        o->p2[j] = in ? g->p2[j] : 1.0f;
        o->p3[j] = (in && gamma) ? g->p3[j] : 0.0f;
        if (in && (!std::isfinite(g->p0[j]) || !std::isfinite(g->p1[j]) || !std::isfinite(g->p2[j]))) return GLABC_ERR_ARG;
        if (in && g->kind == GLABC_DIST_DIAG_GAUSS && !(g->p2[j] > 0.0f)) return GLABC_ERR_ARG;
        if (in && gamma && (!(g->p0[j] > 0.0f) || !(g->p1[j] > 0.0f) || !(g->p2[j] > 0.0f) || !std::isfinite(g->p3[j])))
            return GLABC_ERR_ARG;                                       // shape, rate, scale > 0; gammaln(shape) finite
    }
    return GLABC_OK;
}

static int pack_common(int algo, const glabc_dist* local, const glabc_dist* global, const glabc_chains* c, const glabc_run* r,
                       const glabc_step_io* io, GenArgs* a)
{
    if (!c || !r || !io) return GLABC_ERR_NULL;
    if (algo != GLABC_ALGO_GLMCMC && algo != GLABC_ALGO_GLOBALMCMC && algo != GLABC_ALGO_GLMALA) return GLABC_ERR_KIND;
    if (io->theta_dim < 1 || io->y_dim < 1 || io->noise_dim < 0) return GLABC_ERR_DIM;
    if (io->n_prop < 1 || (algo == GLABC_ALGO_GLOBALMCMC && io->n_prop != 1)) return GLABC_ERR_ARG;
    if ((int64_t)io->n_prop * c->n_chains > (int64_t)1 << 40) return GLABC_ERR_ARG;
    if (r->n_steps != 1 || r->tape || r->math_mode != GLABC_MATH_EXACT || r->dump_draws) return GLABC_ERR_ARG;
    if (c->n_chains < 0 || c->stride < c->n_chains || c->chain0 < 0) return GLABC_ERR_ARG;
    if (!c->theta || !c->y) return GLABC_ERR_NULL;
    if (!(r->global_frequency >= 0.0f) && !(r->global_frequency < 0.0f)) return GLABC_ERR_ARG;
    std::memset(a, 0, sizeof *a);
    int rc = pack_gen_dist(local, io->theta_dim, &a->local, false);
    if (rc) return rc;
    rc = pack_gen_dist(global, io->theta_dim, &a->global, true);
    if (rc) return rc;
    a->algo = algo;
    a->n_prop = io->n_prop;
    a->theta_dim = io->theta_dim;
    a->y_dim = io->y_dim;
    a->noise_dim = io->noise_dim;
    a->theta = c->theta;
    a->y = c->y;
    a->log_w = c->log_w;
    a->flags = c->flags;
    a->n_moves = c->n_moves;
    a->n_chains = c->n_chains;
    a->chain0 = c->chain0;
    a->stride = c->stride;
    a->seed_lo = (uint32_t)r->seed;
    a->seed_hi = (uint32_t)(r->seed >> 32);
    a->step = r->step0;
    a->step_dev = r->step0_device;
    a->gf = r->global_frequency;
    a->gf_chain = r->global_frequency_per_chain;
    a->history = r->history;
    a->hist_stride = r->hist_stride;
    if (r->history && r->hist_stride < c->n_chains) return GLABC_ERR_ARG;
    if (r->moments) {
        if (!r->moments->sum_theta || !r->moments->sum_outer || !r->moments->sum_jump) return GLABC_ERR_NULL;
        a->sum_theta = r->moments->sum_theta;
        a->sum_outer = r->moments->sum_outer;
        a->sum_jump = r->moments->sum_jump;
    }
    a->theta_prop = io->theta_prop;
    a->log_q = io->log_q;
    a->sim_noise = io->sim_noise;
    a->log_u = io->log_u;
    a->u_res = io->u_res;
    a->is_global = io->is_global;
    a->y_prop = io->y_prop;
    a->prior_prop = io->prior_prop;
    a->kern_prop = io->kern_prop;
    a->prior_cur = io->prior_cur;
    a->kern_cur = io->kern_cur;
    a->q_cur = io->q_cur;
    a->n_valid = io->n_valid;
    return GLABC_OK;
}

static unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" {

__attribute__((visibility("default"))) int glabc_propose(int algo, const glabc_dist* local, const glabc_dist* global,
                                                         const glabc_chains* chains, const glabc_run* run,
                                                         const glabc_step_io* io, void* stream)
{
    GenArgs a;
    int rc = pack_common(algo, local, global, chains, run, io, &a);
    if (rc) return rc;
    if (!io->theta_prop || !io->log_q || !io->log_u || !io->u_res || !io->is_global) return GLABC_ERR_NULL;
    if ((local || global) && io->theta_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (io->noise_dim > 0 && !io->sim_noise) return GLABC_ERR_NULL;
    if (chains->n_chains == 0) return GLABC_OK;
    hipLaunchKernelGGL(propose_kernel, dim3(blocks_for(chains->n_chains * io->n_prop)), dim3(256), 0, (hipStream_t)stream, a);
    return finish();
}

__attribute__((visibility("default"))) int glabc_propose_redraw(const glabc_dist* local, const glabc_chains* chains,
                                                                const glabc_run* run, const glabc_step_io* io, int32_t round,
                                                                int32_t* n_redrawn, void* stream)
{
    GenArgs a;
    if (!local || !n_redrawn) return GLABC_ERR_NULL;
    int rc = pack_common(GLABC_ALGO_GLMCMC, local, nullptr, chains, run, io, &a);
    if (rc) return rc;
    if (!io->theta_prop || !io->is_global || !io->prior_prop) return GLABC_ERR_NULL;
    if (round < 1 || round > (1 << 24)) return GLABC_ERR_ARG;
    if (chains->n_chains == 0) return GLABC_OK;
    a.redraw_round = round;
    a.n_redrawn = n_redrawn;
    hipLaunchKernelGGL(redraw_kernel, dim3(blocks_for(chains->n_chains)), dim3(256), 0, (hipStream_t)stream, a);
    return finish();
}

__attribute__((visibility("default"))) int glabc_select(int algo, const glabc_dist* global, const glabc_chains* chains,
                                                        const glabc_run* run, const glabc_step_io* io, void* stream)
{
    GenArgs a;
    int rc = pack_common(algo, nullptr, global, chains, run, io, &a);
    if (rc) return rc;
    if (!io->theta_prop || !io->log_q || !io->log_u || !io->u_res || !io->is_global || !io->y_prop || !io->prior_prop ||
        !io->kern_prop || !io->prior_cur || !io->kern_cur)
        return GLABC_ERR_NULL;
    if (algo != GLABC_ALGO_GLOBALMCMC && (!chains->log_w || !chains->flags)) return GLABC_ERR_NULL;
    if (!global && !io->q_cur) return GLABC_ERR_NULL;                     // someone has to supply q(Theta_old)
    if (chains->n_chains == 0) return GLABC_OK;
    hipLaunchKernelGGL(select_kernel, dim3(blocks_for(chains->n_chains)), dim3(256), 0, (hipStream_t)stream, a);
    return finish();
}

__attribute__((visibility("default"))) int glabc_selftest_rowsum(const float* x, int32_t n_rows, int32_t n, float* out, void* stream)
{
    if (!x || !out) return GLABC_ERR_NULL;
    if (n_rows < 0 || n < 1) return GLABC_ERR_ARG;
    if (n_rows == 0) return GLABC_OK;
    hipLaunchKernelGGL(rowsum_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, n_rows, n, out);
    return finish();
}

__attribute__((visibility("default"))) int glabc_model_simulate(const glabc_model* m, const float* theta, const float* eps,
                                                                int64_t n, uint64_t seed, int64_t row0, float* y, void* stream)
{
    if (!m || !theta || !y) return GLABC_ERR_NULL;
    if (m->sim_kind != GLABC_SIM_ABS_GAUSS && m->sim_kind != GLABC_SIM_GK) return GLABC_ERR_KIND;
    if (m->theta_dim < 1 || m->theta_dim > GLABC_MAX_DIM || m->y_dim < 1 || m->y_dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (m->sim_kind == GLABC_SIM_GK && (m->theta_dim != 4 || m->y_dim != 8)) return GLABC_ERR_DIM;
    if (m->sim_kind == GLABC_SIM_ABS_GAUSS && (m->y_dim != m->theta_dim || m->noise.dim != m->y_dim)) return GLABC_ERR_DIM;
    if (n < 0 || row0 < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    SimArgs s;
    std::memset(&s, 0, sizeof s);
    s.sim_kind = m->sim_kind;
    s.theta_dim = m->theta_dim;
    s.y_dim = m->y_dim;
    s.gk_c = m->gk_c;
    for (int j = 0; j < m->y_dim && m->sim_kind == GLABC_SIM_ABS_GAUSS; ++j) {
        s.noise_loc[j] = m->noise.p0[j];
        s.noise_scale[j] = m->noise.p2[j];
    }
    s.theta = theta;
    s.eps = eps;
    s.y = y;
    s.n = n;
    s.row0 = row0;
    s.seed_lo = (uint32_t)seed;
    s.seed_hi = (uint32_t)(seed >> 32);
    hipLaunchKernelGGL(simulate_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, s);
    return finish();
}

}  // extern "C"
