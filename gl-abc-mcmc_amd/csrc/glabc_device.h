// glabc_device.h -- per-chain device code of the fused GL-ABC-MCMC step (gfx950).
//
// One work-item owns one chain for the whole launch: state (theta, y, cached iSIR
// log-weight, flags, streaming moments) is loaded once into VGPRs, K iterations run
// back to back in registers, and only the Theta_Re history row (coalesced,
// chain-major) leaves the CU per iteration.  Every quantity that is the same for all
// chains (model / proposal parameters, seed, global_frequency) arrives in the kernel
// argument block, i.e. in SGPRs.
//
// The arithmetic follows the reference line by line (citations = /root/reference
// paths) in float32 with -ffp-contract=off, using the elementary functions and the
// Philox stream of include/glabc_numerics.h, so that the CPU checker in oracle/
// (an independent plain-C restatement) can be compared bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glabc.h"
#include "../../include/glabc_numerics.h"

namespace glabc {

#define GLABC_DEV static __device__ __forceinline__

// ---- argument block ------------------------------------------------------------
template <int D>
struct DistArgs {
    int32_t kind;
    float c0;
    float p0[D], p1[D], p2[D];
};

template <int D>
struct StepArgs {
    // Model callbacks (examples/Mixture.py:13-45); y_dim == theta_dim for |theta| + noise
    DistArgs<D> prior;
    float noise_loc[D], noise_scale[D];
    float y_obs[D];
    float kern_log_scale, kern_scale, kern_c0;
    // proposals: local increment (GLMCMC.py:91) and global / importance (GLMCMC.py:66, GlobalMCMC.py:40)
    DistArgs<D> local, global;
    // chains
    float* theta;
    float* y;
    float* log_w;
    uint32_t* flags;
    uint32_t* n_moves;
    int64_t n_chains, chain0, stride;
    // run
    uint32_t seed_lo, seed_hi, step0;
    int32_t n_steps;
    float gf;
    float* history;
    int64_t hist_stride;
    double* sum_theta;
    double* sum_outer;
    double* sum_jump;
};

// ---- torch.sum association over a contiguous float32 row (GLMCMC.py:82) ---------
// Probed on the reference's torch build (DESIGN.md "row-sum order"): n < 8 -> four
// scalar lanes, leftovers into lane 0, lanes combined left to right; n >= 8 -> the
// same scheme over 8-wide vectors, then a scalar accumulator takes the n%8 tail in
// order followed by the 8 vector partials in order.  N is a compile-time constant,
// so all of this unrolls into a fixed add tree.
template <int N>
GLABC_DEV float aten_rowsum(const float (&x)[N])
{
    if constexpr (N < 4) {
        float s = x[0];
#pragma unroll
        for (int i = 1; i < N; ++i) s = s + x[i];
        return s;
    } else if constexpr (N < 8) {
        float l0 = x[0];
#pragma unroll
        for (int i = 4; i < N; ++i) l0 = l0 + x[i];
        return ((l0 + x[1]) + x[2]) + x[3];
    } else {
        constexpr int NV = N / 8;
        constexpr int G = NV / 4;
        float acc[8];
        if constexpr (G == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = x[k];
#pragma unroll
                for (int v = 1; v < NV; ++v) acc[k] = acc[k] + x[8 * v + k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) l[q] = x[8 * q + k];
#pragma unroll
                for (int i = 1; i < G; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) l[q] = l[q] + x[8 * (4 * i + q) + k];
#pragma unroll
                for (int v = 4 * G; v < NV; ++v) l[0] = l[0] + x[8 * v + k];
                acc[k] = ((l[0] + l[1]) + l[2]) + l[3];
            }
        }
        float fa = 0.0f;
#pragma unroll
        for (int i = 8 * NV; i < N; ++i) fa = fa + x[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) fa = fa + acc[k];
        return fa;
    }
}

// ---- distribution.py ---------------------------------------------------------------
// DiagGaussian.log_prob, distribution.py:176-181 / Uniform.log_prob, distribution.py:81-86
template <int D>
GLABC_DEV float dist_log_prob(const DistArgs<D>& g, const float (&z)[D])
{
    if (g.kind == GLABC_DIST_DIAG_GAUSS) {
        float t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float e = (z[j] - g.p0[j]) / g.p2[j];
            t[j] = g.p1[j] + 0.5f * (e * e);
        }
        return g.c0 - aten_rowsum<D>(t);
    } else {
        bool out = false;
#pragma unroll
        for (int j = 0; j < D; ++j) out = out || (z[j] < g.p0[j]) || (z[j] > g.p1[j]);
        return out ? -__builtin_inff() : g.c0;
    }
}

// forward() given its noise: DiagGaussian distribution.py:166-174 (noise = N(0,1) draws),
// Uniform distribution.py:73-79 (noise = [0,1) draws)
template <int D>
GLABC_DEV float dist_forward(const DistArgs<D>& g, const float (&noise)[D], float (&z)[D])
{
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = g.p0[j] + g.p2[j] * noise[j];
    if (g.kind == GLABC_DIST_DIAG_GAUSS) {
        float t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) t[j] = g.p1[j] + 0.5f * (noise[j] * noise[j]);
        return g.c0 - aten_rowsum<D>(t);
    }
    return g.c0;
}

// ---- examples/Mixture.py ----------------------------------------------------------------
// generate_samples, Mixture.py:19-23: y = |theta| + (loc + scale*eps)
template <int D>
GLABC_DEV void model_simulate(const StepArgs<D>& a, const float (&theta)[D], const float (&eps)[D], float (&y)[D])
{
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float noise = a.noise_loc[j] + a.noise_scale[j] * eps[j];
        y[j] = __builtin_fabsf(theta[j]) + noise;
    }
}

// calculate_log_kernel, Mixture.py:33-45: DiagGaussian(1, 0, log eps).log_prob(||y - y_obs||)
template <int D>
GLABC_DEV float model_log_kernel(const StepArgs<D>& a, const float (&y)[D])
{
    float t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float d = y[j] - a.y_obs[j];
        t[j] = d * d;
    }
    float dis = __builtin_sqrtf(aten_rowsum<D>(t));
    float e = (dis - 0.0f) / a.kern_scale;
    return a.kern_c0 - (a.kern_log_scale + 0.5f * (e * e));
}

// ---- random draws of one (chain, step) ---------------------------------------------
struct StepHead {
    float u_branch, u_accept;
    double u_resample;
};

struct Rng {
    uint32_t c0, c1, k0, k1;
};

GLABC_DEV StepHead draw_head(const Rng& r, uint32_t step)
{
    glabc_u32x4 h = glabc_philox4x32_10(r.c0, r.c1, step, 0u, r.k0, r.k1);
    StepHead s;
    s.u_branch = glabc_uniform_f32(h.v[0]);
    s.u_accept = glabc_uniform_f32(h.v[1]);
    s.u_resample = glabc_uniform_f64(h.v[2], h.v[3]);
    return s;
}

// Proposal j of a step: D proposal draws then D simulator draws out of
// ceil(2D/4) Philox blocks at slots 1 + j*spp + b.  Normals come in Box-Muller
// pairs from words (2i, 2i+1); a Uniform proposal takes word i as a [0,1) uniform.
template <int D>
GLABC_DEV void draw_proposal(const Rng& r, uint32_t step, int j, bool uniform_prop, float (&e)[D], float (&s)[D])
{
    constexpr int M = 2 * D;
    constexpr int SPP = (M + 3) / 4;
    uint32_t w[4 * SPP];
#pragma unroll
    for (int b = 0; b < SPP; ++b) {
        glabc_u32x4 o = glabc_philox4x32_10(r.c0, r.c1, step, (uint32_t)(1 + j * SPP + b), r.k0, r.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[4 * b + q] = o.v[q];
    }
    float nrm[2 * ((M + 1) / 2)];
#pragma unroll
    for (int i = 0; 2 * i < M; ++i) glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        e[i] = uniform_prop ? glabc_uniform_f32(w[i]) : nrm[i];
        s[i] = nrm[D + i];
    }
}

// ---- chain state in registers ---------------------------------------------------------
template <int D>
struct Chain {
    float theta[D];
    float y[D];
    float log_w;
    uint32_t flags;
    uint32_t n_moves;
};

// random-walk MH local move, GLMCMC.py:90-104 == GlobalMCMC.py:55-68
//   theta' = Local_Proposal.sample(1) + theta ; log_acc = ((prior' + K') - prior) - K
template <int D>
GLABC_DEV bool local_move(const StepArgs<D>& a, const Rng& r, uint32_t step, float u_accept, Chain<D>& c)
{
    float e[D], s[D], inc[D], th[D], y[D];
    draw_proposal<D>(r, step, 0, a.local.kind == GLABC_DIST_UNIFORM, e, s);
    (void)dist_forward<D>(a.local, e, inc);
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = inc[j] + c.theta[j];
    model_simulate<D>(a, th, s, y);
    float log_acc = ((dist_log_prob<D>(a.prior, th) + model_log_kernel<D>(a, y)) - dist_log_prob<D>(a.prior, c.theta)) -
                    model_log_kernel<D>(a, c.y);
    bool acc = glabc_logf(u_accept) < log_acc;
    if (acc) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            c.theta[j] = th[j];
            c.y[j] = y[j];
        }
    }
    return acc;
}

// independence MH global move, GlobalMCMC.py:39-53
//   log_acc = ((((prior' + K') + q(theta)) - q') - prior) - K
template <int D>
GLABC_DEV bool independence_move(const StepArgs<D>& a, const Rng& r, uint32_t step, float u_accept, Chain<D>& c)
{
    float e[D], s[D], th[D], y[D];
    draw_proposal<D>(r, step, 0, a.global.kind == GLABC_DIST_UNIFORM, e, s);
    float lq_new = dist_forward<D>(a.global, e, th);
    model_simulate<D>(a, th, s, y);
    float lq_old = dist_log_prob<D>(a.global, c.theta);
    float log_acc = ((((dist_log_prob<D>(a.prior, th) + model_log_kernel<D>(a, y)) + lq_old) - lq_new) -
                     dist_log_prob<D>(a.prior, c.theta)) -
                    model_log_kernel<D>(a, c.y);
    bool acc = glabc_logf(u_accept) < log_acc;
    if (acc) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            c.theta[j] = th[j];
            c.y[j] = y[j];
        }
    }
    return acc;
}

// log_weight_old, GLMCMC.py:53-55 / 62-64
template <int D>
GLABC_DEV float isir_weight_of_state(const StepArgs<D>& a, const Chain<D>& c)
{
    return (dist_log_prob<D>(a.prior, c.theta) + model_log_kernel<D>(a, c.y)) - dist_log_prob<D>(a.global, c.theta);
}

// iSIR global move, GLMCMC.py:60-88.  The N proposals, their simulations and the N+1
// weights live in registers (N is a template parameter); the resampling index is the
// reference's double-precision running sum against a double uniform (GLMCMC.py:7-22).
// (The NaN-row filter of GLMCMC.py:67-70 cannot trigger: the entry point rejects
// non-finite proposal parameters and the draws are finite.)
template <int D, int N>
GLABC_DEV bool isir_move(const StepArgs<D>& a, const Rng& r, uint32_t step, double u_resample, Chain<D>& c)
{
    float th[N][D], y[N][D];
    float lw[N + 1], w[N + 1];
    if (c.flags & GLABC_FLAG_LOCAL) c.log_w = isir_weight_of_state<D>(a, c);   // :60-64
    c.flags &= ~GLABC_FLAG_LOCAL;                                              // :65
    lw[0] = c.log_w;
    const bool uni = a.global.kind == GLABC_DIST_UNIFORM;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float e[D], s[D];
        draw_proposal<D>(r, step, j, uni, e, s);
        float lq = dist_forward<D>(a.global, e, th[j]);                       // :66
        model_simulate<D>(a, th[j], s, y[j]);                                  // :71
        lw[j + 1] = (dist_log_prob<D>(a.prior, th[j]) + model_log_kernel<D>(a, y[j])) - lq;   // :72-74
    }
#pragma unroll
    for (int k = 0; k <= N; ++k) {
        float v = glabc_expf(lw[k]);                                           // :78
        w[k] = (v != v) ? 0.0f : v;                                            // :80-81
    }
    float tot = aten_rowsum<N + 1>(w);                                         // :82
    int ind = -1;
    double run = 0.0;
#pragma unroll
    for (int k = 0; k <= N; ++k) {
        run += (double)(w[k] / tot);
        ind = (ind < 0 && u_resample < run) ? k : ind;                         // weight_sampling :17-22
    }
    bool moved = ind > 0;                                                      // :84
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (ind == j + 1) {
#pragma unroll
            for (int q = 0; q < D; ++q) {
                c.theta[q] = th[j][q];
                c.y[q] = y[j][q];
            }
            c.log_w = lw[j + 1];
        }
    }
    return moved;
}

}  // namespace glabc
