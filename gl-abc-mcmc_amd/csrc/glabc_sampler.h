// glabc_sampler.h -- the fused sampler kernel and its per-dimension launcher.
//
// sampler_kernel<ALGO, D, YD, N, L, VAR>: K iterations of GLMCMC / GlobalMCMC for every chain of
// the shard in ONE launch.  Chain state is loaded once, lives in VGPRs for the K
// iterations, and is stored once; per iteration only the Theta_Re history row (chain-major,
// coalesced) leaves the CU.  L lanes cooperate on a chain (glabc_device.h).
//
// Each theta_dim is compiled in its own translation unit (glabc_sampler_dim.hip with
// -DGLABC_DIM=d) so the 4 x 16 x 3 instantiations build in parallel.
#pragma once

#include "glabc_device.h"

namespace glabc {

constexpr int BLOCK = 64;      // one wavefront per workgroup: 65 536 chains x L lanes spread evenly over 1024 SIMDs

// SCHED only names the object the instantiation lives in: the same source is compiled twice, with the default
// (occupancy-oriented) instruction schedule and with -amdgpu-sched-strategy=max-ilp (see the Makefile and run_sampler)
// (The body stays INSIDE the __global__ function: as a device function called with the argument block the same code compiled
// to a 2 % slower kernel -- 224 instead of 219 VGPRs, 40 instead of 43 scalar loads -- measured back to back on one box.  The
// run-time compiled form of glabc_rtc.hip therefore instantiates this very template.)
template <int ALGO, int D, int YD, int N, int L, int VAR, int SCHED>
__global__ void __launch_bounds__(BLOCK) sampler_kernel(const StepArgs<D, YD> a)
{
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t chain = tid / L;
    const int sub = (int)(tid % L);
    const bool valid = chain < a.n_chains;
    const int64_t i = valid ? chain : a.n_chains - 1;      // tail lanes shadow the last chain (no stores): the
                                                           // group exchanges need every lane of the wave alive
    const bool writer = valid && sub == 0;

    Chain<D, YD> c;
#pragma unroll
    for (int j = 0; j < D; ++j) c.theta[j] = a.theta[j * a.stride + i];
#pragma unroll
    for (int j = 0; j < YD; ++j) c.y[j] = a.y[j * a.stride + i];
    c.log_w = (ALGO == ALGO_GLMCMC) ? a.log_w[i] : 0.0f;
    c.flags = (ALGO == ALGO_GLMCMC) ? a.flags[i] : 0u;
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;
    c.gf = a.gf_chain ? a.gf_chain[i] : a.gf;
    refresh_cache<D, YD, VAR == VAR_GAMMA>(a, c);
    c.lw_cur = (c.flags & GLABC_FLAG_LOCAL) ? (c.prior + c.kern) - c.q : c.log_w;          // GLMCMC.py:60-64
    {
        const float v = glabc_expf(c.lw_cur);
        c.w_cur = (v != v) ? 0.0f : v;                                                      // GLMCMC.py:78-81
    }

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

        const bool moved = chain_step<ALGO, D, YD, N, L, VAR>(a, rng, step, sub, c, i * (int64_t)a.n_steps + t);
        c.n_moves += moved ? 1u : 0u;

        if (hist && writer) {                                       // Theta_Re[i,:] = Theta_old, GLMCMC.py:89,104
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

    if (writer) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.theta[j * a.stride + i] = c.theta[j];
#pragma unroll
        for (int j = 0; j < YD; ++j) a.y[j * a.stride + i] = c.y[j];
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
}

#if !defined(__HIPCC_RTC__)       // host-side declarations: not for the run-time compiled form (glabc_rtc.hip)
// host-side launcher of one theta_dim; defined in glabc_sampler_dim.hip (one TU per D and schedule).
// lanes = lanes per chain actually compiled for (1, 2 or 4).  Returns a glabc_status.
// SCHED_ILP objects hold the one-lane-per-chain kernels only (the schedule for launches of at most two waves per SIMD).
constexpr int SCHED_DEFAULT = 0, SCHED_ILP = 1;
template <int D, int YD, int SCHED>
int launch_sampler_dim(int algo, int n_batch, int lanes, const StepArgs<D, YD>& a, hipStream_t stream);

// GLMCMC with batch_size > GLABC_MAX_BATCH: lane groups of one wavefront share a chain's proposals (glabc_wide.hip).
// lanes = 0 (choose) or 8 / 16 / 32 / 64.
template <int D, int YD>
int launch_wide(const StepArgs<D, YD>& a, int n_batch, int lanes, hipStream_t stream);
#endif

}  // namespace glabc
