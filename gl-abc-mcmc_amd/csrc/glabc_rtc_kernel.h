// glabc_rtc_kernel.h -- what hiprtc compiles around a USER simulator (glabc_rtc.hip).
//
// The translation unit hiprtc sees is
//     <fixed-width typedefs>
//     #define GLABC_RTC_ALGO / _D / _YD / _N / _L        the configuration (L = lanes per chain, chosen by the host)
//     #define GLABC_USER_SIM 1, GLABC_USER_NOISE_DIM    make model_simulate call the user's function
//     #include "glabc_numerics.h"                        so that the user's source can use glabc_expf, glabc_logf, ...
//     <the user's source: GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
//      and, each announced by a #define in that source (glabc_device.h, "Run-time compiled builds may replace ..."):
//      glabc_user_prior_log_prob / glabc_user_discrepancy / glabc_user_log_kernel>
//     #include "glabc_rtc_kernel.h"
// i.e. an explicit instantiation of sampler_kernel (glabc_sampler.h) -- the SAME template the library's built-in kernels are
// instantiated from -- with the user's function inlined where the built-in simulators are; the host fills the same
// StepArgs<D, YD> argument block as for the built-in kernels (glabc_pack.h).
#pragma once

#include "glabc_sampler.h"

namespace glabc {

template __global__ void sampler_kernel<GLABC_RTC_ALGO, GLABC_RTC_D, GLABC_RTC_YD, GLABC_RTC_N, GLABC_RTC_L, VAR_GENERIC, 0>(
    const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>);
#if GLABC_RTC_YD == GLABC_RTC_D && !defined(GLABC_USER_PRIOR) && !defined(GLABC_USER_DISCREPANCY) && !defined(GLABC_USER_KERNEL)
// ... and the branch-free variant for unit-scale Gaussian prior / global proposal (chosen per launch by the host, as for the
// built-in kernels: glabc_pack.h gauss_unit_config)
template __global__ void sampler_kernel<GLABC_RTC_ALGO, GLABC_RTC_D, GLABC_RTC_YD, GLABC_RTC_N, GLABC_RTC_L, VAR_GAUSS_UNIT, 0>(
    const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>);
#endif

// ... and the team geometry of glabc_team.h (two or three wavefronts per 64 chains: what the built-in GLMCMC kernels run at 16 384 ..
// 131 072 chains), where the host found the configuration to fit (GLABC_RTC_TEAM3 / GLABC_RTC_TEAM2)
#if GLABC_RTC_ALGO == 0 && (defined(GLABC_RTC_TEAM3) || defined(GLABC_RTC_TEAM2))
}  // namespace glabc
#include "glabc_team.h"
namespace glabc {
#define GLABC_RTC_TEAM_INST(NW_, VAR_)                                                                                         \
    template __global__ void team_sampler_kernel<GLABC_RTC_D, GLABC_RTC_YD, GLABC_RTC_N, VAR_, NW_, false>(                   \
        const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>, int);
#ifdef GLABC_RTC_TEAM3
GLABC_RTC_TEAM_INST(3, VAR_GENERIC)
#endif
#ifdef GLABC_RTC_TEAM2
GLABC_RTC_TEAM_INST(2, VAR_GENERIC)
#endif
#if GLABC_RTC_YD == GLABC_RTC_D && !defined(GLABC_USER_PRIOR) && !defined(GLABC_USER_DISCREPANCY) && !defined(GLABC_USER_KERNEL)
#ifdef GLABC_RTC_TEAM3
GLABC_RTC_TEAM_INST(3, VAR_GAUSS_UNIT)
#endif
#ifdef GLABC_RTC_TEAM2
GLABC_RTC_TEAM_INST(2, VAR_GAUSS_UNIT)
#endif
#endif
#endif

// GlobalMCMC: the team of two wavefronts (global_team_kernel: the helper draws the random numbers a chunk of iterations ahead)
#if GLABC_RTC_ALGO == 1 && defined(GLABC_RTC_GTEAM)
}  // namespace glabc
#include "glabc_team.h"
namespace glabc {
template __global__ void global_team_kernel<GLABC_RTC_D, GLABC_RTC_YD, VAR_GENERIC, 2>(const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>, int);
#if GLABC_RTC_YD == GLABC_RTC_D && !defined(GLABC_USER_PRIOR) && !defined(GLABC_USER_DISCREPANCY) && !defined(GLABC_USER_KERNEL)
template __global__ void global_team_kernel<GLABC_RTC_D, GLABC_RTC_YD, VAR_GAUSS_UNIT, 2>(const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>, int);
#endif
#endif

// generate_samples(theta, 1) on rows with the noise supplied (the Model protocol's callback, for y0 and the split-phase path):
// theta[n][D], eps[n][ND] -> y[n][YD]
extern "C" __global__ void __launch_bounds__(256) glabc_rtc_simulate_rows(const float* __restrict__ theta, const float* __restrict__ eps,
                                                                          float* __restrict__ y, const int64_t n)
{
    constexpr int D = GLABC_RTC_D, YD = GLABC_RTC_YD, ND = GLABC_USER_NOISE_DIM;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    float th[D], e[ND], yy[YD];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = theta[r * D + j];
#pragma unroll
    for (int j = 0; j < ND; ++j) e[j] = eps[r * ND + j];
    glabc_user_simulate(th, e, yy);
#pragma unroll
    for (int j = 0; j < YD; ++j) y[r * YD + j] = yy[j];
}

// The Model protocol's other callbacks on rows, through the same functions the fused kernel calls (the user's where announced,
// the descriptor's otherwise): prior_log_prob(theta[n][D]) -- only built when the prior is the user's (the library's own
// glabc_model_prior_log_prob serves descriptor priors) --, discrepancy(y[n][YD]) and calculate_log_kernel(y[n][YD]).
struct RtcRowArgs {
    const float* in;
    float* out;
    int64_t n;
    float y_obs[GLABC_RTC_YD];
    float kern_scale, kern_log_scale, kern_c0;
    int32_t what;                 // 0 prior_log_prob, 1 discrepancy, 2 calculate_log_kernel
};

extern "C" __global__ void __launch_bounds__(256) glabc_rtc_model_rows_kernel(const RtcRowArgs a)
{
    constexpr int D = GLABC_RTC_D, YD = GLABC_RTC_YD;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n) return;
    if (a.what == 0) {
#ifdef GLABC_USER_PRIOR
        float th[D];
#pragma unroll
        for (int j = 0; j < D; ++j) th[j] = a.in[r * D + j];
        a.out[r] = glabc_user_prior_log_prob(th);
#endif
        return;
    }
    float yy[YD];
#pragma unroll
    for (int j = 0; j < YD; ++j) yy[j] = a.in[r * YD + j];
    const float dis = model_discrepancy_rows<YD>(yy, a.y_obs);
    a.out[r] = a.what == 1 ? dis : model_log_kernel_of(dis, a.kern_scale, a.kern_log_scale, a.kern_c0);
}

}  // namespace glabc
