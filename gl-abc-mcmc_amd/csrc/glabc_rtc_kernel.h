// glabc_rtc_kernel.h -- what hiprtc compiles around a USER simulator (glabc_rtc.hip).
//
// The translation unit hiprtc sees is
//     <fixed-width typedefs>
//     #define GLABC_RTC_ALGO / _D / _YD / _N / _L        the configuration (L = lanes per chain, chosen by the host)
//     #define GLABC_USER_SIM 1, GLABC_USER_NOISE_DIM    make model_simulate call the user's function
//     #include "glabc_numerics.h"                        so that the user's source can use glabc_expf, glabc_logf, ...
//     <the user's source: GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)>
//     #include "glabc_rtc_kernel.h"
// i.e. an explicit instantiation of sampler_kernel (glabc_sampler.h) -- the SAME template the library's built-in kernels are
// instantiated from -- with the user's function inlined where the built-in simulators are; the host fills the same
// StepArgs<D, YD> argument block as for the built-in kernels (glabc_pack.h).
#pragma once

#include "glabc_sampler.h"

namespace glabc {

template __global__ void sampler_kernel<GLABC_RTC_ALGO, GLABC_RTC_D, GLABC_RTC_YD, GLABC_RTC_N, GLABC_RTC_L, VAR_GENERIC, 0>(
    const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>);
#if GLABC_RTC_YD == GLABC_RTC_D
// ... and the branch-free variant for unit-scale Gaussian prior / global proposal (chosen per launch by the host, as for the
// built-in kernels: glabc_pack.h gauss_unit_config)
template __global__ void sampler_kernel<GLABC_RTC_ALGO, GLABC_RTC_D, GLABC_RTC_YD, GLABC_RTC_N, GLABC_RTC_L, VAR_GAUSS_UNIT, 0>(
    const StepArgs<GLABC_RTC_D, GLABC_RTC_YD>);
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

}  // namespace glabc
