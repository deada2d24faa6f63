// glabc_sampler_dim.hip -- instantiates sampler_kernel for ONE theta_dim (-DGLABC_DIM=d):
// GLMCMC for batch sizes 1..GLABC_MAX_BATCH x lanes-per-chain {1,2,4}, GlobalMCMC (one lane).
#include "glabc_sampler.h"

#ifndef GLABC_DIM
#error "compile with -DGLABC_DIM=<theta_dim> [-DGLABC_YDIM=<y_dim>]"
#endif
#ifndef GLABC_YDIM
#define GLABC_YDIM GLABC_DIM
#endif
#ifndef GLABC_SCHED
#define GLABC_SCHED 0
#endif

namespace glabc {

// the all-DiagGaussian, unit-scale prior / global configuration gets the branch-free variant
template <int D, int YD>
static bool gauss_unit(const StepArgs<D, YD>& a)
{
    const bool y_obs_away_from_zero = a.y_obs_away != 0;    // lets the variant use the lean square root (model_log_kernel)
    return a.prior.kind == GLABC_DIST_DIAG_GAUSS && a.prior.unit_scale && a.global.kind == GLABC_DIST_DIAG_GAUSS &&
           a.global.unit_scale && a.local.kind == GLABC_DIST_DIAG_GAUSS && y_obs_away_from_zero && a.kern_rinv != 0.0f;
}

template <int ALGO, int D, int YD, int N, int L>
static int launch_one(const StepArgs<D, YD>& a, hipStream_t s)
{
    const int64_t lanes = a.n_chains * L;
    const unsigned grid = (unsigned)((lanes + BLOCK - 1) / BLOCK);
    if (a.tape_u) {
        if constexpr (L == 1)
            hipLaunchKernelGGL((sampler_kernel<ALGO, D, YD, N, 1, VAR_TAPE, GLABC_SCHED>), dim3(grid), dim3(BLOCK), 0, s, a);
        else
            return GLABC_ERR_ARG;
    } else if (a.prior.kind == GLABC_DIST_GAMMA || a.global.kind == GLABC_DIST_GAMMA) {
#if GLABC_SCHED == 0
        if constexpr (L == 1 && D <= 4 && YD == D)
            hipLaunchKernelGGL((sampler_kernel<ALGO, D, YD, N, 1, VAR_GAMMA, GLABC_SCHED>), dim3(grid), dim3(BLOCK), 0, s, a);
        else
            return GLABC_ERR_KIND;
#else
        return GLABC_ERR_KIND;                                      // the Gamma variant lives in the default-schedule objects
#endif
    } else if (YD == D && gauss_unit<D, YD>(a))
        hipLaunchKernelGGL((sampler_kernel<ALGO, D, YD, N, L, (YD == D ? VAR_GAUSS_UNIT : VAR_GENERIC), GLABC_SCHED>), dim3(grid), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((sampler_kernel<ALGO, D, YD, N, L, VAR_GENERIC, GLABC_SCHED>), dim3(grid), dim3(BLOCK), 0, s, a);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

template <int D, int YD, int N>
static int launch_lanes(int lanes, const StepArgs<D, YD>& a, hipStream_t s)
{
#if GLABC_SCHED == 0
    if constexpr (N >= 2) {
        if (lanes == 2) return launch_one<ALGO_GLMCMC, D, YD, N, 2>(a, s);
    }
    if constexpr (N >= 3) {
        if (lanes == 4) return launch_one<ALGO_GLMCMC, D, YD, N, 4>(a, s);
    }
#else
    if (lanes != 1) return GLABC_ERR_ARG;                       // the max-ilp objects hold one-lane kernels only
#endif
    return launch_one<ALGO_GLMCMC, D, YD, N, 1>(a, s);
}

template <>
int launch_sampler_dim<GLABC_DIM, GLABC_YDIM, GLABC_SCHED>(int algo, int n_batch, int lanes, const StepArgs<GLABC_DIM, GLABC_YDIM>& a,
                                              hipStream_t s)
{
    constexpr int D = GLABC_DIM, YD = GLABC_YDIM;
    if (algo == ALGO_GLOBAL) return launch_one<ALGO_GLOBAL, D, YD, 1, 1>(a, s);
    switch (n_batch) {
#define GLABC_CASE(n) case n: return launch_lanes<D, YD, n>(lanes, a, s);
        GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
        GLABC_CASE(9) GLABC_CASE(10) GLABC_CASE(11) GLABC_CASE(12) GLABC_CASE(13) GLABC_CASE(14) GLABC_CASE(15) GLABC_CASE(16)
#undef GLABC_CASE
    default: return GLABC_ERR_ARG;
    }
}

}  // namespace glabc
