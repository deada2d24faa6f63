// glabc_team_dim.hip -- instantiates team_sampler_kernel (glabc_team.h) for ONE theta_dim (-DGLABC_DIM=d [-DGLABC_YDIM=yd]),
// batch sizes 2..GLABC_MAX_BATCH as far as a workgroup's two candidate buffers fit the LDS budget.
#include "glabc_pack.h"
#include "glabc_team.h"

#ifndef GLABC_DIM
#error "compile with -DGLABC_DIM=<theta_dim> [-DGLABC_YDIM=<y_dim>]"
#endif
#ifndef GLABC_YDIM
#define GLABC_YDIM GLABC_DIM
#endif

namespace glabc {

template <int D, int YD, int N, int NW, bool FAST>
static int launch_team(const StepArgs<D, YD>& a, int prio, hipStream_t s)
{
    if constexpr (team_config_ok(D, YD, N, NW) && (!FAST || YD == D)) {
        const unsigned grid = (unsigned)((a.n_chains + 63) / 64);
        if (a.prior.kind == GLABC_DIST_GAMMA || a.global.kind == GLABC_DIST_GAMMA) {
            // VAR_GAMMA (float64 draws and densities): teams of two and three wavefronts, the |theta| + noise simulator, exact arithmetic
            if constexpr (NW <= 3 && YD == D && !FAST)
                hipLaunchKernelGGL((team_sampler_kernel<D, YD, N, VAR_GAMMA, NW, false>), dim3(grid), dim3(64 * NW), 0, s, a, prio);
            else
                return GLABC_ERR_ARG;
        } else if (gauss_unit_config<D, YD>(a))
            hipLaunchKernelGGL((team_sampler_kernel<D, YD, N, (YD == D ? VAR_GAUSS_UNIT : VAR_GENERIC), NW, FAST>), dim3(grid), dim3(64 * NW), 0, s, a, prio);
        else
            hipLaunchKernelGGL((team_sampler_kernel<D, YD, N, VAR_GENERIC, NW, FAST>), dim3(grid), dim3(64 * NW), 0, s, a, prio);
        return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
    } else {
        return GLABC_ERR_ARG;
    }
}

template <int D, int YD, int N>
static int launch_team_nw(int nw, const StepArgs<D, YD>& a, int prio, bool fast, hipStream_t s)
{
    if (fast) {                                             // GLABC_MATH_FAST: teams of two or three wavefronts
        switch (nw) {
        case 2: return launch_team<D, YD, N, 2, true>(a, prio, s);
        case 3: return launch_team<D, YD, N, 3, true>(a, prio, s);
        default: return GLABC_ERR_ARG;
        }
    }
    switch (nw) {
    case 2: return launch_team<D, YD, N, 2, false>(a, prio, s);
    case 3: return launch_team<D, YD, N, 3, false>(a, prio, s);
    case 4: return launch_team<D, YD, N, 4, false>(a, prio, s);
    default: return GLABC_ERR_ARG;
    }
}

template <>
int launch_team_dim<GLABC_DIM, GLABC_YDIM>(int n_batch, int nw, const StepArgs<GLABC_DIM, GLABC_YDIM>& a, int prio, bool fast, hipStream_t s)
{
    constexpr int D = GLABC_DIM, YD = GLABC_YDIM;
    switch (n_batch) {
#define GLABC_CASE(n) case n: return launch_team_nw<D, YD, n>(nw, a, prio, fast, s);
        GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
        GLABC_CASE(9) GLABC_CASE(10) GLABC_CASE(11) GLABC_CASE(12) GLABC_CASE(13) GLABC_CASE(14) GLABC_CASE(15) GLABC_CASE(16)
#undef GLABC_CASE
    default: return GLABC_ERR_ARG;
    }
}

template <int NW>
static int launch_global_team(const StepArgs<GLABC_DIM, GLABC_YDIM>& a, int prio, hipStream_t s)
{
    constexpr int D = GLABC_DIM, YD = GLABC_YDIM;
    const unsigned grid = (unsigned)((a.n_chains + 63) / 64);
    if (YD == D && gauss_unit_config<D, YD>(a))
        hipLaunchKernelGGL((global_team_kernel<D, YD, (YD == D ? VAR_GAUSS_UNIT : VAR_GENERIC), NW>), dim3(grid), dim3(64 * NW), 0, s, a, prio);
    else
        hipLaunchKernelGGL((global_team_kernel<D, YD, VAR_GENERIC, NW>), dim3(grid), dim3(64 * NW), 0, s, a, prio);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

template <>
int launch_global_team_dim<GLABC_DIM, GLABC_YDIM>(int nw, const StepArgs<GLABC_DIM, GLABC_YDIM>& a, int prio, hipStream_t s)
{
    if (a.prior.kind == GLABC_DIST_GAMMA || a.global.kind == GLABC_DIST_GAMMA) return GLABC_ERR_ARG;      // one lane per chain
    return nw == 3 ? launch_global_team<3>(a, prio, s) : launch_global_team<2>(a, prio, s);
}

}  // namespace glabc
