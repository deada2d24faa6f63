// glabc_mala_dim.hip -- instantiates the GLMALA kernels for ONE theta_dim (-DGLABC_DIM=d),
// batch sizes 1..GLABC_MAX_BATCH.
#include "glabc_mala.h"

#ifndef GLABC_DIM
#error "compile with -DGLABC_DIM=<theta_dim>"
#endif

namespace glabc {

template <int D, int N>
static int launch_mala(const MalaArgs<D>& m, hipStream_t s)
{
    const unsigned grid = (unsigned)((m.s.n_chains + 63) / 64);
    if constexpr (D == 2) {
        // theta_dim 2: launches of fewer than two wavefronts of glmala_kernel per SIMD (1024 SIMDs) run as teams of two wavefronts
        // per 64 chains (glabc_mala.h glmala_team_kernel; lanes_per_chain 1 / 2 force one / two wavefronts)
        const int nw = m.lanes ? m.lanes : (grid < 2048u ? 2 : 1);
        if (nw == 2) {
            if (m.s.y_obs_away) hipLaunchKernelGGL((glmala_team_kernel<N, true, 2>), dim3(grid), dim3(128), 0, s, m);
            else hipLaunchKernelGGL((glmala_team_kernel<N, false, 2>), dim3(grid), dim3(128), 0, s, m);
            return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
        }
    }
    if (m.s.y_obs_away)
        hipLaunchKernelGGL((glmala_kernel<D, N, true>), dim3(grid), dim3(64), 0, s, m);
    else
        hipLaunchKernelGGL((glmala_kernel<D, N, false>), dim3(grid), dim3(64), 0, s, m);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

template <>
int launch_glmala_dim<GLABC_DIM>(int n_batch, const MalaArgs<GLABC_DIM>& m, hipStream_t s)
{
    constexpr int D = GLABC_DIM;
    switch (n_batch) {
#define GLABC_CASE(n) case n: return launch_mala<D, n>(m, s);
        GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) GLABC_CASE(5) GLABC_CASE(6) GLABC_CASE(7) GLABC_CASE(8)
        GLABC_CASE(9) GLABC_CASE(10) GLABC_CASE(11) GLABC_CASE(12) GLABC_CASE(13) GLABC_CASE(14) GLABC_CASE(15) GLABC_CASE(16)
#undef GLABC_CASE
    default: return GLABC_ERR_ARG;
    }
}

template <>
int launch_glmala_init_dim<GLABC_DIM>(const MalaArgs<GLABC_DIM>& m, hipStream_t s)
{
    const unsigned grid = (unsigned)((m.s.n_chains + 63) / 64);
    hipLaunchKernelGGL((glmala_init_kernel<GLABC_DIM>), dim3(grid), dim3(64), 0, s, m);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

}  // namespace glabc
