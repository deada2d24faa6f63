// glabc_rtc_kernel.h -- the fused sampler around a USER simulator, compiled at run time (hiprtc) by glabc_rtc.hip.
//
// The translation unit hiprtc sees is
//     <fixed-width typedefs>
//     #define GLABC_RTC_ALGO / _D / _YD / _ND / _N      the configuration
//     #define GLABC_USER_SIM 1, GLABC_USER_NOISE_DIM    make model_simulate call the user's function
//     #include "glabc_numerics.h"                        so that the user's source can use glabc_expf, glabc_logf, ...
//     <the user's source: GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)>
//     #include "glabc_rtc_kernel.h"
// i.e. chain_step / sampler_body of glabc_device.h / glabc_sampler.h -- the SAME code as the library's built-in kernels --
// with the user's function inlined where the built-in simulators are.  The entry point takes one plain-C argument block
// (RtcArgs: the descriptors of include/glabc.h by value) so that the host needs no per-(D, YD) struct layouts, and turns it into
// the StepArgs the sampler code reads (uniform values: they stay in SGPRs).
#pragma once

#include "glabc_sampler.h"

namespace glabc {

struct RtcArgs {
    glabc_model model;
    glabc_dist local, global;
    glabc_chains chains;
    uint64_t seed;
    uint32_t step0;
    int32_t n_steps;
    float gf;
    int32_t exact_index;
    const float* gf_chain;
    float* history;
    int64_t hist_stride;
    double* sum_theta;
    double* sum_outer;
    double* sum_jump;
};

#if defined(__HIPCC_RTC__)
template <int D>
GLABC_DEV DistArgs<D> rtc_dist(const glabc_dist& g)
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
        unit = unit && (g.p2[j] == 1.0f) && (g.p1[j] == 0.0f);
    }
    o.unit_scale = unit ? 1 : 0;
    return o;
}

extern "C" __global__ void __launch_bounds__(BLOCK) glabc_rtc_entry(const RtcArgs r)
{
    constexpr int D = GLABC_RTC_D, YD = GLABC_RTC_YD;
    StepArgs<D, YD> a;
    a.prior = rtc_dist<D>(r.model.prior);
    a.sim_kind = r.model.sim_kind;
    a.gk_c = 0.0f;
#pragma unroll
    for (int j = 0; j < YD; ++j) {
        a.noise_loc[j] = 0.0f;
        a.noise_scale[j] = 1.0f;
        a.y_obs[j] = r.model.y_obs[j];
    }
    a.kern_log_scale = r.model.kern_log_scale;
    a.kern_scale = r.model.kern_scale;
    a.kern_c0 = r.model.kern_c0;
    a.y_obs_away = 0;
    a.kern_rinv = 0.0f;
    a.local = rtc_dist<D>(r.local);
    a.global = rtc_dist<D>(r.global);
    a.theta = r.chains.theta;
    a.y = r.chains.y;
    a.log_w = r.chains.log_w;
    a.flags = r.chains.flags;
    a.n_moves = r.chains.n_moves;
    a.n_chains = r.chains.n_chains;
    a.chain0 = r.chains.chain0;
    a.stride = r.chains.stride;
    a.seed_lo = (uint32_t)r.seed;
    a.seed_hi = (uint32_t)(r.seed >> 32);
    a.step0 = r.step0;
    a.n_steps = r.n_steps;
    a.gf = r.gf;
    a.gf_chain = r.gf_chain;
    a.history = r.history;
    a.hist_stride = r.hist_stride;
    a.sum_theta = r.sum_theta;
    a.sum_outer = r.sum_outer;
    a.sum_jump = r.sum_jump;
    a.tape_u = nullptr;
    a.tape_r = nullptr;
    a.tape_z = nullptr;
    a.tape_nprop = 0;
    a.exact_index = r.exact_index;
    sampler_body<GLABC_RTC_ALGO, D, YD, GLABC_RTC_N, 1, VAR_GENERIC>(a);
}

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
#endif

}  // namespace glabc
