// glabc_pack.h -- host side: glabc_model / glabc_dist / glabc_chains / glabc_run -> the kernels' argument block StepArgs<D, YD>.
// Shared by glabc_hip.hip (built-in kernels) and glabc_rtc.hip (run-time compiled kernels).
#pragma once

#include <cmath>
#include <cstring>

#include "glabc_device.h"

namespace glabc {

template <int D>
inline DistArgs<D> pack_dist(const glabc_dist* g)
{
    DistArgs<D> o;
    o.kind = g->kind;
    o.c0 = g->c0;
    bool unit = g->kind == GLABC_DIST_DIAG_GAUSS;
    for (int j = 0; j < D; ++j) {
        o.p0[j] = g->p0[j];
        o.p1[j] = g->p1[j];
        o.p2[j] = g->p2[j];
        o.p3[j] = g->p3[j];
        unit = unit && (g->p2[j] == 1.0f) && (g->p1[j] == 0.0f);
    }
    o.unit_scale = unit ? 1 : 0;
    return o;
}

// RN(1/s) if the three-instruction division of model_log_kernel (IEEE mul + two fused multiply-adds) equals the IEEE quotient for
// EVERY float32 dividend with this divisor -- checked exhaustively on the host, ~7 ms, remembered per divisor -- else 0
// (glabc_hip.hip)
float verified_reciprocal(float s);

// the all-DiagGaussian, unit-scale prior / global configuration (the reference example's) gets the branch-free kernel variant
// VAR_GAUSS_UNIT (same results bit for bit: tests/test_hip_parity.py::test_unit_gaussian_variant_and_its_fallback)
template <int D, int YD>
inline bool gauss_unit_config(const StepArgs<D, YD>& a)
{
    return YD == D && a.prior.kind == GLABC_DIST_DIAG_GAUSS && a.prior.unit_scale && a.global.kind == GLABC_DIST_DIAG_GAUSS &&
           a.global.unit_scale && a.local.kind == GLABC_DIST_DIAG_GAUSS && a.y_obs_away != 0 && a.kern_rinv != 0.0f;
}

// kern_rinv: verified_reciprocal(kern_scale) or 0
template <int D, int YD = D>
inline StepArgs<D, YD> pack_args_rinv(const glabc_model* m, const glabc_dist* local, const glabc_dist* global,
                                      const glabc_chains* c, const glabc_run* r, float kern_rinv)
{
    StepArgs<D, YD> a;
    std::memset(&a, 0, sizeof a);
    a.prior = pack_dist<D>(&m->prior);
    a.sim_kind = m->sim_kind;
    a.gk_c = m->gk_c;
    for (int j = 0; j < YD; ++j) {
        a.noise_loc[j] = m->noise.p0[j];
        a.noise_scale[j] = m->noise.p2[j];
        a.y_obs[j] = m->y_obs[j];
    }
    a.y_obs_away = 1;
    for (int j = 0; j < YD; ++j) a.y_obs_away = a.y_obs_away && (std::fabs(m->y_obs[j]) >= 0x1p-6f);
    a.kern_log_scale = m->kern_log_scale;
    a.kern_scale = m->kern_scale;
    a.kern_c0 = m->kern_c0;
    a.kern_rinv = kern_rinv;
    a.local = pack_dist<D>(local ? local : global);
    a.global = pack_dist<D>(global);
    a.theta = c->theta;
    a.y = c->y;
    a.log_w = c->log_w;
    a.flags = c->flags;
    a.n_moves = c->n_moves;
    a.n_chains = c->n_chains;
    a.chain0 = c->chain0;
    a.stride = c->stride;
    if (r) {
        a.seed_lo = (uint32_t)r->seed;
        a.seed_hi = (uint32_t)(r->seed >> 32);
        a.step0 = r->step0;
        a.n_steps = r->n_steps;
        a.exact_index = (r->debug_flags & GLABC_DEBUG_EXACT_INDEX) ? 1 : 0;
        a.gf = r->global_frequency;
        a.gf_chain = r->global_frequency_per_chain;
        a.history = r->history;
        a.hist_stride = r->hist_stride;
        if (r->moments) {
            a.sum_theta = r->moments->sum_theta;
            a.sum_outer = r->moments->sum_outer;
            a.sum_jump = r->moments->sum_jump;
        }
        if (r->math_mode == GLABC_MATH_FAST && r->dump_draws) {
            a.dump_u = r->dump_draws->u;
            a.dump_r = r->dump_draws->r;
            a.dump_z = r->dump_draws->z;
        }
        if (r->tape) {
            a.tape_u = r->tape->u;
            a.tape_r = r->tape->r;
            a.tape_z = r->tape->z;
            a.tape_nprop = r->tape->n_prop;
        }
    }
    return a;
}

}  // namespace glabc
