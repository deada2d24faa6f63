// glabc_mala.h -- fused GLMALA step (GLMALA.py:118-230) for gfx950.
//
// One work-item owns one chain.  The chain's state lives in VGPRs as doubles together with the
// two sticky precision bits of include/glabc.h (GLABC_FLAG_TH64 / GLABC_FLAG_LW64): the reference's
// tensors change dtype while a chain runs, so every density is evaluated in the precision the
// reference would be using at that point (float32-exact values are carried in doubles).
//
// The MALA drift is the reference's common-random-number central-difference gradient of a
// synthetic-likelihood log-ABC: 2 * theta_dim * num_grad simulator calls per local move
// (numberical_gradient_logABC, GLMALA.py:46-95), shared by the lanes of the wavefront (coop_gradient:
// exact fixed-point sums, so the split over lanes cannot change a bit).  Its noise comes from Philox
// slots GRAD_BASE + g*GRAD_STRIDE + block (normal n = (k*num + s)*y_dim + j in block n/4), shared by
// the +d and -d simulations of a coordinate as the reference's reseeding does (GLMALA.py:76,80).
//
// The operation order is the reference's, line by line (citations on the right); the CPU checker
// restates the same lines independently in plain C and is compared bit for bit in the tests.
#pragma once

#include "glabc_device.h"

namespace glabc {

constexpr uint32_t GRAD_BASE = 0x100000u;
constexpr uint32_t GRAD_STRIDE = 0x80000u;

template <int D>
struct MalaArgs {
    StepArgs<D> s;               // model, importance proposal (s.global), chains, run
    double* theta64;
    double* y64;
    double* log_w64;
    double* grad;
    double tau, tau_sq, eps_sq;
    int32_t num_grad;
    int32_t lanes;               // glabc_run.lanes_per_chain: 0 = choose, 1 = one wavefront per 64 chains (glmala_kernel), 2 = a team
                                 // of two wavefronts per 64 chains (glmala_team_kernel, theta_dim 2)
    int32_t credit;              // team kernel: gradient work items the main wavefront's lanes are let off for running the iSIR move
    int32_t prio;                // team kernel: s_setprio of the main wavefront (it carries the serial part of an iteration)
};

// distribution.py:176-181 / 81-86 on a float64 tensor
template <int D>
GLABC_DEV double dist_log_prob_f64(const DistArgs<D>& g, const double (&z)[D])
{
    if (g.kind == GLABC_DIST_DIAG_GAUSS) {
        double t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double e = (z[j] - (double)g.p0[j]) / (double)g.p2[j];
            t[j] = (double)g.p1[j] + 0.5 * (e * e);
        }
        return (-0.5 * (double)D * GLABC_LOG_2PI) - aten_rowsum_f64<D>(t);
    }
    bool out = false;
#pragma unroll
    for (int j = 0; j < D; ++j) out = out || (z[j] < (double)g.p0[j]) || (z[j] > (double)g.p1[j]);
    return out ? -__builtin_inf() : (double)g.c0;
}

// Mixture.py:33-45 on a float64 y
template <int D>
GLABC_DEV double model_log_kernel_f64(const StepArgs<D>& a, const double (&y)[D])
{
    double t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double d = y[j] - (double)a.y_obs[j];
        t[j] = d * d;
    }
    double dis = __builtin_sqrt(aten_rowsum_f64<D>(t));
    double e = (dis - 0.0) / (double)a.kern_scale;
    return (-0.5 * 1.0 * GLABC_LOG_2PI) - ((double)a.kern_log_scale + 0.5 * (e * e));
}

// discrepancy, Mixture.py:33-36, float32.  LEAN: the launcher knows every |y_obs_j| >= 2^-6 (StepArgs::y_obs_away), so
// the sum of squares is 0 or >= 2^-62 and the square root without the rescaling of tiny arguments is the correctly
// rounded one (glabc_numerics.h) -- 8 instructions less, 400 times per gradient
template <int D, bool LEAN = false>
GLABC_DEV float model_discrepancy(const StepArgs<D>& a, const float (&y)[D])
{
    float t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float d = y[j] - a.y_obs[j];
        t[j] = d * d;
    }
    const float ss = aten_rowsum<D>(t);
    return LEAN ? glabc_sqrtf_normal(ss) : __builtin_sqrtf(ss);
}

template <int D>
struct MalaChain {
    double theta[D], y[D], grad[D];
    double log_w;
    uint32_t flags, n_moves;
};

template <int D>
GLABC_DEV double state_prior(const StepArgs<D>& a, const MalaChain<D>& c)
{
    if (c.flags & GLABC_FLAG_TH64) return dist_log_prob_f64<D>(a.prior, c.theta);
    float th[D];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = (float)c.theta[j];
    return (double)dist_log_prob<D>(a.prior, th);
}

template <int D>
GLABC_DEV double state_kernel(const StepArgs<D>& a, const MalaChain<D>& c)
{
    if (c.flags & GLABC_FLAG_TH64) return model_log_kernel_f64<D>(a, c.y);
    float y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) y[j] = (float)c.y[j];
    return (double)model_log_kernel<D>(a, y);
}

// 64-bit / double lane exchange helpers (two 32-bit moves)
GLABC_DEV uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
GLABC_DEV double shfl_f64(double v, int src)
{
    const uint64_t u = glabc_d2u(v);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)u, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(u >> 32), src, 64);
    return glabc_u2d(((uint64_t)hi << 32) | lo);
}

// numberical_gradient_logABC, GLMALA.py:46-95, wave-cooperative.
//
// LDS of one wavefront's gradient work (one wavefront per workgroup)
struct GradShared {
    unsigned long long acc[64][8];   // theta_dim 2: per needing chain (s1, a, b, c) of the + and the - side, glabc_fxsplit
    float th[64][2];                 //              its theta
    uint32_t c0[64], c1[64];         //              its Philox counter words (global chain id)
    int srcmap[64];                  // rank -> lane of the needing chains
};

// theta_dim 2: the n needing chains' work -- per coordinate n x nb Philox blocks of two simulations each -- is dealt to
// the 64 lanes as CONTIGUOUS ranges of ceil(n nb / 64) items, whatever n is (lane groups of a power of two leave up to a
// third of the wavefront idle: 13 chains -> 16 groups of 4 lanes, 13 rounds instead of 10.4).  A lane's range touches at
// most two chains; it keeps the current chain's partial sums in registers and adds them to that chain's accumulators in
// LDS (64-bit integer atomics: exact, order-free) when the chain changes and at the end; the owners then finish the
// statistics of their own chain.
template <bool LEAN>
GLABC_DEV void coop_gradient_flat2(const MalaArgs<2>& m, const Rng& rng, uint32_t step, int g, bool need, unsigned long long mask,
                                   const float (&theta)[2], double (&grad)[2], GradShared* sh)
{
    constexpr int D = 2;
    const StepArgs<D>& a = m.s;
    const int lane = (int)(threadIdx.x & 63u);
    const int n = __popcll(mask);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    const int num = m.num_grad;
    const float h = 0.1f, hp = 0.00001f;
    __syncthreads();                                                    // earlier readers of the shared block are done
    if (need) {
        sh->th[rank][0] = theta[0];
        sh->th[rank][1] = theta[1];
        sh->c0[rank] = rng.c0;
        sh->c1[rank] = rng.c1;
#pragma unroll
        for (int j = 0; j < 8; ++j) sh->acc[rank][j] = 0ull;
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < D; ++k) {
        const int64_t first = (int64_t)k * num;                         // global simulation index of s = 0
        const int64_t b_lo = first >> 1, b_hi = (first + num - 1) >> 1; // a Philox block = the normals of two simulations
        const int nb = (int)(b_hi - b_lo + 1);
        const int total = n * nb;
        const int q = (total + 63) >> 6;
        const int it0 = lane * q;
        const int it1 = (it0 + q < total) ? it0 + q : total;
        int cur = (it0 < total) ? it0 / nb : 0, bi = (it0 < total) ? it0 - cur * nb : 0;
        float tp[D], tm[D];
        double c_p = 0.0, c_m = 0.0;
        uint32_t cc0 = 0u, cc1 = 0u;
        glabc_fxsplit ap = {0, 0, 0, 0}, am = {0, 0, 0, 0};
        auto enter = [&](int c) {                                      // this coordinate's constants of chain c
            float th[D], zero[D], y0p[D], y0m[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                th[j] = sh->th[c][j];
                tp[j] = th[j] + (j == k ? h : 0.0f);                    // GLMALA.py:67
                tm[j] = th[j] - (j == k ? h : 0.0f);                    // GLMALA.py:68
                zero[j] = 0.0f;
            }
            model_simulate<D>(a, tp, zero, y0p);                        // centres: the noise-free discrepancies
            model_simulate<D>(a, tm, zero, y0m);
            c_p = (double)model_discrepancy<D, LEAN>(a, y0p);
            c_m = (double)model_discrepancy<D, LEAN>(a, y0m);
            cc0 = sh->c0[c];
            cc1 = sh->c1[c];
        };
        auto flush = [&](int c) {
            atomicAdd(&sh->acc[c][0], (unsigned long long)ap.s1);
            atomicAdd(&sh->acc[c][1], ap.a);
            atomicAdd(&sh->acc[c][2], (unsigned long long)ap.b);
            atomicAdd(&sh->acc[c][3], ap.c);
            atomicAdd(&sh->acc[c][4], (unsigned long long)am.s1);
            atomicAdd(&sh->acc[c][5], am.a);
            atomicAdd(&sh->acc[c][6], (unsigned long long)am.b);
            atomicAdd(&sh->acc[c][7], am.c);
            ap = glabc_fxsplit{0, 0, 0, 0};
            am = glabc_fxsplit{0, 0, 0, 0};
        };
        auto one_sim = [&](const float (&eps)[D]) {
            float yp[D], ym[D];
            model_simulate<D>(a, tp, eps, yp);                          // GLMALA.py:78
            model_simulate<D>(a, tm, eps, ym);                          // GLMALA.py:82 (same noise)
            glabc_fxs_add(&ap, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, yp) - c_p));
            glabc_fxs_add(&am, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, ym) - c_m));
        };
        if (it0 < it1) enter(cur);
#pragma unroll 1
        for (int it = it0; it < it1; ++it) {
            const int64_t b = b_lo + bi;
            glabc_u32x4 blk = glabc_philox4x32_10(cc0, cc1, step, GRAD_BASE + (uint32_t)g * GRAD_STRIDE + (uint32_t)b, rng.k0, rng.k1);
            float e0[2], e1[2];
            glabc_normal_pair(blk.v[0], blk.v[1], &e0[0], &e0[1]);
            glabc_normal_pair(blk.v[2], blk.v[3], &e1[0], &e1[1]);
            const int64_t s0 = 2 * b - first, s1 = s0 + 1;              // simulation indices within coordinate k
            if (s0 >= 0 && s0 < num) one_sim(e0);
            if (s1 >= 0 && s1 < num) one_sim(e1);
            if (++bi == nb && it + 1 < it1) {                           // the range crosses into the next chain
                flush(cur);
                bi = 0;
                ++cur;
                enter(cur);
            }
        }
        if (it0 < it1) flush(cur);
        __syncthreads();
        if (need) {                                                     // the owner finishes its chain's coordinate k
            glabc_fxsplit sp, sm;
            sp.s1 = (int64_t)sh->acc[rank][0]; sp.a = sh->acc[rank][1]; sp.b = (int64_t)sh->acc[rank][2]; sp.c = sh->acc[rank][3];
            sm.s1 = (int64_t)sh->acc[rank][4]; sm.a = sh->acc[rank][5]; sm.b = (int64_t)sh->acc[rank][6]; sm.c = sh->acc[rank][7];
#pragma unroll
            for (int j = 0; j < 8; ++j) sh->acc[rank][j] = 0ull;        // ready for the next coordinate
            float op[D], om[D], zero[D], y0p[D], y0m[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                op[j] = theta[j] + (j == k ? h : 0.0f);
                om[j] = theta[j] - (j == k ? h : 0.0f);
                zero[j] = 0.0f;
            }
            model_simulate<D>(a, op, zero, y0p);
            model_simulate<D>(a, om, zero, y0m);
            const double o_p = (double)model_discrepancy<D, LEAN>(a, y0p), o_m = (double)model_discrepancy<D, LEAN>(a, y0m);
            const glabc_fxsum fp = glabc_fxs_finish(&sp), fm = glabc_fxs_finish(&sm);
            const double nd = (double)num;
            const double s1p = glabc_fx_sum1(&fp), s2p = glabc_fx_sum2(&fp), s1m = glabc_fx_sum1(&fm), s2m = glabc_fx_sum2(&fm);
            const double mu_p = o_p + s1p / nd, mu_m = o_m + s1m / nd;                           // GLMALA.py:86-87
            const double var_p = (s2p - (s1p * s1p) / nd) / (nd - 1.0), var_m = (s2m - (s1m * s1m) / nd) / (nd - 1.0);   // :88-89
            const double lp = (-0.5 * glabc_log(var_p + m.eps_sq)) - ((0.5 * (mu_p * mu_p)) / (var_p + m.eps_sq));   // :90-91
            const double lm = (-0.5 * glabc_log(var_m + m.eps_sq)) - ((0.5 * (mu_m * mu_m)) / (var_m + m.eps_sq));   // :92-93
            const double gll = (lp - lm) / 0.2;                                                  // :94
#pragma unroll
            for (int j = 0; j < D; ++j) {
                op[j] = theta[j] + (j == k ? hp : 0.0f);                                         // :84
                om[j] = theta[j] - (j == k ? hp : 0.0f);
            }
            const float gp = (dist_log_prob<D>(a.prior, op) - dist_log_prob<D>(a.prior, om)) / 0.00002f;   // :84-85
            const double gk = gll + (double)gp;                                                  // :95
#pragma unroll
            for (int j = 0; j < D; ++j)
                if (j == k) grad[j] = gk;
        }
        __syncthreads();                                                // accumulators are zero again before anyone adds
    }
}

// Every lane of the wavefront calls this; `need` marks the lanes (chains) that want the gradient of their `theta`
// (float32, GLMALA.py:62).  The n needing chains are dealt to lane groups of G = 64 / 2^ceil(log2 n) lanes; the
// lanes of a group split the chain's num_grad simulations per coordinate (s = sub, sub + G, ...), accumulate the
// shifted discrepancies in exact fixed point (glabc_fxsum: integer sums, so the split cannot change a bit), merge
// the partial sums with xor-shuffles and all finish the statistics; the result goes back to the owning lane.
// srcmap: 64 ints of LDS (one wavefront per workgroup).
template <int D, bool LEAN = false>
GLABC_DEV void coop_gradient(const MalaArgs<D>& m, const Rng& rng, uint32_t step, int g, bool need, const float (&theta)[D],
                             double (&grad)[D], GradShared* sh)
{
    const StepArgs<D>& a = m.s;
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return;                                           // wave-uniform
    if constexpr (D == 2) {
        coop_gradient_flat2<LEAN>(m, rng, step, g, need, mask, theta, grad, sh);
        return;
    }
    int* srcmap = sh->srcmap;
    const int lane = (int)(threadIdx.x & 63u);
    const int n = __popcll(mask);
    int G = 64;
    while (G * n > 64) G >>= 1;                                         // wave-uniform
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();                                                    // previous readers of srcmap are done
    if (need) srcmap[rank] = lane;
    __syncthreads();
    const int grp = lane / G, sub = lane - grp * G;
    const int src = srcmap[grp < n ? grp : 0];                          // idle groups shadow chain 0 (results unused)
    float th[D];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = __shfl(theta[j], src, 64);
    Rng r2;
    r2.c0 = (uint32_t)__shfl((int)rng.c0, src, 64);
    r2.c1 = (uint32_t)__shfl((int)rng.c1, src, 64);
    r2.k0 = rng.k0;
    r2.k1 = rng.k1;
    const int num = m.num_grad;
    const int rounds = (num + G - 1) / G;
    const float h = 0.1f, hp = 0.00001f;
    double gout[D];
#pragma unroll 1
    for (int k = 0; k < D; ++k) {
        float tp[D], tm[D], zero[D], y0p[D], y0m[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            tp[j] = th[j] + (j == k ? h : 0.0f);                        // GLMALA.py:67
            tm[j] = th[j] - (j == k ? h : 0.0f);                        // GLMALA.py:68
            zero[j] = 0.0f;
        }
        model_simulate<D>(a, tp, zero, y0p);                            // centres: the noise-free discrepancies
        model_simulate<D>(a, tm, zero, y0m);
        const double c_p = (double)model_discrepancy<D, LEAN>(a, y0p), c_m = (double)model_discrepancy<D, LEAN>(a, y0m);
        glabc_fxsplit ap = {0, 0, 0, 0}, am = {0, 0, 0, 0};             // exact sums, split squares (glabc_numerics.h)
        // one simulation of coordinate k with noise index s
        auto one_sim = [&](const float (&eps)[D]) {
            float yp[D], ym[D];
            model_simulate<D>(a, tp, eps, yp);                          // GLMALA.py:78
            model_simulate<D>(a, tm, eps, ym);                          // GLMALA.py:82 (same noise)
            glabc_fxs_add(&ap, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, yp) - c_p));
            glabc_fxs_add(&am, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, ym) - c_m));
        };
        if constexpr (D == 2) {
            // a Philox block holds the four normals of TWO consecutive simulations: lanes walk blocks, not simulations
            const int64_t first = (int64_t)k * num;                     // global simulation index of s = 0
            const int64_t b_lo = first >> 1, b_hi = (first + num - 1) >> 1;
            const int n_blocks = (int)(b_hi - b_lo + 1);
            const int brounds = (n_blocks + G - 1) / G;
#pragma unroll 1
            for (int rd = 0; rd < brounds; ++rd) {
                const int bi = rd * G + sub;
                if (bi < n_blocks) {
                    const int64_t b = b_lo + bi;
                    glabc_u32x4 blk = glabc_philox4x32_10(r2.c0, r2.c1, step, GRAD_BASE + (uint32_t)g * GRAD_STRIDE + (uint32_t)b,
                                                          r2.k0, r2.k1);
                    float e0[2], e1[2];
                    glabc_normal_pair(blk.v[0], blk.v[1], &e0[0], &e0[1]);
                    glabc_normal_pair(blk.v[2], blk.v[3], &e1[0], &e1[1]);
                    const int64_t s0 = 2 * b - first, s1 = s0 + 1;      // simulation indices within coordinate k
                    if (s0 >= 0 && s0 < num) one_sim(e0);
                    if (s1 >= 0 && s1 < num) one_sim(e1);
                }
            }
        } else {
#pragma unroll 1
            for (int rd = 0; rd < rounds; ++rd) {
                const int s = rd * G + sub;
                if (s < num) {
                    float eps[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const int64_t nn = ((int64_t)k * num + s) * D + j;
                        glabc_u32x4 blk = glabc_philox4x32_10(r2.c0, r2.c1, step,
                                                              GRAD_BASE + (uint32_t)g * GRAD_STRIDE + (uint32_t)(nn >> 2), r2.k0, r2.k1);
                        const int p = (int)((nn & 3) >> 1);
                        float z0, z1;
                        glabc_normal_pair(p ? blk.v[2] : blk.v[0], p ? blk.v[3] : blk.v[1], &z0, &z1);
                        eps[j] = (nn & 1) ? z1 : z0;
                    }
                    one_sim(eps);
                }
            }
        }
        for (int mm = G >> 1; mm >= 1; mm >>= 1) {                      // merge the group's partial sums (exact integers)
            glabc_fxsplit bp, bm;
            bp.s1 = (int64_t)shfl_xor_u64((uint64_t)ap.s1, mm);
            bp.a = shfl_xor_u64(ap.a, mm);
            bp.b = (int64_t)shfl_xor_u64((uint64_t)ap.b, mm);
            bp.c = shfl_xor_u64(ap.c, mm);
            bm.s1 = (int64_t)shfl_xor_u64((uint64_t)am.s1, mm);
            bm.a = shfl_xor_u64(am.a, mm);
            bm.b = (int64_t)shfl_xor_u64((uint64_t)am.b, mm);
            bm.c = shfl_xor_u64(am.c, mm);
            glabc_fxs_merge(&ap, &bp);
            glabc_fxs_merge(&am, &bm);
        }
        const glabc_fxsum fp = glabc_fxs_finish(&ap), fm = glabc_fxs_finish(&am);
        const double nd = (double)num;
        const double s1p = glabc_fx_sum1(&fp), s2p = glabc_fx_sum2(&fp), s1m = glabc_fx_sum1(&fm), s2m = glabc_fx_sum2(&fm);
        const double mu_p = c_p + s1p / nd, mu_m = c_m + s1m / nd;                           // GLMALA.py:86-87
        const double var_p = (s2p - (s1p * s1p) / nd) / (nd - 1.0), var_m = (s2m - (s1m * s1m) / nd) / (nd - 1.0);   // :88-89
        const double lp = (-0.5 * glabc_log(var_p + m.eps_sq)) - ((0.5 * (mu_p * mu_p)) / (var_p + m.eps_sq));   // :90-91
        const double lm = (-0.5 * glabc_log(var_m + m.eps_sq)) - ((0.5 * (mu_m * mu_m)) / (var_m + m.eps_sq));   // :92-93
        const double gll = (lp - lm) / 0.2;                                                  // :94
#pragma unroll
        for (int j = 0; j < D; ++j) {
            tp[j] = th[j] + (j == k ? hp : 0.0f);                                            // :84
            tm[j] = th[j] - (j == k ? hp : 0.0f);
        }
        const float gp = (dist_log_prob<D>(a.prior, tp) - dist_log_prob<D>(a.prior, tm)) / 0.00002f;   // :84-85
        const double gk = gll + (double)gp;                                                  // :95
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j == k) gout[j] = gk;
    }
    // back to the owner: chain of rank r was computed by group r, whose first lane is r*G
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double v = shfl_f64(gout[j], rank * G);
        if (need) grad[j] = v;
    }
}

// the MALA local move, GLMALA.py:182-200.  Every lane calls it (the gradients are wave-cooperative);
// `is_local` marks the chains that take the move this iteration.
template <int D, bool LEAN = false>
GLABC_DEV bool mala_move(const MalaArgs<D>& m, const Rng& rng, uint32_t step, bool is_local, float u_accept,
                         const float (&zn)[2 * D], MalaChain<D>& c, GradShared* sh)
{
    const StepArgs<D>& a = m.s;
    float thf[D];
#pragma unroll
    for (int j = 0; j < D; ++j) thf[j] = (float)c.theta[j];
    const bool init = is_local && !(c.flags & GLABC_FLAG_HAS_GRAD);                          // :183-184
    coop_gradient<D, LEAN>(m, rng, step, 0, init, thf, c.grad, sh);
    if (init) c.flags |= GLABC_FLAG_HAS_GRAD;
    // Local_proposal_forward, :25-44
    float t[D];
    double x[D];
    const float tauf = (float)m.tau;
#pragma unroll
    for (int j = 0; j < D; ++j) t[j] = 0.0f + 0.5f * (zn[j] * zn[j]);
    const float log_pro = (float)(-0.5 * (double)D * GLABC_LOG_2PI) - aten_rowsum<D>(t);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const float z = 0.0f + 1.0f * zn[j];
        const float av = z * tauf;
        const double b = (c.flags & GLABC_FLAG_TH64) ? ((double)av + c.theta[j]) : (double)(av + (float)c.theta[j]);
        x[j] = b + (c.grad[j] * m.tau_sq) / 2.0;                                             // :43
    }
    double gprop[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        thf[j] = (float)x[j];                                                                // :62
        gprop[j] = 0.0;
    }
    coop_gradient<D, LEAN>(m, rng, step, 1, is_local, thf, gprop, sh);                         // :187
    if (!is_local) return false;
    double y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const float noise = a.noise_loc[j] + a.noise_scale[j] * zn[D + j];
        y[j] = __builtin_fabs(x[j]) + (double)noise;                                         // :188-189
    }
    double tq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double arg = ((c.theta[j] - x[j]) - (gprop[j] * m.tau_sq) / 2.0) / m.tau;      // :115
        const double e = (arg - 0.0) / 1.0;
        tq[j] = 0.0 + 0.5 * (e * e);
    }
    const double lq_rev = (-0.5 * (double)D * GLABC_LOG_2PI) - aten_rowsum_f64<D>(tq);
    double log_acc = dist_log_prob_f64<D>(a.prior, x) + model_log_kernel_f64<D>(a, y);       // :190
    log_acc = log_acc + lq_rev;                                                              // :191
    log_acc = log_acc - state_prior<D>(a, c);                                                // :192
    log_acc = log_acc - state_kernel<D>(a, c);
    log_acc = log_acc - (double)log_pro;                                                     // :193
    const double log_u = (double)glabc_logf(u_accept);                                       // :194
    if (log_u < log_acc) {                                                                   // :195-199
#pragma unroll
        for (int j = 0; j < D; ++j) {
            c.theta[j] = x[j];
            c.y[j] = y[j];
            c.grad[j] = gprop[j];
        }
        c.flags |= GLABC_FLAG_TH64;
        return true;                         // `local` is NOT set: GLMALA.py:195-199 vs GLMCMC.py:100
    }
    return false;
}

// Philox blocks of candidate j: D proposal draws, then D simulator draws from the next even word (same layout as GLMCMC,
// glabc_device.h chain_step)
template <int D>
GLABC_DEV void candidate_draws(const Rng& rng, uint32_t step, int j, bool uniform_prop, float (&e)[D], float (&s)[D])
{
    constexpr int DP = D + (D & 1);
    constexpr int SPP = (DP + D + 3) / 4;
    uint32_t w[4 * SPP];
#pragma unroll
    for (int b = 0; b < SPP; ++b) {
        glabc_u32x4 o = glabc_philox4x32_10(rng.c0, rng.c1, step, (uint32_t)(1 + j * SPP + b), rng.k0, rng.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[4 * b + q] = o.v[q];
    }
    float nrm[4 * SPP];
#pragma unroll
    for (int i = 0; 2 * i < DP + D; ++i) glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        e[i] = uniform_prop ? glabc_uniform_f32(w[i]) : nrm[i];
        s[i] = nrm[DP + i];
    }
}

// the iSIR global move of GLMALA.py:151-180
template <int D, int N>
GLABC_DEV bool mala_isir_move(const MalaArgs<D>& m, const Rng& rng, uint32_t step, double u_res, MalaChain<D>& c)
{
    const StepArgs<D>& a = m.s;
    if (c.flags & GLABC_FLAG_LOCAL) {                                                        // :152-156
        if (c.flags & GLABC_FLAG_TH64) {
            c.log_w = (state_prior<D>(a, c) + state_kernel<D>(a, c)) - dist_log_prob_f64<D>(a.global, c.theta);
            c.flags |= GLABC_FLAG_LW64;
        } else {
            float thf[D];
#pragma unroll
            for (int j = 0; j < D; ++j) thf[j] = (float)c.theta[j];
            const float q = dist_log_prob<D>(a.global, thf);
            c.log_w = (double)(((float)state_prior<D>(a, c) + (float)state_kernel<D>(a, c)) - q);
        }
    }
    c.flags &= ~GLABC_FLAG_LOCAL;                                                            // :157
    float th[N][D], yy[N][D], lw0[N];
    const bool uni = a.global.kind == GLABC_DIST_UNIFORM;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float e[D], s[D];
        candidate_draws<D>(rng, step, j, uni, e, s);
#pragma unroll
        for (int q = 0; q < D; ++q) th[j][q] = a.global.p0[q] + a.global.p2[q] * e[q];       // :158
        const float lq = dist_forward_log_p<D>(a.global, e);
        model_simulate<D>(a, th[j], s, yy[j]);                                               // :163
        lw0[j] = (dist_log_prob<D>(a.prior, th[j]) + model_log_kernel<D>(a, yy[j])) - lq;    // :164-165
    }
    int ind = -1;
    if (c.flags & GLABC_FLAG_LW64) {
        double w[N + 1];
        w[0] = glabc_exp(c.log_w);
#pragma unroll
        for (int j = 0; j < N; ++j) w[j + 1] = glabc_exp((double)lw0[j]);                    // :169
#pragma unroll
        for (int k = 0; k <= N; ++k) w[k] = (w[k] != w[k]) ? 0.0 : w[k];                     // :171-172
        const double tot = aten_rowsum_f64<N + 1>(w);                                        // :173
        double run = 0.0;
#pragma unroll
        for (int k = 0; k <= N; ++k) {
            run += w[k] / tot;
            ind = (ind < 0 && u_res < run) ? k : ind;                                        // :174
        }
    } else {
        float w[N + 1];
        w[0] = glabc_expf((float)c.log_w);
#pragma unroll
        for (int j = 0; j < N; ++j) w[j + 1] = glabc_expf(lw0[j]);
#pragma unroll
        for (int k = 0; k <= N; ++k) w[k] = (w[k] != w[k]) ? 0.0f : w[k];
        const float tot = aten_rowsum<N + 1>(w);
        double run = 0.0;
#pragma unroll
        for (int k = 0; k <= N; ++k) {
            run += (double)(w[k] / tot);
            ind = (ind < 0 && u_res < run) ? k : ind;
        }
    }
    const bool moved = ind > 0;                                                              // :175-179
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (ind == j + 1) {
#pragma unroll
            for (int q = 0; q < D; ++q) {
                c.theta[q] = (double)th[j][q];
                c.y[q] = (double)yy[j][q];
            }
            c.log_w = (double)lw0[j];
        }
    }
    return moved;
}

// At most 256 registers (amdgpu_waves_per_eu): launches of more than one wavefront per SIMD then run two per SIMD.
template <int D, int N, bool LEAN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) glmala_kernel(const MalaArgs<D> m)
{
    const StepArgs<D>& a = m.s;
    __shared__ GradShared gsh;
    const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool valid = tid < a.n_chains;
    const int64_t i = valid ? tid : a.n_chains - 1;          // tail lanes shadow the last chain (no stores): the cooperative
                                                             // gradient needs every lane of the wavefront alive
    MalaChain<D> c;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        c.theta[j] = m.theta64[j * a.stride + i];
        c.y[j] = m.y64[j * a.stride + i];
        c.grad[j] = m.grad[j * a.stride + i];
    }
    c.log_w = m.log_w64[i];
    c.flags = a.flags[i];
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;

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
        float prev[D], cur[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = (float)c.theta[j];

        glabc_u32x4 h = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
        const float u_branch = glabc_uniform_f32(h.v[0]);
        const bool is_global = u_branch < a.gf;                                              // GLMALA.py:151
        bool moved = false;
        if (is_global) moved = mala_isir_move<D, N>(m, rng, step, glabc_uniform_f64(h.v[2], h.v[3]), c);
        {
            float e[D], s[D], zn[2 * D];
            candidate_draws<D>(rng, step, 0, false, e, s);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                zn[j] = e[j];
                zn[D + j] = s[j];
            }
            const bool mv = mala_move<D, LEAN>(m, rng, step, !is_global && valid, glabc_uniform_f32(h.v[1]), zn, c, &gsh);
            moved = is_global ? moved : mv;
        }
        c.n_moves += moved ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < D; ++j) cur[j] = (float)c.theta[j];                              // Theta_Re is float32, :148,180,200

        if (hist && valid) {
#pragma unroll
            for (int j = 0; j < D; ++j) hist[((int64_t)t * D + j) * a.hist_stride] = cur[j];
        }
        if (mom) {
            int k = 0;
#pragma unroll
            for (int p = 0; p < D; ++p) {
                s1[p] += (double)cur[p];
#pragma unroll
                for (int q = p; q < D; ++q, ++k) {
                    s2[k] += (double)cur[p] * (double)cur[q];
                    double dp = (double)cur[p] - (double)prev[p];
                    double dq = (double)cur[q] - (double)prev[q];
                    sj[k] += dp * dq;
                }
            }
        }
    }

    if (!valid) return;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        m.theta64[j * a.stride + i] = c.theta[j];
        m.y64[j * a.stride + i] = c.y[j];
        m.grad[j * a.stride + i] = c.grad[j];
        a.theta[j * a.stride + i] = (float)c.theta[j];
        a.y[j * a.stride + i] = (float)c.y[j];
    }
    m.log_w64[i] = c.log_w;
    if (a.log_w) a.log_w[i] = (float)c.log_w;
    a.flags[i] = c.flags;
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

// ---- theta_dim 2: a TEAM of NW wavefronts per 64 chains ---------------------------------------------------------------------
//
// 65 536 chains are 1024 wavefronts of glmala_kernel: ONE per SIMD, which issues a vector instruction at best every other
// slot and has nothing to cover its LDS atomics with (profiles/r02_pmc_glmala.json: 73 % VALU-active at 4.1 cycles per
// instruction).  Halving the chains per wavefront does not help: ~40 % of an iteration's instructions are per wavefront, not
// per chain (the iSIR move, the head draws, the end of a gradient) and would be paid twice (measured: 56.3 against 53.1 ms).
// The team keeps them paid ONCE: wavefront 0 (the main wavefront) owns the 64 chains -- state in its registers, head draws,
// iSIR move, MALA proposal, accept, Theta_Re row, sums -- and wavefronts 1..NW-1 only help with the one part that is dealt
// flat over lanes anyway, the 2 * theta_dim * num_grad simulations of the finite-difference gradients.  While the helpers
// work through the gradient items the main wavefront runs the iSIR move of the chains on the global branch (they are other
// chains than the ones waiting for a gradient) and then takes a correspondingly smaller share of the items (`credit`).
// 1024 workgroups x NW wavefronts = NW per SIMD.  Per iteration: the main wavefront publishes the chains that need a gradient
// (theta, Philox counter) in LDS -> barrier -> every lane of the team walks its contiguous range of (chain, coordinate, Philox
// block) items and adds its partial sums to the (chain, coordinate, side) accumulators in LDS (64-bit integer atomics: exact,
// order-free) -> barrier -> the main wavefront finishes the statistics, four lanes per chain (coordinate x side), and the
// owners take the accept decision.  A chain's first local move needs the gradient at its current state first (GLMALA.py:
// 183-184): such an iteration runs the publish / items / finish sequence twice (round 0, then round 1).
// Geometry only: every draw is a function of (seed, chain id, iteration, slot), the sums are exact integers, every float
// operation is the one glmala_kernel performs -- tests/test_hip_parity.py holds both kernels to the checker bit for bit.
struct TeamShared {
    unsigned long long acc[64][2][2][4];   // [rank][coordinate][side +,-][s1, a, b, c] (glabc_fxsplit)
    float th[64][2];                       // theta of the chain of that rank
    uint32_t c0[64], c1[64];               // its Philox counter words (global chain id)
    int n;                                 // chains published for this round
    int round;                             // 0: gradients at the current states (GLMALA.py:183-184), 1: at the proposals (:187)
};

// this lane's share of the n * (blocks of coordinate 0 + blocks of coordinate 1) items of a round
template <bool LEAN, int NW>
GLABC_DEV void team_items(const MalaArgs<2>& m, uint32_t key0, uint32_t key1, uint32_t step, int g, int n, int wave, int lane,
                          int credit, TeamShared* sh)
{
    constexpr int D = 2;
    const StepArgs<D>& a = m.s;
    const int num = m.num_grad;
    const float h = 0.1f;
    // a Philox block = the four normals of two consecutive simulations; coordinate k owns simulations k num .. k num + num - 1
    const int nb0 = ((num - 1) >> 1) + 1;
    const int blo1 = num >> 1, nb1 = ((2 * num - 1) >> 1) - blo1 + 1;
    const int per = nb0 + nb1, total = n * per;
    int q_m, q_h = 0;
    if constexpr (NW == 1) {
        q_m = (total + 63) >> 6;
    } else {
        constexpr int LH = 64 * (NW - 1);
        int num_h = total + 64 * credit;                                // credit < 0: the main wavefront takes MORE than a helper
        num_h = num_h > 0 ? num_h : 0;
        q_h = (num_h + 64 * NW - 1) / (64 * NW);
        q_m = q_h - credit;                                             // 64 q_m + LH q_h >= total
        if (q_m < 0) {
            q_m = 0;
            q_h = (total + LH - 1) / LH;
        }
    }
    int it0 = (wave == 0) ? lane * q_m : 64 * q_m + ((wave - 1) * 64 + lane) * q_h;
    int it1 = it0 + ((wave == 0) ? q_m : q_h);
    it1 = it1 < total ? it1 : total;
    if (it0 >= it1) return;
    int cur = it0 / per;
    int r = it0 - cur * per;
    int k = (r >= nb0) ? 1 : 0;
    int bi = k ? r - nb0 : r;
    float tp[D], tm[D];
    double c_p = 0.0, c_m = 0.0;
    uint32_t cc0 = 0u, cc1 = 0u;
    glabc_fxsplit ap = {0, 0, 0, 0}, am = {0, 0, 0, 0};
    auto enter = [&]() {                                                // the constants of (chain cur, coordinate k)
        float zero[D], y0p[D], y0m[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float th = sh->th[cur][j];
            tp[j] = th + (j == k ? h : 0.0f);                           // GLMALA.py:67
            tm[j] = th - (j == k ? h : 0.0f);                           // GLMALA.py:68
            zero[j] = 0.0f;
        }
        model_simulate<D>(a, tp, zero, y0p);                            // centres: the noise-free discrepancies
        model_simulate<D>(a, tm, zero, y0m);
        c_p = (double)model_discrepancy<D, LEAN>(a, y0p);
        c_m = (double)model_discrepancy<D, LEAN>(a, y0m);
        cc0 = sh->c0[cur];
        cc1 = sh->c1[cur];
    };
    auto flush = [&]() {
        unsigned long long* dst = &sh->acc[cur][k][0][0];
        atomicAdd(dst + 0, (unsigned long long)ap.s1);
        atomicAdd(dst + 1, ap.a);
        atomicAdd(dst + 2, (unsigned long long)ap.b);
        atomicAdd(dst + 3, ap.c);
        atomicAdd(dst + 4, (unsigned long long)am.s1);
        atomicAdd(dst + 5, am.a);
        atomicAdd(dst + 6, (unsigned long long)am.b);
        atomicAdd(dst + 7, am.c);
        ap = glabc_fxsplit{0, 0, 0, 0};
        am = glabc_fxsplit{0, 0, 0, 0};
    };
    auto one_sim = [&](const float (&eps)[D]) {
        float yp[D], ym[D];
        model_simulate<D>(a, tp, eps, yp);                              // GLMALA.py:78
        model_simulate<D>(a, tm, eps, ym);                              // GLMALA.py:82 (same noise)
        glabc_fxs_add(&ap, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, yp) - c_p));
        glabc_fxs_add(&am, glabc_fx_quantize((double)model_discrepancy<D, LEAN>(a, ym) - c_m));
    };
    enter();
#pragma unroll 1
    for (int it = it0; it < it1; ++it) {
        const int b = (k ? blo1 : 0) + bi;
        glabc_u32x4 blk = glabc_philox4x32_10(cc0, cc1, step, GRAD_BASE + (uint32_t)g * GRAD_STRIDE + (uint32_t)b, key0, key1);
        float e0[2], e1[2];
        glabc_normal_pair(blk.v[0], blk.v[1], &e0[0], &e0[1]);
        glabc_normal_pair(blk.v[2], blk.v[3], &e1[0], &e1[1]);
        const int s0 = 2 * b - k * num, s1 = s0 + 1;                    // simulation indices within coordinate k
        if (s0 >= 0 && s0 < num) one_sim(e0);
        if (s1 >= 0 && s1 < num) one_sim(e1);
        if (++bi == (k ? nb1 : nb0) && it + 1 < it1) {                  // the range crosses into the next (chain, coordinate)
            flush();
            bi = 0;
            cur += k;
            k ^= 1;
            enter();
        }
    }
    flush();
}

// End of a round, main wavefront (all 64 lanes call): the statistics of GLMALA.py:86-94 for the n published chains, one lane per
// (chain, coordinate, side), then the owners (`need`, rank among the set bits of `mask`) assemble their gradient.  The
// accumulators are left zero for the next round.
template <bool LEAN>
GLABC_DEV void team_finish(const MalaArgs<2>& m, bool need, unsigned long long mask, const float (&theta)[2], double (&grad)[2],
                           TeamShared* sh)
{
    constexpr int D = 2;
    const StepArgs<D>& a = m.s;
    const int lane = (int)(threadIdx.x & 63u);
    const int n = __popcll(mask);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    const float h = 0.1f, hp = 0.00001f;
    const double nd = (double)m.num_grad;
    double l4[4] = {0.0, 0.0, 0.0, 0.0};
    const int passes = (4 * n + 63) >> 6;
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
        const int job = p * 64 + lane;
        const int r = job >> 2, k = (job >> 1) & 1, side = job & 1;
        double l = 0.0;
        if (r < n) {
            unsigned long long* src = &sh->acc[r][k][side][0];
            glabc_fxsplit sp;
            sp.s1 = (int64_t)src[0]; sp.a = src[1]; sp.b = (int64_t)src[2]; sp.c = src[3];
            src[0] = 0ull; src[1] = 0ull; src[2] = 0ull; src[3] = 0ull;
            float o[D], zero[D], y0[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float th = sh->th[r][j];
                const float hh = (j == k) ? h : 0.0f;
                o[j] = side ? th - hh : th + hh;                                             // GLMALA.py:67-68
                zero[j] = 0.0f;
            }
            model_simulate<D>(a, o, zero, y0);
            const double centre = (double)model_discrepancy<D, LEAN>(a, y0);
            const glabc_fxsum f = glabc_fxs_finish(&sp);
            const double s1 = glabc_fx_sum1(&f), s2 = glabc_fx_sum2(&f);
            const double mu = centre + s1 / nd;                                              // GLMALA.py:86-87
            const double var = (s2 - (s1 * s1) / nd) / (nd - 1.0);                           // :88-89
            l = (-0.5 * glabc_log(var + m.eps_sq)) - ((0.5 * (mu * mu)) / (var + m.eps_sq)); // :90-93
        }
        const int src_lane = (rank & 15) << 2;
        const double v0 = shfl_f64(l, src_lane), v1 = shfl_f64(l, src_lane + 1), v2 = shfl_f64(l, src_lane + 2),
                     v3 = shfl_f64(l, src_lane + 3);
        if (need && (rank >> 4) == p) {
            l4[0] = v0; l4[1] = v1; l4[2] = v2; l4[3] = v3;
        }
    }
    if (need) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double gll = (l4[2 * k] - l4[2 * k + 1]) / 0.2;                            // :94
            float op[D], om[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                op[j] = theta[j] + (j == k ? hp : 0.0f);                                     // :84
                om[j] = theta[j] - (j == k ? hp : 0.0f);
            }
            const float gp = (dist_log_prob<D>(a.prior, op) - dist_log_prob<D>(a.prior, om)) / 0.00002f;   // :84-85
            grad[k] = gll + (double)gp;                                                      // :95
        }
    }
}

// main wavefront: hand the chains marked `need` (theta, Philox counter) to the team
GLABC_DEV void team_publish(bool need, unsigned long long mask, const float (&theta)[2], const Rng& rng, int round, TeamShared* sh)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (need) {
        sh->th[rank][0] = theta[0];
        sh->th[rank][1] = theta[1];
        sh->c0[rank] = rng.c0;
        sh->c1[rank] = rng.c1;
    }
    if (lane == 0) {
        sh->n = __popcll(mask);
        sh->round = round;
    }
}

template <int N, bool LEAN, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) glmala_team_kernel(const MalaArgs<2> m)
{
    constexpr int D = 2;
    const StepArgs<D>& a = m.s;
    __shared__ TeamShared sh;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    {                                                                   // accumulators start at zero (team_finish keeps them so)
        unsigned long long* z = &sh.acc[0][0][0][0];
        for (int q = (int)threadIdx.x; q < 64 * 16; q += 64 * NW) z[q] = 0ull;
    }
    if (wave != 0) {                                                    // ---- helper wavefronts: gradient items only ----
#pragma unroll 1
        for (int t = 0; t < a.n_steps; ++t) {
            const uint32_t step = a.step0 + (uint32_t)t;
            __syncthreads();                                            // a round is published
            if (sh.round == 0) {
                team_items<LEAN, NW>(m, a.seed_lo, a.seed_hi, step, 0, sh.n, wave, lane, 0, &sh);
                __syncthreads();                                        // round 0 summed
                __syncthreads();                                        // round 1 published
            }
            team_items<LEAN, NW>(m, a.seed_lo, a.seed_hi, step, 1, sh.n, wave, lane, m.credit, &sh);
            __syncthreads();                                            // round 1 summed
        }
        return;
    }
    // ---- main wavefront: the 64 chains ----
    if (m.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (m.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (m.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int64_t tid = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = tid < a.n_chains;
    const int64_t i = valid ? tid : a.n_chains - 1;          // tail lanes shadow the last chain (no stores)
    MalaChain<D> c;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        c.theta[j] = m.theta64[j * a.stride + i];
        c.y[j] = m.y64[j * a.stride + i];
        c.grad[j] = m.grad[j * a.stride + i];
    }
    c.log_w = m.log_w64[i];
    c.flags = a.flags[i];
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;

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
    const float tauf = (float)m.tau;

#pragma unroll 1
    for (int t = 0; t < a.n_steps; ++t) {
        const uint32_t step = a.step0 + (uint32_t)t;
        float prev[D], cur[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = (float)c.theta[j];

        glabc_u32x4 hd = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
        const bool is_global = glabc_uniform_f32(hd.v[0]) < a.gf;                            // GLMALA.py:151
        const bool is_local = !is_global && valid;
        float e[D], sn[D];
        candidate_draws<D>(rng, step, 0, false, e, sn);                                      // the MALA move's z and simulator noise

        // ---- round 0: grad_logABC_Theta_old of a chain's first local move, GLMALA.py:183-184 ----
        const bool init = is_local && !(c.flags & GLABC_FLAG_HAS_GRAD);
        const unsigned long long mask0 = __ballot(init);
        float thf[D];
        if (mask0 != 0ull) {                                                                 // wave-uniform
#pragma unroll
            for (int j = 0; j < D; ++j) thf[j] = (float)c.theta[j];                          // :62
            team_publish(init, mask0, thf, rng, 0, &sh);
            __syncthreads();
            team_items<LEAN, NW>(m, rng.k0, rng.k1, step, 0, __popcll(mask0), 0, lane, 0, &sh);
            __syncthreads();
            team_finish<LEAN>(m, init, mask0, thf, c.grad, &sh);
            if (init) c.flags |= GLABC_FLAG_HAS_GRAD;
        }
        // ---- Local_proposal_forward, GLMALA.py:25-44 ----
        float tt[D];
        double x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) tt[j] = 0.0f + 0.5f * (e[j] * e[j]);
        const float log_pro = (float)(-0.5 * (double)D * GLABC_LOG_2PI) - aten_rowsum<D>(tt);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float z = 0.0f + 1.0f * e[j];
            const float av = z * tauf;
            const double b = (c.flags & GLABC_FLAG_TH64) ? ((double)av + c.theta[j]) : (double)(av + (float)c.theta[j]);
            x[j] = b + (c.grad[j] * m.tau_sq) / 2.0;                                         // :43
            thf[j] = (float)x[j];                                                            // :62
        }
        // ---- round 1: the gradient at the proposal (:187) by the team, the iSIR move of the other chains meanwhile ----
        const unsigned long long mask1 = __ballot(is_local);
        team_publish(is_local, mask1, thf, rng, 1, &sh);
        __syncthreads();
        bool moved = false;
        if (is_global) moved = mala_isir_move<D, N>(m, rng, step, glabc_uniform_f64(hd.v[2], hd.v[3]), c);
        team_items<LEAN, NW>(m, rng.k0, rng.k1, step, 1, __popcll(mask1), 0, lane, m.credit, &sh);
        __syncthreads();
        double gprop[D] = {0.0, 0.0};
        if (mask1 != 0ull) team_finish<LEAN>(m, is_local, mask1, thf, gprop, &sh);
        if (is_local) {                                                                      // GLMALA.py:188-199
            double y[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float noise = a.noise_loc[j] + a.noise_scale[j] * sn[j];
                y[j] = __builtin_fabs(x[j]) + (double)noise;                                 // :188-189
            }
            double tq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double arg = ((c.theta[j] - x[j]) - (gprop[j] * m.tau_sq) / 2.0) / m.tau;   // :115
                const double ee = (arg - 0.0) / 1.0;
                tq[j] = 0.0 + 0.5 * (ee * ee);
            }
            const double lq_rev = (-0.5 * (double)D * GLABC_LOG_2PI) - aten_rowsum_f64<D>(tq);
            double log_acc = dist_log_prob_f64<D>(a.prior, x) + model_log_kernel_f64<D>(a, y);    // :190
            log_acc = log_acc + lq_rev;                                                      // :191
            log_acc = log_acc - state_prior<D>(a, c);                                        // :192
            log_acc = log_acc - state_kernel<D>(a, c);
            log_acc = log_acc - (double)log_pro;                                             // :193
            const double log_u = (double)glabc_logf(glabc_uniform_f32(hd.v[1]));             // :194
            if (log_u < log_acc) {                                                           // :195-199
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    c.theta[j] = x[j];
                    c.y[j] = y[j];
                    c.grad[j] = gprop[j];
                }
                c.flags |= GLABC_FLAG_TH64;
                moved = true;                    // `local` is NOT set: GLMALA.py:195-199 vs GLMCMC.py:100
            }
        }
        c.n_moves += moved ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < D; ++j) cur[j] = (float)c.theta[j];                              // Theta_Re is float32, :148,180,200
        if (hist && valid) {
#pragma unroll
            for (int j = 0; j < D; ++j) hist[((int64_t)t * D + j) * a.hist_stride] = cur[j];
        }
        if (mom) {
            int k = 0;
#pragma unroll
            for (int p = 0; p < D; ++p) {
                s1[p] += (double)cur[p];
#pragma unroll
                for (int q = p; q < D; ++q, ++k) {
                    s2[k] += (double)cur[p] * (double)cur[q];
                    double dp = (double)cur[p] - (double)prev[p];
                    double dq = (double)cur[q] - (double)prev[q];
                    sj[k] += dp * dq;
                }
            }
        }
    }

    if (!valid) return;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        m.theta64[j * a.stride + i] = c.theta[j];
        m.y64[j * a.stride + i] = c.y[j];
        m.grad[j * a.stride + i] = c.grad[j];
        a.theta[j * a.stride + i] = (float)c.theta[j];
        a.y[j * a.stride + i] = (float)c.y[j];
    }
    m.log_w64[i] = c.log_w;
    if (a.log_w) a.log_w[i] = (float)c.log_w;
    a.flags[i] = c.flags;
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

template <int D>
__global__ void __launch_bounds__(64) glmala_init_kernel(const MalaArgs<D> m)
{
    const StepArgs<D>& a = m.s;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_chains) return;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        m.theta64[j * a.stride + i] = (double)a.theta[j * a.stride + i];
        m.y64[j * a.stride + i] = (double)a.y[j * a.stride + i];
        m.grad[j * a.stride + i] = 0.0;
    }
    m.log_w64[i] = 0.0;
    a.flags[i] = GLABC_FLAG_LOCAL;                                                           // GLMALA.py:146-147
}

// host-side launcher of one theta_dim; defined in glabc_mala_dim.hip (one TU per D)
template <int D>
int launch_glmala_dim(int n_batch, const MalaArgs<D>& m, hipStream_t stream);
template <int D>
int launch_glmala_init_dim(const MalaArgs<D>& m, hipStream_t stream);

}  // namespace glabc
