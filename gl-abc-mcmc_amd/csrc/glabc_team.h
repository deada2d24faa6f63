// glabc_team.h -- GLMCMC (GLMCMC.py:58-104) with a TEAM of two to four wavefronts per 64 chains.
//
// 65 536 chains are 1024 wavefronts of sampler_kernel: ONE per SIMD, and a lone wavefront issues a vector instruction at
// best every other slot (MI355X_MICROARCH.md, "vector-instruction ISSUE cost") -- the launch runs at 0.86 of that one-wave
// limit, i.e. at 0.43 of what the vector unit can do.  Two lanes per chain inside a wavefront double the
// wavefronts but pay 1.31 x the instructions, because both lanes keep the chain's state and take the decision.  The team
// splits the iteration where nothing has to be duplicated:
//
//   wavefronts 1 .. NW-1 (helpers): candidates NA .. N-1 of every chain, split evenly.  An iSIR candidate is a pure function of (seed, chain id,
//       iteration, j): theta' ~ importance proposal, y' = simulate(theta'), prior', K', log q' and the weight
//       exp((prior' + K') - q')  (GLMCMC.py:66-81).  The helper holds no chain state at all; it leaves the candidates in LDS.
//   wavefront 0 (main): the chain state in registers, the step head (branch / accept / resampling draws), candidates
//       0 .. NA-1 -- candidate 0 doubles as the local move exactly as in chain_step -- and then the decision: weights of the
//       helper's candidates from LDS, torch.sum order, double-precision index, MH test, state update, Theta_Re row, sums.
//
// The helpers work one iteration AHEAD into a second LDS buffer, so the wavefronts meet at ONE barrier per iteration and
// none waits for another's arithmetic: 1024 workgroups x NW wavefronts = NW per SIMD with (almost) the instruction count
// of one work-item per chain.  The main wavefront runs at s_setprio 1: it carries the serial part.
//
// Geometry only.  Philox slots, words, every float operation and their order are chain_step's (glabc_device.h); the tests
// hold this kernel, sampler_kernel and the CPU checker to the same bits (tests/test_hip_parity.py, tests/fuzz_parity.py).
#pragma once

#include "glabc_device.h"

namespace glabc {

// Split of an iteration's N candidates over the NW wavefronts of a team.  The main wavefront (candidate 0 included) also draws
// the step head and takes the decision, which cost it about 1.1 candidates (85 + 165 of 227 vector instructions); the helpers
// share the rest evenly.  N = 5: two wavefronts 2 | 3, three wavefronts 1 | 2 2, four 1 | 2 1 1.
constexpr int team_main_candidates(int n, int nw)
{
    const int t10 = (10 * n + 11) / nw - 11;               // ten times ((n + 1.1) / nw - 1.1)
    const int na = (t10 + 5) / 10;                         // rounded
    return na < 1 ? 1 : (na > n - (nw - 1) ? n - (nw - 1) : na);
}
// first candidate of helper h (h = 0 .. nw-2; h = nw-1 gives n)
constexpr int team_helper_first(int n, int nw, int h)
{
    const int na = team_main_candidates(n, nw), nh = n - na, per = nh / (nw - 1), extra = nh % (nw - 1);
    return na + h * per + (h < extra ? h : extra);
}
constexpr bool team_split_ok(int n, int nw) { return n >= nw && team_main_candidates(n, nw) >= 1; }

template <int D, int YD, int NH>
struct TeamCand {                          // one iteration's candidates of the helper, [field][slot][lane]: conflict-free
    float wl[NH][64];                      // exp(lw), NaN -> 0   GLMCMC.py:78-81
    float lw[NH][64];                      // (prior' + K') - q'  GLMCMC.py:74
    float pr[NH][64];
    float kk[NH][64];
    float th[NH][D][64];
    float yy[NH][YD][64];
};

// candidate j of (chain, step): chain_step's slot body.  FIRST = candidate 0, which is the local move's proposal when `loc`
// FAST = GLABC_MATH_FAST (include/glabc.h): hardware transcendentals; the draws used are recorded when a.dump_z is set (`row` =
// chain * n_steps + iteration of this launch)
template <int D, int YD, int VAR, bool FIRST, bool FAST = false>
GLABC_DEV void team_candidate(const StepArgs<D, YD>& a, const Rng& rng, uint32_t step, int j, bool loc, const Chain<D, YD>& c,
                              float (&th)[D], float (&yy)[YD], float& lw, float& pr, float& kk, float& wl, int64_t row = 0,
                              int n_prop = 0, bool dump = false)
{
    constexpr bool GU = (VAR == VAR_GAUSS_UNIT);
    constexpr bool GM = (VAR == VAR_GAMMA);               // GLABC_DIST_GAMMA as the importance proposal / the prior (glabc_device.h)
    constexpr int DP = D + (D & 1);
    constexpr int ND = NoiseDim<YD>::value;
    constexpr int M = DP + ND;
    constexpr int SPP = (M + 3) / 4;
    const bool g_uni = !GU && a.global.kind == GLABC_DIST_UNIFORM;
    const bool l_uni = !GU && a.local.kind == GLABC_DIST_UNIFORM;
    uint32_t w[4 * SPP];
#pragma unroll
    for (int b = 0; b < SPP; ++b) {
        glabc_u32x4 o = glabc_philox4x32_10(rng.c0, rng.c1, step, (uint32_t)(1 + j * SPP + b), rng.k0, rng.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[4 * b + q] = o.v[q];
    }
    const bool lc = FIRST && loc;
    const bool uni = lc ? l_uni : g_uni;
    float nrm[2 * ((M + 1) / 2)], e[D], s[ND];
#pragma unroll
    for (int i = 0; 2 * i < M; ++i) {
        if constexpr (FAST) fast_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
        else glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) e[i] = (!GU && uni) ? glabc_uniform_f32(w[i]) : nrm[i];
#pragma unroll
    for (int i = 0; i < ND; ++i) s[i] = nrm[DP + i];
    if constexpr (FAST) {
        if (dump) {                                                           // glabc_draws_out.z[chain][t][j][D + YD]
            float* z = a.dump_z + (row * n_prop + j) * (D + ND);
#pragma unroll
            for (int i = 0; i < D; ++i) z[i] = e[i];
#pragma unroll
            for (int i = 0; i < ND; ++i) z[D + i] = s[i];
        }
    }
#pragma unroll
    for (int q = 0; q < D; ++q) {
        const float p0 = lc ? a.local.p0[q] : a.global.p0[q];
        const float p2 = lc ? a.local.p2[q] : a.global.p2[q];
        const float t = p0 + p2 * e[q];                                       // distribution.py:170 / :77
        th[q] = lc ? (t + c.theta[q]) : t;                                    // GLMCMC.py:91
    }
    // a Gamma importance proposal (wave-uniform): the candidate and forward()'s log q from the chain's Gamma slots, exactly as
    // chain_step draws them; a lane on the local branch keeps candidate 0 as built above
    float lq_gamma = 0.0f;
    const bool g_gam = GM && a.global.kind == GLABC_DIST_GAMMA;
    if constexpr (GM) {
        if (g_gam) {
            float tg[D];
            dist_gamma_forward<D>(a.global, rng.c0, rng.c1, rng.k0, rng.k1, step, j, tg, lq_gamma);
#pragma unroll
            for (int q = 0; q < D; ++q) th[q] = lc ? th[q] : tg[q];
        }
    }
    float lq;
    if constexpr (GU) {
        float v[D];                                                           // both are c0 - sum 0.5 v^2 in the unit variant
#pragma unroll
        for (int q = 0; q < D; ++q) v[q] = lc ? (th[q] - a.global.p0[q]) : e[q];
        lq = dist_forward_log_p<D, GU>(a.global, v);
    } else {
        lq = lc ? dist_log_prob<D, GU, GM>(a.global, th) : (g_gam ? lq_gamma : dist_forward_log_p<D, GU>(a.global, e));
    }
    model_simulate<D, YD>(a, th, s, yy);
    pr = model_prior<D, YD, GU, GM>(a, th);
    kk = model_log_kernel<D, YD, GU, FAST>(a, yy);
    lw = (pr + kk) - lq;                                                      // GLMCMC.py:74
    const float v = FAST ? fast_expf(lw) : glabc_expf(lw);                    // GLMCMC.py:78
    wl = (v != v) ? 0.0f : v;                                                 // GLMCMC.py:80-81
}

// helper wavefront: candidates LO .. HI-1 of every iteration, one iteration ahead of the main wavefront
template <int D, int YD, int N, int VAR, int NA, int LO, int HI, bool FAST>
GLABC_DEV void team_helper(const StepArgs<D, YD>& a, const Rng& rng, int lane, TeamCand<D, YD, N - NA> (&buf)[2], int64_t chain, bool valid)
{
    Chain<D, YD> none;                                     // never read: FIRST = false
#pragma unroll 1
    for (int t = 0; t < a.n_steps; ++t) {
        const uint32_t step = a.step0 + (uint32_t)t;
        TeamCand<D, YD, N - NA>& o = buf[t & 1];
#pragma unroll
        for (int j = LO; j < HI; ++j) {
            float th[D], yy[YD], lw, pr, kk, wl;
            team_candidate<D, YD, VAR, false, FAST>(a, rng, step, j, false, none, th, yy, lw, pr, kk, wl,
                                                    chain * (int64_t)a.n_steps + t, N, FAST && valid && a.dump_z != nullptr);
            o.wl[j - NA][lane] = wl;
            o.lw[j - NA][lane] = lw;
            o.pr[j - NA][lane] = pr;
            o.kk[j - NA][lane] = kk;
#pragma unroll
            for (int q = 0; q < D; ++q) o.th[j - NA][q][lane] = th[q];
#pragma unroll
            for (int q = 0; q < YD; ++q) o.yy[j - NA][q][lane] = yy[q];
        }
        __syncthreads();                                   // iteration t's candidates are in LDS
    }
}

template <int D, int YD, int N, int VAR, int NW, bool FAST = false>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW))) team_sampler_kernel(const StepArgs<D, YD> a, int prio)
{
    constexpr bool GU = (VAR == VAR_GAUSS_UNIT);
    constexpr bool GM = (VAR == VAR_GAMMA);
    static_assert(NW >= 2 && NW <= 4 && team_split_ok(N, NW), "team of two to four wavefronts, at least one candidate each");
    constexpr int NA = team_main_candidates(N, NW), NH = N - NA;
    __shared__ TeamCand<D, YD, NH> buf[2];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int64_t tid = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = tid < a.n_chains;
    const int64_t i = valid ? tid : a.n_chains - 1;        // tail lanes shadow the last chain (no stores)
    const uint64_t gid = (uint64_t)(a.chain0 + i);
    Rng rng;
    rng.c0 = (uint32_t)gid;
    rng.c1 = (uint32_t)(gid >> 32);
    rng.k0 = a.seed_lo;
    rng.k1 = a.seed_hi;

    if (wave != 0) {                                       // ---- helpers: a pure function of (seed, chain id, iteration, j) ----
        if (wave == 1) team_helper<D, YD, N, VAR, NA, team_helper_first(N, NW, 0), team_helper_first(N, NW, 1), FAST>(a, rng, lane, buf, i, valid);
        if constexpr (NW >= 3) {
            if (wave == 2) team_helper<D, YD, N, VAR, NA, team_helper_first(N, NW, 1), team_helper_first(N, NW, 2), FAST>(a, rng, lane, buf, i, valid);
        }
        if constexpr (NW >= 4) {
            if (wave == 3) team_helper<D, YD, N, VAR, NA, team_helper_first(N, NW, 2), team_helper_first(N, NW, 3), FAST>(a, rng, lane, buf, i, valid);
        }
        return;
    }

    // ---- main: state, head, candidates 0 .. NA-1, decision ----
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    Chain<D, YD> c;
#pragma unroll
    for (int j = 0; j < D; ++j) c.theta[j] = a.theta[j * a.stride + i];
#pragma unroll
    for (int j = 0; j < YD; ++j) c.y[j] = a.y[j * a.stride + i];
    c.log_w = a.log_w[i];
    c.flags = a.flags[i];
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;
    c.gf = a.gf_chain ? a.gf_chain[i] : a.gf;
    refresh_cache<D, YD, GM>(a, c);
    if constexpr (FAST) c.kern = model_log_kernel<D, YD, false, true>(a, c.y);             // the cached K(y_old) in the arithmetic of the candidates
    c.lw_cur = (c.flags & GLABC_FLAG_LOCAL) ? (c.prior + c.kern) - c.q : c.log_w;          // GLMCMC.py:60-64
    {
        const float v = FAST ? fast_expf(c.lw_cur) : glabc_expf(c.lw_cur);
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
    float* hist = a.history ? a.history + i : nullptr;

#pragma unroll 1
    for (int t = 0; t < a.n_steps; ++t) {
        const uint32_t step = a.step0 + (uint32_t)t;
        float prev[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = c.theta[j];

        // step head, GLMCMC.py:59-65,98 (chain_step take_head)
        glabc_u32x4 hd = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
        const float ub = glabc_uniform_f32(hd.v[0]), ua = glabc_uniform_f32(hd.v[1]);
        const float log_u = FAST ? fast_logf(ua) : ((ua == 0.0f) ? -__builtin_inff() : glabc_logf_normal(ua));
        const bool is_global = ub < c.gf;
        const int64_t row = i * (int64_t)a.n_steps + t;
        const bool dump = FAST && valid && a.dump_z != nullptr;
        if constexpr (FAST) {
            if (dump) {                                                           // glabc_draws_out.u / .r
                a.dump_u[2 * row] = ub;
                a.dump_u[2 * row + 1] = ua;
                a.dump_r[row] = glabc_uniform_f64(hd.v[2], hd.v[3]);
            }
        }
        if (is_global) {
            if (c.flags & GLABC_FLAG_LOCAL) c.log_w = c.lw_cur;                   // GLMCMC.py:60-64
            c.flags &= ~GLABC_FLAG_LOCAL;                                         // GLMCMC.py:65
        }
        // own candidates
        float th[NA][D], yy[NA][YD], lw[NA], pr[NA], kk[NA], wl[NA];
        team_candidate<D, YD, VAR, true, FAST>(a, rng, step, 0, !is_global, c, th[0], yy[0], lw[0], pr[0], kk[0], wl[0], row, N, dump);
        const bool acc_mh = log_u < (((pr[0] + kk[0]) - c.prior) - c.kern);       // GLMCMC.py:96-99
#pragma unroll
        for (int r = 1; r < NA; ++r)
            team_candidate<D, YD, VAR, false, FAST>(a, rng, step, r, false, c, th[r], yy[r], lw[r], pr[r], kk[r], wl[r], row, N, dump);

        __syncthreads();                                                          // the helper's candidates of iteration t
        const TeamCand<D, YD, NH>& in = buf[t & 1];

        // ---- winner index: 0 = stay, k = candidate k-1 (chain_step, L = 1) ----
        float w[N + 1];
        w[0] = c.w_cur;                                                           // exp(log_weight_old), GLMCMC.py:75-81
#pragma unroll
        for (int r = 0; r < NA; ++r) w[1 + r] = wl[r];
#pragma unroll
        for (int r = 0; r < NH; ++r) w[1 + NA + r] = in.wl[r][lane];
        const float tot = aten_rowsum<N + 1>(w);                                  // GLMCMC.py:82
        const double u_res = glabc_uniform_f64(hd.v[2], hd.v[3]);
        int ig = -1;
        bool sure = !a.exact_index;
        {
            const float rinv = __builtin_amdgcn_rcpf(tot);
            const float u32 = (float)u_res;
            float run = 0.0f;
#pragma unroll
            for (int k = 0; k <= N; ++k) {
                run += w[k] * rinv;
                const float gap = u32 - run;
                sure = sure && (__builtin_fabsf(gap) > 4e-6f);
                ig = (ig < 0 && gap < 0.0f) ? k : ig;
            }
            sure = sure && (run > 0.999f) && (run < 1.001f);
        }
        if (!sure) {
            ig = -1;
            double run = 0.0;
#pragma unroll
            for (int k = 0; k <= N; ++k) {
                run += (double)(w[k] / tot);
                ig = (ig < 0 && u_res < run) ? k : ig;
            }
        }
        ig = ig < 0 ? 0 : ig;                                                     // None -> stay, GLMCMC.py:84
        const int ind = is_global ? ig : (acc_mh ? 1 : 0);

        // ---- move ----
        const bool moved = ind > 0;
        if (__any(moved)) {
            float nt[D], ny[YD], nlw = lw[0], npr = pr[0], nkk = kk[0], nw = wl[0];
#pragma unroll
            for (int q = 0; q < D; ++q) nt[q] = th[0][q];
#pragma unroll
            for (int q = 0; q < YD; ++q) ny[q] = yy[0][q];
#pragma unroll
            for (int r = 1; r < NA; ++r) {
                if (ind - 1 == r) {
#pragma unroll
                    for (int q = 0; q < D; ++q) nt[q] = th[r][q];
#pragma unroll
                    for (int q = 0; q < YD; ++q) ny[q] = yy[r][q];
                    nlw = lw[r];
                    npr = pr[r];
                    nkk = kk[r];
                    nw = wl[r];
                }
            }
            if (ind - 1 >= NA) {                                                  // one of the helper's candidates: from LDS
                const int r = ind - 1 - NA;
#pragma unroll
                for (int q = 0; q < D; ++q) nt[q] = in.th[r][q][lane];
#pragma unroll
                for (int q = 0; q < YD; ++q) ny[q] = in.yy[r][q][lane];
                nlw = in.lw[r][lane];
                npr = in.pr[r][lane];
                nkk = in.kk[r][lane];
                nw = in.wl[r][lane];
            }
            if (moved) {
#pragma unroll
                for (int q = 0; q < D; ++q) c.theta[q] = nt[q];
#pragma unroll
                for (int q = 0; q < YD; ++q) c.y[q] = ny[q];
                c.prior = npr;
                c.kern = nkk;
                c.q = dist_log_prob<D, GU, GM>(a.global, c.theta);
                c.lw_cur = nlw;
                c.w_cur = nw;
                if (is_global)
                    c.log_w = nlw;                                                // GLMCMC.py:86
                else
                    c.flags |= GLABC_FLAG_LOCAL;                                  // GLMCMC.py:100
            }
        }
        c.n_moves += moved ? 1u : 0u;

        if (hist && valid) {                                                      // Theta_Re[i,:] = Theta_old, GLMCMC.py:89,104
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

    if (valid) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.theta[j * a.stride + i] = c.theta[j];
#pragma unroll
        for (int j = 0; j < YD; ++j) a.y[j * a.stride + i] = c.y[j];
        a.log_w[i] = c.log_w;
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
}

// ---- GlobalMCMC (GlobalMCMC.py:37-68): a TEAM of two wavefronts per 64 chains ---------------------------------------------
//
// One candidate per iteration leaves nothing to split between wavefronts as above -- but 54 % of the iteration's vector
// instructions are its RANDOM NUMBERS (Philox blocks, Box-Muller, the logarithm of the accept uniform: 173 of 319, DESIGN.md
// 4.1), and those are a pure function of (seed, chain id, iteration) and of the chain's constant global_frequency.  The helper
// wavefront draws them ONE ITERATION AHEAD into LDS (branch, log u, the proposal's D draws -- normals or uniforms, whichever the
// branch's distribution takes -- and the simulator's normals); the main wavefront keeps the state and does the rest: proposal,
// simulator, prior, kernel, log q, the MH test of :44-47 / :60-61, update, Theta_Re row, sums.  65 536 chains are then two
// wavefronts per SIMD of about half the instructions each instead of one (which issues at best every other slot).  Geometry
// only: the draws are chain_step's words in chain_step's arithmetic, and so is everything the main wavefront computes from them.
constexpr int GLOBAL_TEAM_CHUNK = 8;
template <int D, int YD>
struct GlobalDraws {                        // one iteration, [field][lane]: conflict-free
    float log_u[64];                        // log of the accept uniform (-inf for u = 0)
    uint32_t is_global[64];
    float e[D][64];                         // the proposal's draws
    uint32_t sw[2 * ((NoiseDim<YD>::value + 1) / 2)][64];   // the Philox words of the simulator's normals (the main wavefront runs their
                                            // Box-Muller pairs: the two wavefronts then carry about the same number of instructions)
};

// NW = 3 splits the helper's work once more: wavefront 1 draws the step head (branch, log u), wavefront 2 the candidate's Philox
// block, the proposal's draws and the simulator's words.  Same chains; measured SLOWER than two at 65 536 chains (1.22 against
// 1.17 ms per 2000 iterations: a third barrier party and the main wavefront's dependent chain gain nothing from it), so two is what
// the library launches; GLABC_TEAM_WAVES=3 reaches this form (tests).
template <int D, int YD, int VAR, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW))) global_team_kernel(const StepArgs<D, YD> a, int prio)
{
    static_assert(NW == 2 || NW == 3, "a team of two or three wavefronts");
    constexpr bool GU = (VAR == VAR_GAUSS_UNIT);
    constexpr int DP = D + (D & 1);
    constexpr int ND = NoiseDim<YD>::value;
    constexpr int M = DP + ND;
    constexpr int SPP = (M + 3) / 4;
    constexpr int CH = GLOBAL_TEAM_CHUNK;                  // iterations per barrier: the wavefronts meet once per chunk, not per iteration
    __shared__ GlobalDraws<D, YD> buf[2][CH];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int64_t tid = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = tid < a.n_chains;
    const int64_t i = valid ? tid : a.n_chains - 1;        // tail lanes shadow the last chain (no stores)
    const uint64_t gid = (uint64_t)(a.chain0 + i);
    Rng rng;
    rng.c0 = (uint32_t)gid;
    rng.c1 = (uint32_t)(gid >> 32);
    rng.k0 = a.seed_lo;
    rng.k1 = a.seed_hi;
    const float gf = a.gf_chain ? a.gf_chain[i] : a.gf;

    if (wave != 0) {                                       // ---- helpers: the draws of a chunk of iterations, one chunk ahead ----
        const bool g_uni = !GU && a.global.kind == GLABC_DIST_UNIFORM;
        const bool l_uni = !GU && a.local.kind == GLABC_DIST_UNIFORM;
        const bool do_head = NW == 2 || wave == 1, do_cand = NW == 2 || wave == 2;       // wave-uniform
        const bool need_branch = do_head || g_uni != l_uni;                                // which kind of draw the proposal takes
#pragma unroll 1
        for (int t0 = 0; t0 < a.n_steps; t0 += CH) {
#pragma unroll 1
          for (int t = t0; t < t0 + CH && t < a.n_steps; ++t) {
            const uint32_t step = a.step0 + (uint32_t)t;
            GlobalDraws<D, YD>& o = buf[(t0 / CH) & 1][t - t0];
            bool is_global = false;
            if (need_branch) {
                const glabc_u32x4 h = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);   // the step head, chain_step
                const float ub = glabc_uniform_f32(h.v[0]);
                is_global = ub < gf;                                                                    // GlobalMCMC.py:39
                if (do_head) {
                    const float ua = glabc_uniform_f32(h.v[1]);
                    o.log_u[lane] = (ua == 0.0f) ? -__builtin_inff() : glabc_logf_normal(ua);
                    o.is_global[lane] = is_global ? 1u : 0u;
                }
            }
            if (do_cand) {
                uint32_t w[4 * SPP];
#pragma unroll
                for (int b = 0; b < SPP; ++b) {
                    const glabc_u32x4 v = glabc_philox4x32_10(rng.c0, rng.c1, step, (uint32_t)(1 + b), rng.k0, rng.k1);   // candidate 0
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[4 * b + q] = v.v[q];
                }
                const bool uni = is_global ? g_uni : l_uni;    // (equal kinds: the branch does not matter and was not drawn)
                float nrm[DP];                                 // the proposal's pairs (words 0 .. DP-1)
#pragma unroll
                for (int k = 0; 2 * k < DP; ++k) glabc_normal_pair(w[2 * k], w[2 * k + 1], &nrm[2 * k], &nrm[2 * k + 1]);
#pragma unroll
                for (int k = 0; k < D; ++k) o.e[k][lane] = (!GU && uni) ? glabc_uniform_f32(w[k]) : nrm[k];
#pragma unroll
                for (int k = 0; k < 2 * ((ND + 1) / 2); ++k) o.sw[k][lane] = w[DP + k];
            }
          }
            __syncthreads();                               // the chunk's draws are in LDS
        }
        return;
    }

    // ---- main: state, proposal, simulator, densities, decision ----
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    Chain<D, YD> c;
#pragma unroll
    for (int j = 0; j < D; ++j) c.theta[j] = a.theta[j * a.stride + i];
#pragma unroll
    for (int j = 0; j < YD; ++j) c.y[j] = a.y[j * a.stride + i];
    c.log_w = 0.0f;
    c.flags = 0u;
    c.n_moves = a.n_moves ? a.n_moves[i] : 0u;
    c.gf = gf;
    refresh_cache<D, YD>(a, c);
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
    float* hist = a.history ? a.history + i : nullptr;

#pragma unroll 1
    for (int t0 = 0; t0 < a.n_steps; t0 += CH) {
      __syncthreads();                                     // the helper has left this chunk's draws (it is already on the next)
#pragma unroll 1
      for (int t = t0; t < t0 + CH && t < a.n_steps; ++t) {
        float prev[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = c.theta[j];
        const GlobalDraws<D, YD>& in = buf[(t0 / CH) & 1][t - t0];
        const float log_u = in.log_u[lane];
        const bool is_global = in.is_global[lane] != 0u;
        float e[D], sn[ND];
#pragma unroll
        for (int k = 0; k < D; ++k) e[k] = in.e[k][lane];
        {
            float nn[2 * ((ND + 1) / 2)];                  // the simulator's pairs (words DP ..): chain_step's normals nrm[DP + k]
#pragma unroll
            for (int k = 0; 2 * k < ND; ++k) glabc_normal_pair(in.sw[2 * k][lane], in.sw[2 * k + 1][lane], &nn[2 * k], &nn[2 * k + 1]);
#pragma unroll
            for (int k = 0; k < ND; ++k) sn[k] = nn[k];
        }
        const bool loc = !is_global;
        float th[D], yy[YD];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const float p0 = loc ? a.local.p0[q] : a.global.p0[q];
            const float p2 = loc ? a.local.p2[q] : a.global.p2[q];
            const float tt = p0 + p2 * e[q];                                  // distribution.py:170 / :77
            th[q] = loc ? (tt + c.theta[q]) : tt;                             // GlobalMCMC.py:40 / :56
        }
        const float lq = dist_forward_log_p<D, GU>(a.global, e);              // unused by the local move
        model_simulate<D, YD>(a, th, sn, yy);
        const float pr = model_prior<D, YD, GU, false>(a, th);
        const float kk = model_log_kernel<D, YD, GU>(a, yy);
        const float pk = pr + kk;
        const float log_acc = loc ? ((pk - c.prior) - c.kern)                 // GlobalMCMC.py:60-61
                                  : ((((pk + c.q) - lq) - c.prior) - c.kern); // GlobalMCMC.py:44-46
        const bool moved = log_u < log_acc;                                   // GlobalMCMC.py:47 / :62
        if (moved) {
#pragma unroll
            for (int q = 0; q < D; ++q) c.theta[q] = th[q];
#pragma unroll
            for (int q = 0; q < YD; ++q) c.y[q] = yy[q];
            c.prior = pr;
            c.kern = kk;
            c.q = dist_log_prob<D, GU>(a.global, c.theta);
        }
        c.n_moves += moved ? 1u : 0u;
        if (hist && valid) {                                                  // Theta_Re[i,:] = Theta_old, GlobalMCMC.py:49,66
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
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.theta[j * a.stride + i] = c.theta[j];
#pragma unroll
        for (int j = 0; j < YD; ++j) a.y[j * a.stride + i] = c.y[j];
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

// LDS of a global_team_kernel workgroup: two chunks of GLOBAL_TEAM_CHUNK iterations' draws
constexpr int global_team_lds_bytes(int d, int nd) { return 2 * GLOBAL_TEAM_CHUNK * (2 + d + 2 * ((nd + 1) / 2)) * 64 * 4; }

// LDS of one workgroup (two iterations of the helpers' candidates); a CU hosts 1024 / 256 = 4 workgroups of a 65 536-chain launch
constexpr int team_lds_bytes(int d, int yd, int n, int nw) { return 2 * (n - team_main_candidates(n, nw)) * (4 + d + yd) * 64 * 4; }
constexpr int TEAM_MAX_LDS = 40 * 1024;
constexpr bool team_config_ok(int d, int yd, int n, int nw)
{
    return n >= 2 && n <= GLABC_MAX_BATCH && team_split_ok(n, nw) && team_lds_bytes(d, yd, n, nw) <= TEAM_MAX_LDS;
}

#if !defined(__HIPCC_RTC__)
// host-side launcher of one (theta_dim, y_dim); defined in glabc_team_dim.hip.  nw = wavefronts per 64 chains (2, 3 or 4).
// GLABC_ERR_ARG when the configuration has no such team kernel (too few candidates, or candidates beyond the LDS budget).
template <int D, int YD>
int launch_team_dim(int n_batch, int nw, const StepArgs<D, YD>& a, int prio, bool fast, hipStream_t stream);
// ... and of global_team_kernel (GlobalMCMC, Gaussian / Uniform descriptors)
template <int D, int YD>
int launch_global_team_dim(int nw, const StepArgs<D, YD>& a, int prio, hipStream_t stream);
#endif

}  // namespace glabc
