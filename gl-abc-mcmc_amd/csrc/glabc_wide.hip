// glabc_wide.hip -- GLMCMC's iSIR move for batch sizes beyond the register kernels (N > GLABC_MAX_BATCH): a group of L
// lanes of one wavefront owns a chain and shares its N candidates.
//
//   candidates   lane `sub` of the group evaluates candidates j = sub, sub + L, ... one at a time (same Philox slots, words and
//                operation order as chain_step, glabc_device.h) and leaves its weight exp((prior' + K') - q') in the group's
//                LDS row w[1 + j]; w[0] is the current state's weight                                  GLMCMC.py:66-81
//   total        torch.sum's association (GLMCMC.py:82) is a fixed tree over 32 accumulator lanes (8 vector lanes x 4
//                accumulators, cascade levels inside, glabc_generic.hip / DESIGN.md "row-sum order"): the 32 lane sums are
//                dealt to the group's lanes, combined in ATen's order by every lane
//   index        weight_sampling (GLMCMC.py:7-22) is a sequential double running sum against a double uniform.  Each lane
//                sums its contiguous chunk of w_k / total in double, an exclusive scan over the group's lanes (DPP / bpermute
//                shuffles) gives every lane its starting partial sum, and the first k with u < partial sum is the minimum over
//                lanes.  A scan adds in another order than the reference's loop: the partial sums differ from the sequential
//                ones by at most n 2^-53, so the index can differ only if u lies within that of one of them -- lanes check
//                |u - partial| <= n 2^-51 and the whole group then redoes the sequential loop (never seen in tests; a
//                debug flag forces it)
//   winner       with one candidate per lane (N <= L) the winner is fetched from its owner by shuffles; otherwise every lane
//                re-evaluates candidate `ind - 1` (deterministic: same Philox counter -> same bits)
//
// The local move (GLMCMC.py:90-104) is candidate 0 evaluated as theta + increment by every lane of the group.  State is
// replicated over the group's lanes (registers); lane 0 writes history, sums and the final state.  Results equal the CPU
// checker's bit for bit for every L, like the register kernels' (tests/test_hip_parity.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "glabc_lds_grant.h"
#include "glabc_sampler.h"

namespace glabc {

constexpr int WIDE_BLOCK = 256;

struct Cand {
    float lw, wl, pr, kk, log_acc;
};

// candidate j of (chain, step): the arithmetic of chain_step's generic variant for one slot
// GM: the instantiation also knows GLABC_DIST_GAMMA as importance proposal / prior (chain_step's VAR_GAMMA)
template <int D, int YD, bool GM>
GLABC_DEV Cand eval_candidate(const StepArgs<D, YD>& a, const Rng& rng, uint32_t step, int j, bool loc, const Chain<D, YD>& c,
                              float (&th)[D], float (&yy)[YD])
{
    constexpr int DP = D + (D & 1), M = DP + YD, SPP = (M + 3) / 4;
    uint32_t w[4 * SPP];
#pragma unroll
    for (int b = 0; b < SPP; ++b) {
        const glabc_u32x4 o = glabc_philox4x32_10(rng.c0, rng.c1, step, (uint32_t)(1 + j * SPP + b), rng.k0, rng.k1);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[4 * b + q] = o.v[q];
    }
    const bool uni = (loc ? a.local.kind : a.global.kind) == GLABC_DIST_UNIFORM;
    float nrm[2 * ((M + 1) / 2)], e[D], s[YD];
#pragma unroll
    for (int i = 0; 2 * i < M; ++i) glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < D; ++i) e[i] = uni ? glabc_uniform_f32(w[i]) : nrm[i];
#pragma unroll
    for (int i = 0; i < YD; ++i) s[i] = nrm[DP + i];
#pragma unroll
    for (int q = 0; q < D; ++q) {
        const float p0 = loc ? a.local.p0[q] : a.global.p0[q];
        const float p2 = loc ? a.local.p2[q] : a.global.p2[q];
        const float t = p0 + p2 * e[q];                                       // distribution.py:170 / :77
        th[q] = loc ? (t + c.theta[q]) : t;                                   // GLMCMC.py:91
    }
    float lq_gamma = 0.0f;
    const bool g_gam = GM && a.global.kind == GLABC_DIST_GAMMA;
    if constexpr (GM) {
        if (g_gam) {                                                          // wave-uniform; local-branch lanes keep theta + increment
            float tg[D];
            dist_gamma_forward<D>(a.global, rng.c0, rng.c1, rng.k0, rng.k1, step, j, tg, lq_gamma);
#pragma unroll
            for (int q = 0; q < D; ++q) th[q] = loc ? th[q] : tg[q];
        }
    }
    const float lq = loc ? dist_log_prob<D, false, GM>(a.global, th) : (g_gam ? lq_gamma : dist_forward_log_p<D>(a.global, e));
    model_simulate<D, YD>(a, th, s, yy);                                      // GLMCMC.py:71,94
    Cand r;
    r.pr = dist_log_prob<D, false, GM>(a.prior, th);
    r.kk = model_log_kernel<D, YD>(a, yy);
    const float pk = r.pr + r.kk;
    r.lw = pk - lq;                                                           // GLMCMC.py:74
    r.log_acc = (pk - c.prior) - c.kern;                                      // GLMCMC.py:96-97
    const float v = glabc_expf(r.lw);                                         // GLMCMC.py:78
    r.wl = (v != v) ? 0.0f : v;                                               // GLMCMC.py:80-81
    return r;
}

template <int L>
GLABC_DEV float grp_get(float v, int src_sub)
{
    const int lane = (int)(threadIdx.x & 63u);
    return __shfl(v, (lane & ~(L - 1)) | src_sub, 64);
}

template <int L>
GLABC_DEV int grp_get_i(int v, int src_sub)
{
    const int lane = (int)(threadIdx.x & 63u);
    return __shfl(v, (lane & ~(L - 1)) | src_sub, 64);
}

template <int D, int YD, int L, bool GM>
__global__ void __launch_bounds__(WIDE_BLOCK) wide_kernel(const StepArgs<D, YD> a, const int N)
{
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    constexpr int GROUPS = WIDE_BLOCK / L;
    const int sub = (int)(threadIdx.x % L), grp = (int)(threadIdx.x / L);
    const int64_t chain = (int64_t)blockIdx.x * GROUPS + grp;
    const bool valid = chain < a.n_chains;
    const int64_t i = valid ? chain : a.n_chains - 1;          // tail groups shadow the last chain (no stores)
    const bool writer = valid && sub == 0;
    const int n = N + 1;
    float* w = wide_lds + (size_t)grp * (n + 32);              // this group's weights w[0..N] ...
    float* pbuf = w + n;                                       // ... and the 32 accumulator-lane sums of torch.sum

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
    c.lw_cur = (c.flags & GLABC_FLAG_LOCAL) ? (c.prior + c.kern) - c.q : c.log_w;          // GLMCMC.py:60-64
    {
        const float v = glabc_expf(c.lw_cur);
        c.w_cur = (v != v) ? 0.0f : v;
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
    const int rounds_all = (N + L - 1) / L;
    const int nv = n / 8, G = nv / 4;
    const int chunk = (n + L - 1) / L;
    const double margin = (double)n * 0x1p-51;

    for (int t = 0; t < a.n_steps; ++t) {
        const uint32_t step = a.step0 + (uint32_t)t;
        float prev[D];
#pragma unroll
        for (int j = 0; j < D; ++j) prev[j] = c.theta[j];

        // ---- step head (replicated over the group) ----
        const glabc_u32x4 h = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
        const float ub = glabc_uniform_f32(h.v[0]), ua = glabc_uniform_f32(h.v[1]);
        const float log_u = (ua == 0.0f) ? -__builtin_inff() : glabc_logf_normal(ua);       // GLMCMC.py:98
        const bool is_global = ub < c.gf;                                                   // GLMCMC.py:59
        const double u_res = glabc_uniform_f64(h.v[2], h.v[3]);
        if (is_global) {
            if (c.flags & GLABC_FLAG_LOCAL) c.log_w = c.lw_cur;                             // GLMCMC.py:60-64
            c.flags &= ~GLABC_FLAG_LOCAL;                                                   // GLMCMC.py:65
        }

        // ---- candidates ----
        float th[D], yy[YD];
        Cand cd;
        cd.lw = cd.wl = cd.pr = cd.kk = 0.0f;
        cd.log_acc = -__builtin_inff();
        const int rounds = __any(is_global) ? rounds_all : 1;
        for (int r = 0; r < rounds; ++r) {
            const int j = is_global ? sub + L * r : 0;                  // a chain on the local branch: candidate 0, every lane
            if ((is_global && j < N) || (!is_global && r == 0)) {
                cd = eval_candidate<D, YD, GM>(a, rng, step, j, !is_global, c, th, yy);
                if (is_global) w[1 + j] = cd.wl;
            }
        }
        if (sub == 0) w[0] = c.w_cur;                                   // exp(log_weight_old), GLMCMC.py:75-81
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        int ind = 0;
        if (__any(is_global)) {
            // ---- total: torch.sum's tree (n >= 18: the vector path) ----
            for (int idx = sub; idx < 32; idx += L) {
                const int q = idx >> 3, k = idx & 7;
                float p = cascade_lane([&](int i2) { return w[8 * (4 * i2 + q) + k]; }, G);
                if (q == 0)
                    for (int v = 4 * G; v < nv; ++v) p = p + w[8 * v + k];
                pbuf[idx] = p;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float tot = 0.0f;
            for (int k = 8 * nv; k < n; ++k) tot = tot + w[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) tot = tot + (((pbuf[k] + pbuf[8 + k]) + pbuf[16 + k]) + pbuf[24 + k]);     // GLMCMC.py:82

            // ---- index: chunked double prefix sums ----
            const int k0 = sub * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
            double loc_sum = 0.0;
            for (int k = k0; k < k1; ++k) loc_sum += (double)(w[k] / tot);
            double incl = loc_sum;                                      // inclusive scan over the group's lanes
#pragma unroll
            for (int off = 1; off < L; off <<= 1) {
                const int lane = (int)(threadIdx.x & 63u);
                const double up = __shfl(incl, lane - off, 64);
                if (sub >= off) incl += up;
            }
            double run = __shfl(incl, (int)(threadIdx.x & 63u) - 1, 64);       // exclusive: the lane below's inclusive sum
            if (sub == 0) run = 0.0;
            int found = 0x7fffffff;
            bool unsure = a.exact_index != 0;
            for (int k = k0; k < k1; ++k) {
                run += (double)(w[k] / tot);
                const double gap = u_res - run;
                unsure = unsure || !(__builtin_fabs(gap) > margin);
                if (found == 0x7fffffff && gap < 0.0) found = k;
            }
#pragma unroll
            for (int off = 1; off < L; off <<= 1) {                     // minimum / any over the group
                const int lane = (int)(threadIdx.x & 63u);
                const int of = __shfl(found, lane ^ off, 64);
                const int ou = __shfl(unsure ? 1 : 0, lane ^ off, 64);
                found = of < found ? of : found;
                unsure = unsure || ou != 0;
            }
            if (unsure && is_global) {                                  // the reference's loop, GLMCMC.py:17-22
                found = 0x7fffffff;
                double acc = 0.0;
                for (int k = 0; k < n; ++k) {
                    acc += (double)(w[k] / tot);
                    if (u_res < acc) {
                        found = k;
                        break;
                    }
                }
            }
            ind = found == 0x7fffffff ? 0 : found;                      // None -> stay, GLMCMC.py:84
        }
        if (!is_global) ind = (log_u < cd.log_acc) ? 1 : 0;             // GLMCMC.py:98-99 (every lane evaluated candidate 0)
        __builtin_amdgcn_wave_barrier();                                // the group's w row is rewritten by the next step

        // ---- move ----
        const bool moved = ind > 0;
        if (__any(moved)) {
            if (rounds_all == 1 || !is_global) {
                // the winner is still in its owner's registers (one candidate per lane / the local candidate on every lane)
                const int owner = is_global ? ind - 1 : sub;
                const int src = moved ? owner : sub;
#pragma unroll
                for (int q = 0; q < D; ++q) th[q] = grp_get<L>(th[q], src);
#pragma unroll
                for (int q = 0; q < YD; ++q) yy[q] = grp_get<L>(yy[q], src);
                cd.lw = grp_get<L>(cd.lw, src);
                cd.wl = grp_get<L>(cd.wl, src);
                cd.pr = grp_get<L>(cd.pr, src);
                cd.kk = grp_get<L>(cd.kk, src);
            }
            if (rounds_all > 1 && __any(moved && is_global)) {
                // several candidates per lane: every lane re-evaluates the winner (same counter, same bits)
                float th2[D], yy2[YD];
                const Cand c2 = eval_candidate<D, YD, GM>(a, rng, step, moved && is_global ? ind - 1 : 0, false, c, th2, yy2);
                if (moved && is_global) {
#pragma unroll
                    for (int q = 0; q < D; ++q) th[q] = th2[q];
#pragma unroll
                    for (int q = 0; q < YD; ++q) yy[q] = yy2[q];
                    cd = c2;
                }
            }
            if (moved) {
#pragma unroll
                for (int q = 0; q < D; ++q) c.theta[q] = th[q];
#pragma unroll
                for (int q = 0; q < YD; ++q) c.y[q] = yy[q];
                c.prior = cd.pr;
                c.kern = cd.kk;
                c.q = dist_log_prob<D, false, GM>(a.global, c.theta);
                c.lw_cur = cd.lw;
                c.w_cur = cd.wl;
                if (is_global)
                    c.log_w = cd.lw;                                    // GLMCMC.py:86
                else
                    c.flags |= GLABC_FLAG_LOCAL;                        // GLMCMC.py:100
                c.n_moves += 1u;
            }
        }

        if (hist && writer) {                                           // Theta_Re[i,:] = Theta_old, GLMCMC.py:89,104
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
                    const double dp = (double)c.theta[p] - (double)prev[p];
                    const double dq = (double)c.theta[q] - (double)prev[q];
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

template <int D, int YD, int L, bool GM>
static int launch_wide_lg(const StepArgs<D, YD>& a, int N, hipStream_t s)
{
    constexpr int GROUPS = WIDE_BLOCK / L;
    const size_t lds = sizeof(float) * (size_t)GROUPS * (size_t)(N + 1 + 32);
    static LdsGrant grant;                               // per instantiation, per device
    if (!grant_dynamic_lds(grant, (const void*)wide_kernel<D, YD, L, GM>, lds)) return GLABC_ERR_LAUNCH;
    const unsigned grid = (unsigned)((a.n_chains + GROUPS - 1) / GROUPS);
    hipLaunchKernelGGL((wide_kernel<D, YD, L, GM>), dim3(grid), dim3(WIDE_BLOCK), lds, s, a, N);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

template <int D, int YD, int L>
static int launch_wide_l(const StepArgs<D, YD>& a, int N, hipStream_t s)
{
    if (a.prior.kind == GLABC_DIST_GAMMA || a.global.kind == GLABC_DIST_GAMMA) {      // Gamma importance proposal / prior
        if constexpr (YD == D) return launch_wide_lg<D, YD, L, true>(a, N, s);
        else return GLABC_ERR_KIND;
    }
    return launch_wide_lg<D, YD, L, false>(a, N, s);
}

// lanes per chain: the smallest group that keeps a lane at no more than 8 candidates (the per-step head, total and index
// search are executed by every lane of the group, so small groups amortise them best), or the caller's choice
template <int D, int YD>
int launch_wide(const StepArgs<D, YD>& a, int N, int lanes, hipStream_t s)
{
    if (lanes <= 0) lanes = N <= 64 ? 8 : N <= 128 ? 16 : N <= 256 ? 32 : 64;
    switch (lanes) {
    case 8: return launch_wide_l<D, YD, 8>(a, N, s);
    case 16: return launch_wide_l<D, YD, 16>(a, N, s);
    case 32: return launch_wide_l<D, YD, 32>(a, N, s);
    case 64: return launch_wide_l<D, YD, 64>(a, N, s);
    default: return GLABC_ERR_ARG;
    }
}

template int launch_wide<1, 1>(const StepArgs<1, 1>&, int, int, hipStream_t);
template int launch_wide<2, 2>(const StepArgs<2, 2>&, int, int, hipStream_t);
template int launch_wide<3, 3>(const StepArgs<3, 3>&, int, int, hipStream_t);
template int launch_wide<4, 4>(const StepArgs<4, 4>&, int, int, hipStream_t);
template int launch_wide<4, 8>(const StepArgs<4, 8>&, int, int, hipStream_t);

}  // namespace glabc
