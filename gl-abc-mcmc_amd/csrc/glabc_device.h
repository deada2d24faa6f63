// glabc_device.h -- per-chain device code of the fused GL-ABC-MCMC step (gfx950).
//
// Work decomposition.  A chain is owned by a group of L adjacent lanes (L = 1, 2, 4 or 8,
// a launch-geometry choice that never changes results).  All L lanes keep the chain's
// state (theta, y, cached densities, iSIR log-weight, flags) in VGPRs for the whole
// launch; the N proposals of an iteration are dealt round-robin to the lanes
// (proposal j -> lane j % L, slot j / L), the step-head draw goes to the last lane, and
// three small exchanges per iteration (ds_bpermute within the group) put the N weights,
// the local accept bit and the winning candidate on every lane.  With L = 1 this is the
// plain one-work-item-per-chain form; L > 1 exists because at 65 536 chains one
// work-item per chain gives one wave per SIMD, and a lone wave can issue a VALU
// instruction only every other slot (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").
//
// The local move shares proposal slot 0 with the global move: a wavefront holds chains
// on both sides of the `u < global_frequency` test at almost every iteration, so the two
// bodies are merged into one instruction stream with per-lane parameter selects instead
// of two serialized divergent branches.
//
// Everything that is the same for all chains (model / proposal parameters, seed,
// global_frequency) arrives in the kernel argument block, i.e. in SGPRs.
//
// The arithmetic follows the reference line by line (citations = /root/reference paths)
// in float32 with -ffp-contract=off, using the elementary functions and the Philox stream
// of include/glabc_numerics.h, so the CPU checker in oracle/ (an independent plain-C
// restatement) can be compared bit for bit.  Deviations from a literal transcription are
// limited to ones that cannot change a bit: prior(theta_old), K(y_old) and q(theta_old)
// are cached instead of recomputed (pure functions of the state), and x / 1.0f is x.
#pragma once

#if !defined(__HIPCC_RTC__)          // hiprtc (glabc_rtc.hip) brings the HIP device declarations and the fixed-width types itself
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "../../include/glabc.h"
#include "../../include/glabc_numerics.h"

namespace glabc {

#define GLABC_DEV static __device__ __forceinline__

enum Algo { ALGO_GLMCMC = 0, ALGO_GLOBAL = 1 };

// Kernel variants.  VAR_GENERIC reads the distribution kinds / unit-scale flags from the
// argument block (wave-uniform branches).  VAR_GAUSS_UNIT is the reference example's class of
// configuration -- prior, local and global proposals all DiagGaussian, prior and global with
// exp(log_scale) == 1 (examples/Mixture.py:30,68) -- with those facts known at compile time,
// which removes every branch from a candidate's evaluation so the scheduler can interleave
// the independent candidates of a lane.
// VAR_TAPE = VAR_GENERIC with the random numbers replayed from glabc_run.tape instead of Philox (one lane per chain).
// VAR_GAMMA = VAR_GENERIC that also knows GLABC_DIST_GAMMA as the global / importance proposal and as the prior (one lane per
// chain): its float64 log-density and Marsaglia-Tsang loop stay out of the other instantiations' register budgets.
enum Variant { VAR_GENERIC = 0, VAR_GAUSS_UNIT = 1, VAR_TAPE = 2, VAR_GAMMA = 3 };

// ---- argument block ------------------------------------------------------------
template <int D>
struct DistArgs {
    int32_t kind;
    int32_t unit_scale;          // DiagGaussian with every log_scale == 0 and exp(log_scale) == 1.0f:
                                 // (z-loc)/1 == z-loc and 0 + 0.5 e^2 == 0.5 e^2, bit for bit
    float c0;
    float p0[D], p1[D], p2[D];
    float p3[D];                 // Gamma: gammaln(shape) (glabc_dist.p3)
};

template <int D, int YD = D>
struct StepArgs {
    // Model callbacks (examples/Mixture.py:13-45).  sim_kind GLABC_SIM_ABS_GAUSS: y = |theta| + noise (y_dim == theta_dim);
    // GLABC_SIM_GK: y = sorted g-and-k variates (theta = (A, B, g, k), examples/GK.py)
    DistArgs<D> prior;
    int32_t sim_kind;
    float gk_c;
    float noise_loc[YD], noise_scale[YD];
    float y_obs[YD];
    float kern_log_scale, kern_scale, kern_c0;
    int32_t y_obs_away;          // every |y_obs_j| >= 2^-6: a sum of squared differences is 0 or >= 2^-62 (lean sqrt domain)
    float kern_rinv;             // RN(1/kern_scale) if x/kern_scale == fma(fma(-q, s, x), rinv, q), q = x*rinv, was verified
                                 // on the host for every float32 significand of x (verified_reciprocal); else 0
    // proposals: local increment (GLMCMC.py:91) and global / importance (GLMCMC.py:66, GlobalMCMC.py:40)
    DistArgs<D> local, global;
    // chains
    float* theta;
    float* y;
    float* log_w;
    uint32_t* flags;
    uint32_t* n_moves;
    int64_t n_chains, chain0, stride;
    // run
    uint32_t seed_lo, seed_hi, step0;
    int32_t n_steps;
    float gf;
    const float* gf_chain;       // NULL or [n_chains]: per-chain global_frequency (glabc_run.global_frequency_per_chain)
    float* history;
    int64_t hist_stride;
    double* sum_theta;
    double* sum_outer;
    double* sum_jump;
    // replayed random numbers (glabc_tape; VAR_TAPE only)
    const float* tape_u;
    const double* tape_r;
    const float* tape_z;
    int32_t tape_nprop;
    int32_t exact_index;          // GLABC_DEBUG_EXACT_INDEX
    // GLABC_MATH_FAST only: where the kernel records the draws it used (glabc_draws_out), or NULL
    float* dump_u;
    double* dump_r;
    float* dump_z;
};

// ---- torch.sum association over a contiguous float32 row (GLMCMC.py:82) ---------
// Probed on the reference's torch build (DESIGN.md "row-sum order"): n < 8 -> four
// scalar lanes, leftovers into lane 0, lanes combined left to right; n >= 8 -> the
// same scheme over 8-wide vectors, then a scalar accumulator takes the n%8 tail in
// order followed by the 8 vector partials in order.  N is a compile-time constant,
// so all of this unrolls into a fixed add tree.
template <int N>
GLABC_DEV float aten_rowsum(const float (&x)[N])
{
    if constexpr (N < 4) {
        float s = x[0];
#pragma unroll
        for (int i = 1; i < N; ++i) s = s + x[i];
        return s;
    } else if constexpr (N < 8) {
        float l0 = x[0];
#pragma unroll
        for (int i = 4; i < N; ++i) l0 = l0 + x[i];
        return ((l0 + x[1]) + x[2]) + x[3];
    } else {
        constexpr int NV = N / 8;
        constexpr int G = NV / 4;
        float acc[8];
        if constexpr (G == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = x[k];
#pragma unroll
                for (int v = 1; v < NV; ++v) acc[k] = acc[k] + x[8 * v + k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) l[q] = x[8 * q + k];
#pragma unroll
                for (int i = 1; i < G; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) l[q] = l[q] + x[8 * (4 * i + q) + k];
#pragma unroll
                for (int v = 4 * G; v < NV; ++v) l[0] = l[0] + x[8 * v + k];
                acc[k] = ((l[0] + l[1]) + l[2]) + l[3];
            }
        }
        float fa = 0.0f;
#pragma unroll
        for (int i = 8 * NV; i < N; ++i) fa = fa + x[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) fa = fa + acc[k];
        return fa;
    }
}

// ---- torch.sum for run-time lengths (glabc_generic.hip, glabc_wide.hip) ----------------------
// torch.sum over a short register row t[0..n), n <= 8 (distribution.py:172,180) -- the association of aten_rowsum<N>
// (glabc_device.h) spelled with predicates so that t is only ever indexed statically
GLABC_DEV float rowsum_small(const float (&t)[GLABC_MAX_DIM], int n)
{
    if (n < 4) {
        float s = t[0];
        if (n > 1) s = s + t[1];
        if (n > 2) s = s + t[2];
        return s;
    }
    if (n < 8) {
        float l0 = t[0];
        if (n > 4) l0 = l0 + t[4];
        if (n > 5) l0 = l0 + t[5];
        if (n > 6) l0 = l0 + t[6];
        return ((l0 + t[1]) + t[2]) + t[3];
    }
    float fa = 0.0f;                                  // one 8-wide vector: the eight partials added in order
#pragma unroll
    for (int k = 0; k < 8; ++k) fa = fa + t[k];
    return fa;
}

GLABC_DEV int ceil_log2_i(int x)                      // c10 utils::CeilLog2
{
    if (x <= 2) return 1;
    return 32 - __builtin_clz((unsigned)(x - 1));
}

// One accumulator lane of ATen's multi_row_sum (aten/src/ATen/native/cpu/SumKernel.cpp): s(0..G) summed into four cascade
// levels -- every 2^lp additions level 0 is folded into level 1, every 2^(2 lp) level 1 into level 2, ... -- lp =
// max(4, CeilLog2(G)/4).  For G < 16 this is the plain sequential sum.
template <typename F>
GLABC_DEV float cascade_lane(F s, int G)
{
    const int lp = ceil_log2_i(G) / 4 > 4 ? ceil_log2_i(G) / 4 : 4;
    const int step = 1 << lp, mask = step - 1;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int i = 0;
    while (i + step <= G) {
        for (int j = 0; j < step; ++j, ++i) a0 = a0 + s(i);
        a1 = a1 + a0;
        a0 = 0.0f;
        if ((i & (mask << lp)) != 0) continue;
        a2 = a2 + a1;
        a1 = 0.0f;
        if ((i & (mask << (2 * lp))) != 0) continue;
        a3 = a3 + a2;
        a2 = 0.0f;
    }
    for (; i < G; ++i) a0 = a0 + s(i);
    a0 = a0 + a1;
    a0 = a0 + a2;
    a0 = a0 + a3;
    return a0;
}

// torch.sum over a contiguous float32 row x(0..n) of ANY length, as the reference's torch build associates it (probed for
// n up to 20 000, tests/golden/primitives.npz rowsum_*): n < 8 -> four scalar lanes; n >= 8 -> 8-wide vectors dealt to four
// accumulators (vector v -> accumulator v % 4 within the full groups of four, cascade levels inside each accumulator, the
// nv % 4 leftover vectors into accumulator 0), accumulators combined left to right, then a scalar takes the n % 8 tail in
// order followed by the eight vector partials in order.  Same results as the compile-time aten_rowsum<N> (glabc_device.h).
template <typename F>
GLABC_DEV float aten_rowsum_rt(F x, int n)
{
    if (n < 8) {
        if (n < 4) {
            float s = x(0);
            for (int i = 1; i < n; ++i) s = s + x(i);
            return s;
        }
        float l0 = x(0);
        for (int i = 4; i < n; ++i) l0 = l0 + x(i);
        return ((l0 + x(1)) + x(2)) + x(3);
    }
    const int nv = n / 8, G = nv / 4;
    float fa = 0.0f;
    for (int i = 8 * nv; i < n; ++i) fa = fa + x(i);
    for (int k = 0; k < 8; ++k) {
        float p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = cascade_lane([&](int i) { return x(8 * (4 * i + q) + k); }, G);
        for (int v = 4 * G; v < nv; ++v) p[0] = p[0] + x(8 * v + k);
        fa = fa + (((p[0] + p[1]) + p[2]) + p[3]);
    }
    return fa;
}

// ATen's float64 row sum: the float32 scheme of aten_rowsum with 4-wide vectors (probed)
template <int N>
GLABC_DEV double aten_rowsum_f64(const double (&x)[N])
{
    if constexpr (N < 4) {
        double s = x[0];
#pragma unroll
        for (int i = 1; i < N; ++i) s = s + x[i];
        return s;
    } else {
        constexpr int NV = N / 4;
        constexpr int G = NV / 4;
        double acc[4];
        if constexpr (G == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[k] = x[k];
#pragma unroll
                for (int v = 1; v < NV; ++v) acc[k] = acc[k] + x[4 * v + k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) l[q] = x[4 * q + k];
#pragma unroll
                for (int i = 1; i < G; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) l[q] = l[q] + x[4 * (4 * i + q) + k];
#pragma unroll
                for (int v = 4 * G; v < NV; ++v) l[0] = l[0] + x[4 * v + k];
                acc[k] = ((l[0] + l[1]) + l[2]) + l[3];
            }
        }
        double fa = 0.0;
#pragma unroll
        for (int i = 4 * NV; i < N; ++i) fa = fa + x[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) fa = fa + acc[k];
        return fa;
    }
}

// ---- distribution.py ---------------------------------------------------------------
// DiagGaussian.log_prob, distribution.py:176-181 / Uniform.log_prob, distribution.py:81-86
// Gamma.log_prob (distribution.py:123-137) at a float32 point: float64 per coordinate, torch.sum's float64 order, rounded once
template <int D>
GLABC_DEV float dist_log_prob_gamma(const DistArgs<D>& g, const float (&z)[D])
{
    double t[D];
#pragma unroll
    for (int j = 0; j < D; ++j) t[j] = glabc_gamma_log_pdf((double)g.p0[j], (double)g.p2[j], (double)g.p3[j], (double)z[j]);
    return (float)aten_rowsum_f64<D>(t);
}

// Gamma.forward (distribution.py:106-121) for candidate j of (chain, step): theta' = (float) z, log q' = (float) log_prob(z) of
// the DOUBLE variate z (include/glabc.h)
template <int D>
GLABC_DEV void dist_gamma_forward(const DistArgs<D>& g, uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t step, int j,
                                  float (&th)[D], float& lq)
{
    double t[D];
#pragma unroll 1
    for (int q = 0; q < D; ++q) {
        const double z = glabc_gamma_draw_candidate((double)g.p0[q], c0, c1, step, j, q, k0, k1) * (double)g.p2[q];
        const double lp = glabc_gamma_log_pdf((double)g.p0[q], (double)g.p2[q], (double)g.p3[q], z);
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (r == q) {
                th[r] = (float)z;
                t[r] = lp;
            }
        }
    }
    lq = (float)aten_rowsum_f64<D>(t);
}

template <int D, bool KNOWN_GAUSS_UNIT = false, bool GAMMA_OK = false>
GLABC_DEV float dist_log_prob(const DistArgs<D>& g, const float (&z)[D])
{
    if constexpr (GAMMA_OK) {
        if (g.kind == GLABC_DIST_GAMMA) return dist_log_prob_gamma<D>(g, z);
    }
    if (KNOWN_GAUSS_UNIT || g.kind == GLABC_DIST_DIAG_GAUSS) {
        float t[D];
        if (KNOWN_GAUSS_UNIT || g.unit_scale) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float e = z[j] - g.p0[j];                    // == (z - loc) / 1.0f, bit for bit
                t[j] = 0.5f * (e * e);                       // == log_scale + 0.5 e^2 with log_scale = 0 (0.5 e^2 is never -0)
            }
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float e = (z[j] - g.p0[j]) / g.p2[j];
                t[j] = g.p1[j] + 0.5f * (e * e);
            }
        }
        return g.c0 - aten_rowsum<D>(t);
    } else {
        bool out = false;
#pragma unroll
        for (int j = 0; j < D; ++j) out = out || (z[j] < g.p0[j]) || (z[j] > g.p1[j]);
        return out ? -__builtin_inff() : g.c0;
    }
}

// log_p of forward() given its noise: DiagGaussian distribution.py:171-173, Uniform :78
template <int D, bool KNOWN_GAUSS_UNIT = false>
GLABC_DEV float dist_forward_log_p(const DistArgs<D>& g, const float (&noise)[D])
{
    if (KNOWN_GAUSS_UNIT || g.kind == GLABC_DIST_DIAG_GAUSS) {
        float t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float h = 0.5f * (noise[j] * noise[j]);
            t[j] = KNOWN_GAUSS_UNIT ? h : g.p1[j] + h;       // log_scale = 0 in the unit variant
        }
        return g.c0 - aten_rowsum<D>(t);
    }
    return g.c0;
}

// ---- the Model callbacks ---------------------------------------------------------------------
// tanh and x^k for the g-and-k quantile function, spelled with the specified exp / log
GLABC_DEV float gk_tanhf(float x)
{
    // exp of a finite non-positive argument: the clamp at -104 is all glabc_expf adds to its core there (same bits,
    // four instructions less, 40 times per step)
    const float e = glabc_expf_core(__builtin_fmaxf(-2.0f * __builtin_fabsf(x), -104.0f));
    const float r = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -r : r;
}

// ascending sort of N registers (compare-exchange network; odd-even transposition for small N)
template <int N>
GLABC_DEV void sort_ascending(float (&v)[N])
{
#pragma unroll
    for (int pass = 0; pass < N; ++pass) {
#pragma unroll
        for (int i = pass & 1; i + 1 < N; i += 2) {
            const float lo = __builtin_fminf(v[i], v[i + 1]), hi = __builtin_fmaxf(v[i], v[i + 1]);
            v[i] = lo;
            v[i + 1] = hi;
        }
    }
}

// Standard normals one simulation consumes: y_dim for the built-in simulators; a run-time compiled user simulator
// (glabc_rtc.hip defines GLABC_USER_SIM / GLABC_USER_NOISE_DIM before this header) declares its own count.
#ifdef GLABC_USER_SIM
template <int YD> struct NoiseDim { static constexpr int value = GLABC_USER_NOISE_DIM; };
#else
template <int YD> struct NoiseDim { static constexpr int value = YD; };
#endif

// generate_samples for one theta and one simulation.
//   ABS_GAUSS, examples/Mixture.py:19-23:  y = |theta| + (loc + scale*eps)
//   GK, examples/GK.py:  y_j = A + B (1 + c tanh(g z_j / 2)) (1 + z_j^2)^k z_j, then sorted (order statistics)
//   USER (run-time compiled builds only): glabc_user_simulate(theta, eps, y), the caller's C source
template <int D, int YD>
GLABC_DEV void model_simulate(const StepArgs<D, YD>& a, const float (&theta)[D], const float (&eps)[NoiseDim<YD>::value],
                              float (&y)[YD])
{
#ifdef GLABC_USER_SIM
    glabc_user_simulate(theta, eps, y);
    return;
#else
    if constexpr (YD == D) {
        if (D < 4 || a.sim_kind == GLABC_SIM_ABS_GAUSS) {        // D < 4: the g-and-k shape cannot occur, no run-time test
#pragma unroll
            for (int j = 0; j < YD; ++j) {
                float noise = a.noise_loc[j] + a.noise_scale[j] * eps[j];
                y[j] = __builtin_fabsf(theta[j]) + noise;
            }
            return;
        }
    }
    if constexpr (D >= 4) {
#pragma unroll
        for (int j = 0; j < YD; ++j) {
            const float z = eps[j];
            const float t = gk_tanhf((theta[2] * z) * 0.5f);
            // z is a Box-Muller normal (|z| < 7), so 1 + z^2 is a normal float in [1, 50]: the special-case-free log
            // returns glabc_logf's bits with 10 instructions less, 40 times per step
            const float pw = glabc_expf(theta[3] * glabc_logf_normal(1.0f + z * z));
            y[j] = theta[0] + ((theta[1] * (1.0f + a.gk_c * t)) * pw) * z;
        }
        sort_ascending<YD>(y);
    }
#endif
}

// calculate_log_kernel, Mixture.py:33-45: DiagGaussian(1, 0, log eps).log_prob(||y - y_obs||)
// LEAN_SQRT: the caller knows every |y_obs_j| >= 2^-6, so a difference y_j - y_obs_j is 0 or at least 2^-31 in
// magnitude and the sum of squares is 0 or >= 2^-62 -- the domain on which glabc_sqrtf_normal is the correctly
// rounded square root (8 instructions less than the general expansion, five times per step)
// Run-time compiled builds (glabc_rtc.hip) may replace each of the Model's other callbacks as well: the user's source announces
//   #define GLABC_USER_PRIOR 1        float glabc_user_prior_log_prob(const float* theta)                 Mixture.py:28-31
//   #define GLABC_USER_DISCREPANCY 1  float glabc_user_discrepancy(const float* y, const float* y_obs)    Mixture.py:33-36
//   #define GLABC_USER_KERNEL 1       float glabc_user_log_kernel(float dis, float scale)                 Mixture.py:38-45
// (scale = the float32 kernel width of the descriptor, exp(log(epsilon))); whatever is not announced stays the descriptor's.
template <int YD>
GLABC_DEV float model_discrepancy_rows(const float (&y)[YD], const float* y_obs)
{
#ifdef GLABC_USER_DISCREPANCY
    float yo[YD];
#pragma unroll
    for (int j = 0; j < YD; ++j) yo[j] = y_obs[j];
    return glabc_user_discrepancy(y, yo);
#else
    float t[YD];
#pragma unroll
    for (int j = 0; j < YD; ++j) {
        float d = y[j] - y_obs[j];
        t[j] = d * d;
    }
    return __builtin_sqrtf(aten_rowsum<YD>(t));
#endif
}

GLABC_DEV float model_log_kernel_of(float dis, float kern_scale, float kern_log_scale, float kern_c0)
{
#ifdef GLABC_USER_KERNEL
    (void)kern_log_scale;
    (void)kern_c0;
    return glabc_user_log_kernel(dis, kern_scale);
#else
    const float e = (dis - 0.0f) / kern_scale;
    return kern_c0 - (kern_log_scale + 0.5f * (e * e));
#endif
}

template <int D, int YD, bool LEAN_SQRT = false, bool FAST = false>
GLABC_DEV float model_log_kernel(const StepArgs<D, YD>& a, const float (&y)[YD])
{
#if defined(GLABC_USER_DISCREPANCY) || defined(GLABC_USER_KERNEL)
    return model_log_kernel_of(model_discrepancy_rows<YD>(y, a.y_obs), a.kern_scale, a.kern_log_scale, a.kern_c0);
#else
    float t[YD];
#pragma unroll
    for (int j = 0; j < YD; ++j) {
        float d = y[j] - a.y_obs[j];
        t[j] = d * d;
    }
    const float ss = aten_rowsum<YD>(t);
    if constexpr (FAST) {                                               // GLABC_MATH_FAST: v_sqrt_f32, v_rcp_f32
        const float e = __builtin_amdgcn_sqrtf(ss) * __builtin_amdgcn_rcpf(a.kern_scale);
        return a.kern_c0 - (a.kern_log_scale + 0.5f * (e * e));
    }
    float dis = LEAN_SQRT ? glabc_sqrtf_normal(ss) : __builtin_sqrtf(ss);
    float e;
    if constexpr (LEAN_SQRT) {
        // dis / kern_scale in three instructions instead of the twelve of an IEEE division: with r = RN(1/s) the
        // residual-corrected product is the correctly rounded quotient for EVERY significand of dis -- checked
        // exhaustively on the host for this s before the variant is chosen; dis is 0 or in [2^-31, 2^64] here and s
        // in [2^-20, 2^20], so nothing under- or overflows and the binade does not matter.  (dis = inf / nan gives nan
        // instead of inf: such a candidate has weight 0 and is never accepted either way.)
        const float q = dis * a.kern_rinv;
        e = __builtin_fmaf(__builtin_fmaf(-q, a.kern_scale, dis), a.kern_rinv, q);
    } else {
        e = (dis - 0.0f) / a.kern_scale;
    }
    return a.kern_c0 - (a.kern_log_scale + 0.5f * (e * e));
#endif
}

template <int D, int YD, bool GU, bool GM>
GLABC_DEV float model_prior(const StepArgs<D, YD>& a, const float (&theta)[D])
{
#ifdef GLABC_USER_PRIOR
    return glabc_user_prior_log_prob(theta);
#else
    return dist_log_prob<D, GU, GM>(a.prior, theta);
#endif
}

// ---- random draws of one (chain, step) ---------------------------------------------
struct Rng {
    uint32_t c0, c1, k0, k1;
};

// ---- GLABC_MATH_FAST (include/glabc.h): the hardware's transcendental instructions, device only ------------------------
// v_log_f32 is log2, v_exp_f32 is 2^x, v_sin_f32 / v_cos_f32 take their argument in revolutions (sin(2 pi x)), ~1 ulp each.
GLABC_DEV void fast_normal_pair(uint32_t a, uint32_t b, float* z0, float* z1)
{
    const float u1 = glabc_uniform_pos_f32(a);                          // (0, 1]
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), ln = ln 2 * log2
    const float u2 = glabc_uniform_f32(b);                              // [0, 1) revolutions
    *z0 = rad * __builtin_amdgcn_cosf(u2);
    *z1 = rad * __builtin_amdgcn_sinf(u2);
}
// exp: v_exp_f32 flushes results below 2^-126 to zero; a hopeless chain's iSIR weights are ALL that small (far from the
// observation every K(y) is a large negative number) and the reference still resamples among them (torch.exp underflows
// gradually), so the small range is scaled: 2^(t + 64) * 2^-64, the multiplication rounds into the denormals correctly.
GLABC_DEV float fast_expf(float x)
{
    const float t = x * 1.4426950408889634f;                            // exp(-inf) = 0, NaN stays NaN
    const bool tiny = t < -126.0f;
    const float r = __builtin_amdgcn_exp2f(tiny ? t + 64.0f : t);
    return tiny ? r * 0x1p-64f : r;
}
GLABC_DEV float fast_logf(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }       // log(0) = -inf

// ---- lane-group exchange -----------------------------------------------------------------
// Groups are 1, 2 or 4 adjacent lanes, i.e. they sit inside one DPP quad: a value held by
// group lane SRC (compile-time) reaches every lane of the group with one v_mov_b32_dpp
// quad_perm (a VALU op; a ds_bpermute costs ~10x the SIMD time -- tools/ubench).  A run-time
// source lane (the owner of the winning candidate, needed only when some chain of the
// wavefront moves) goes through ds_bpermute.  Every lane of the wave must execute these.
template <int L, int SRC>
GLABC_DEV int group_bcast_i(int v)
{
    if constexpr (L == 1) {
        return v;
    } else if constexpr (L == 2) {
        constexpr int ctrl = SRC | (SRC << 2) | ((2 + SRC) << 4) | ((2 + SRC) << 6);     // quad_perm:[s,s,2+s,2+s]
        return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
    } else {
        static_assert(L == 4, "lane groups are 1, 2 or 4 wide");
        constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);                 // quad_perm:[s,s,s,s]
        return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
    }
}

template <int L, int SRC>
GLABC_DEV float group_bcast(float v)
{
    return __builtin_bit_cast(float, group_bcast_i<L, SRC>(__builtin_bit_cast(int, v)));
}

template <int L>
GLABC_DEV float group_get_dyn(float v, int src)
{
    if constexpr (L == 1) {
        return v;
    } else {
        const int lane = (int)(threadIdx.x & 63u);
        return __shfl(v, (lane & ~(L - 1)) | src, 64);
    }
}

// weight of candidate K-1 (K = 1..N, compile-time) from its owner lane (K-1) % L, slot (K-1) / L
template <int L, int K, int NL>
GLABC_DEV float gather_weight(const float (&wl)[NL])
{
    return group_bcast<L, (K - 1) % L>(wl[(K - 1) / L]);
}

template <int L, int N, int NL, int K = 1>
GLABC_DEV void gather_weights(const float (&wl)[NL], float (&w)[N + 1])
{
    if constexpr (K <= N) {
        w[K] = gather_weight<L, K, NL>(wl);
        gather_weights<L, N, NL, K + 1>(wl, w);
    }
}

// ---- chain state in registers ---------------------------------------------------------
template <int D, int YD = D>
struct Chain {
    float theta[D];
    float y[YD];
    float prior;      // prior_log_prob(theta)         (cache of a pure function of the state)
    float kern;       // calculate_log_kernel(y)       (cache)
    float q;          // global/importance log_prob(theta)  (cache)
    float log_w;      // log_weight_old, GLMCMC.py:53-55 -- stale after an accepted local move until the next global step
    float lw_cur;     // (prior + kern) - q of the CURRENT state: what GLMCMC.py:60-64 will assign to log_weight_old (cache)
    float w_cur;      // exp(lw_cur), NaN -> 0: the current state's iSIR weight, GLMCMC.py:75-81 (cache)
    float gf;         // this chain's global_frequency
    uint32_t flags;
    uint32_t n_moves;
};

template <int D, int YD, bool GAMMA_OK = false>
GLABC_DEV void refresh_cache(const StepArgs<D, YD>& a, Chain<D, YD>& c)
{
    c.prior = model_prior<D, YD, false, GAMMA_OK>(a, c.theta);
    c.kern = model_log_kernel<D, YD>(a, c.y);
    c.q = dist_log_prob<D, false, GAMMA_OK>(a.global, c.theta);
}

// One iteration of GLMCMC (GLMCMC.py:58-104) or GlobalMCMC (GlobalMCMC.py:37-68) for the
// chain owned by this lane group; `sub` is this lane's index in the group.
//
// Candidate j of the iteration (j = 0..N-1) is evaluated by lane j % L in its slot j / L:
//   global move : theta' = loc + scale*eps               GLMCMC.py:66 / GlobalMCMC.py:40
//   local move  : theta' = (loc + scale*eps) + theta     GLMCMC.py:91 / GlobalMCMC.py:56  (candidate 0 only)
//   y' = simulate(theta')                                GLMCMC.py:71,94
//   prior', K'                                           GLMCMC.py:72-74,96
// then
//   iSIR        : w = exp(cat(lw_old, (prior'+K') - q')), NaN -> 0, w /= sum(w), inverse-CDF index in
//                 double against a double uniform         GLMCMC.py:75-84, 7-22
//   MH          : log(u) < ((prior'+K') - prior) - K                       GLMCMC.py:96-99
//                 log(u) < ((((prior'+K') + q) - q') - prior) - K          GlobalMCMC.py:44-47
template <int ALGO, int D, int YD, int N, int L, int VAR>
GLABC_DEV bool chain_step(const StepArgs<D, YD>& a, const Rng& rng, uint32_t step, int sub, Chain<D, YD>& c, int64_t tape_pos)
{
    constexpr bool GU = (VAR == VAR_GAUSS_UNIT);
    constexpr bool TAPE = (VAR == VAR_TAPE);
    constexpr bool GM = (VAR == VAR_GAMMA);
    static_assert(!TAPE || L == 1, "tape replay runs one lane per chain");
    static_assert(!GM || L == 1, "the Gamma variant runs one lane per chain");
    constexpr int NL = (N + L - 1) / L;            // candidate slots per lane
    constexpr int HEAD = L - 1;                    // the lane with the fewest candidates draws the step head
    // When N is not a multiple of L the last slot of lane L-1 holds no candidate: the step head
    // (Philox slot 0) is drawn there, in the same instruction stream as the other lanes'
    // candidates of that pass, and that pass runs first so candidate 0 knows its branch.
    constexpr bool FREE_SLOT = (L > 1) && (N % L != 0) && (NL >= 2);
    // Draws per candidate: D proposal draws (words 0..D-1), then YD simulator normals starting at the next EVEN word
    // index DP: normals come in Box-Muller pairs from words (2i, 2i+1), and a Uniform proposal turns each of its words into
    // one [0,1) draw -- a simulator normal must never share a word with a proposal draw (theta' and y' would not be
    // independent; with odd D and the pair straddling the boundary they were not)
    constexpr int DP = D + (D & 1);
    constexpr int ND = NoiseDim<YD>::value;            // simulator normals (= YD for the built-in simulators)
    constexpr int M = DP + ND;
    constexpr int SPP = (M + 3) / 4;

    uint32_t hw[4];
    float log_u = 0.0f;
    bool is_global = false;
    auto take_head = [&]() {
        if constexpr (L > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[q] = (uint32_t)group_bcast_i<L, HEAD>((int)hw[q]);
        }
        const float ub = TAPE ? a.tape_u[2 * tape_pos] : glabc_uniform_f32(hw[0]);
        const float ua = TAPE ? a.tape_u[2 * tape_pos + 1] : glabc_uniform_f32(hw[1]);
        // GLMCMC.py:98: log(u).  A Philox uniform is k 2^-24 -- zero or a normal float -- so the special cases of
        // glabc_logf reduce to one select (same bits); a tape may hold anything
        log_u = TAPE ? glabc_logf(ua) : ((ua == 0.0f) ? -__builtin_inff() : glabc_logf_normal(ua));
        is_global = ub < c.gf;                                                // GLMCMC.py:59 / GlobalMCMC.py:39
        if (ALGO == ALGO_GLMCMC && is_global) {
            if (c.flags & GLABC_FLAG_LOCAL) c.log_w = c.lw_cur;                   // GLMCMC.py:60-64
            c.flags &= ~GLABC_FLAG_LOCAL;                                         // GLMCMC.py:65
        }
    };
    if constexpr (!FREE_SLOT) {
        if (L == 1 || sub == HEAD) {
            glabc_u32x4 h = glabc_philox4x32_10(rng.c0, rng.c1, step, 0u, rng.k0, rng.k1);
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[q] = h.v[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[q] = 0u;
        }
        take_head();
    }

    // ---- this lane's candidates ----
    float th[NL][D], yy[NL][YD], lw[NL], pr[NL], kk[NL], wl[NL];
    bool acc_mh = false;
    const bool g_uni = !GU && a.global.kind == GLABC_DIST_UNIFORM;
    const bool l_uni = !GU && a.local.kind == GLABC_DIST_UNIFORM;
#pragma unroll
    for (int rr = 0; rr < NL; ++rr) {
        const int r = FREE_SLOT ? (NL - 1 - rr) : rr;
        const int j = sub + L * r;                  // candidate index of this slot (>= N: unused slot)
        const bool head_pass = FREE_SLOT && (r == NL - 1);
        const bool first = (r == 0) && (L == 1 || sub == 0);                  // candidate 0 doubles as the local move
        // Philox blocks of this slot: D proposal draws then (from word DP) YD simulator draws out of SPP
        // blocks at slots 1 + j*SPP + b; normals in Box-Muller pairs from words (2i, 2i+1)
        uint32_t w[4 * SPP];
#pragma unroll
        for (int b = 0; b < SPP; ++b) {
            uint32_t slot_id = (uint32_t)(1 + j * SPP + b);
            if (head_pass && b == 0) slot_id = (sub == HEAD) ? 0u : slot_id;
            glabc_u32x4 o = glabc_philox4x32_10(rng.c0, rng.c1, step, slot_id, rng.k0, rng.k1);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[4 * b + q] = o.v[q];
        }
        if (head_pass) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hw[q] = w[q];
            take_head();
        }
        const bool loc = first && !is_global;
        const bool uni = loc ? l_uni : g_uni;
        float nrm[2 * ((M + 1) / 2)], e[D], s[ND];
#pragma unroll
        for (int i = 0; 2 * i < M; ++i) glabc_normal_pair(w[2 * i], w[2 * i + 1], &nrm[2 * i], &nrm[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < D; ++i) e[i] = (!GU && uni) ? glabc_uniform_f32(w[i]) : nrm[i];
#pragma unroll
        for (int i = 0; i < ND; ++i) s[i] = nrm[DP + i];
        if constexpr (TAPE) {                       // the tape holds this candidate's draws in the same order
            const float* tz = a.tape_z + (tape_pos * a.tape_nprop + (j < a.tape_nprop ? j : 0)) * (D + YD);
#pragma unroll
            for (int i = 0; i < D; ++i) e[i] = tz[i];
#pragma unroll
            for (int i = 0; i < ND; ++i) s[i] = tz[D + i];
        }
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const float p0 = loc ? a.local.p0[q] : a.global.p0[q];
            const float p2 = loc ? a.local.p2[q] : a.global.p2[q];
            const float t = p0 + p2 * e[q];                                   // distribution.py:170 / :77
            th[r][q] = loc ? (t + c.theta[q]) : t;                            // GLMCMC.py:91
        }
        // a Gamma importance / global proposal (wave-uniform): the candidate and forward()'s log q from the chain's Gamma
        // slots; the lanes on the local branch keep candidate 0 as built above
        float lq_gamma = 0.0f;
        const bool g_gam = GM && a.global.kind == GLABC_DIST_GAMMA;
        if constexpr (GM) {
            if (g_gam) {
                float tg[D];
                dist_gamma_forward<D>(a.global, rng.c0, rng.c1, rng.k0, rng.k1, step, j, tg, lq_gamma);
#pragma unroll
                for (int q = 0; q < D; ++q) th[r][q] = loc ? th[r][q] : tg[q];
            }
        }
        // log q of the proposal under the global distribution: from its noise (forward(), GLMCMC.py:66) for an iSIR
        // candidate; for GLMCMC's local move q(theta') itself, so that lw / wl of slot 0 are the log-weight and weight
        // the proposed state will carry if it is accepted (what GLMCMC.py:60-64 computes at the next global step)
        float lq;
        if constexpr (ALGO == ALGO_GLMCMC && GU) {
            float v[D];                                                       // both are c0 - sum 0.5 v^2 in the unit variant
#pragma unroll
            for (int q = 0; q < D; ++q) v[q] = loc ? (th[r][q] - a.global.p0[q]) : e[q];
            lq = dist_forward_log_p<D, GU>(a.global, v);
        } else if constexpr (ALGO == ALGO_GLMCMC) {
            lq = loc ? dist_log_prob<D, GU, GM>(a.global, th[r]) : (g_gam ? lq_gamma : dist_forward_log_p<D, GU>(a.global, e));
        } else {
            lq = g_gam ? lq_gamma : dist_forward_log_p<D, GU>(a.global, e);   // unused by the local move
        }
        model_simulate<D, YD>(a, th[r], s, yy[r]);
        pr[r] = model_prior<D, YD, GU, GM>(a, th[r]);
        kk[r] = model_log_kernel<D, YD, GU>(a, yy[r]);
        const float pk = pr[r] + kk[r];
        lw[r] = pk - lq;                                                      // GLMCMC.py:74
        if (r == 0) {
            float log_acc;
            if (ALGO == ALGO_GLOBAL)
                log_acc = loc ? ((pk - c.prior) - c.kern)                     // GlobalMCMC.py:60-61
                              : ((((pk + c.q) - lq) - c.prior) - c.kern);     // GlobalMCMC.py:44-46
            else
                log_acc = (pk - c.prior) - c.kern;                            // GLMCMC.py:96-97
            acc_mh = log_u < log_acc;                                         // GLMCMC.py:98-99
        }
        if (ALGO == ALGO_GLMCMC) {
            const float v = glabc_expf(lw[r]);                                // GLMCMC.py:78
            wl[r] = (v != v) ? 0.0f : v;                                      // GLMCMC.py:80-81
        }
    }

    // ---- winner index: 0 = stay, k = candidate k-1 ----
    int ind;
    if constexpr (ALGO == ALGO_GLMCMC) {
        float w[N + 1];
        w[0] = c.w_cur;                                                       // exp(log_weight_old), GLMCMC.py:75-81
        gather_weights<L, N, NL>(wl, w);
        const float tot = aten_rowsum<N + 1>(w);                              // GLMCMC.py:82
        const double u_res = TAPE ? a.tape_r[tape_pos] : glabc_uniform_f64(hw[2], hw[3]);
        // weight_sampling, GLMCMC.py:17-22: first k with u < sum_{j<=k} (double)(w_j / tot).  Fast pass in float32:
        // w_j * rcp(tot) is within 2.4e-7 relative of the float32 quotient (1 ulp of v_rcp_f32 + one rounding) and a
        // float32 running sum of <= 17 such terms adds <= 1e-6, so the partial sums are within 1.3e-6 of the reference's
        // and the index can differ only if u lies that close to one of them; lanes where |u - partial sum| <= 4e-6
        // somewhere (or anything is NaN / inf) redo it the reference's way: IEEE divisions, double sums.
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
            // the reciprocal is only trusted where it is accurate: a denormal / huge / zero / non-finite total shows
            // up as fast weights that do not sum to one
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
        ig = ig < 0 ? 0 : ig;                                                 // None -> stay, GLMCMC.py:84
        const int il = group_bcast_i<L, 0>(acc_mh ? 1 : 0);
        ind = is_global ? ig : il;
    } else {
        ind = group_bcast_i<L, 0>(acc_mh ? 1 : 0);
    }

    // ---- move: fetch the winning candidate from its owner lane ----
    const bool moved = ind > 0;
    if (__any(moved)) {
        const int owner = (ind - 1) & (L - 1);
        const int slot = (ind - 1) / L;
        // this lane's candidate in the winning slot (conditional moves over the unrolled slots keep
        // everything in VGPRs; a run-time array index would be promoted to LDS / scratch)
        float nt[D], ny[YD], nlw = lw[0], npr = pr[0], nkk = kk[0], nw = (ALGO == ALGO_GLMCMC) ? wl[0] : 0.0f;
#pragma unroll
        for (int q = 0; q < D; ++q) nt[q] = th[0][q];
#pragma unroll
        for (int q = 0; q < YD; ++q) ny[q] = yy[0][q];
#pragma unroll
        for (int r = 1; r < NL; ++r) {
            if (slot == r) {                      // (measured: these small masked blocks beat per-value selects)
#pragma unroll
                for (int q = 0; q < D; ++q) nt[q] = th[r][q];
#pragma unroll
                for (int q = 0; q < YD; ++q) ny[q] = yy[r][q];
                nlw = lw[r];
                npr = pr[r];
                nkk = kk[r];
                if (ALGO == ALGO_GLMCMC) nw = wl[r];
            }
        }
#pragma unroll
        for (int q = 0; q < D; ++q) nt[q] = group_get_dyn<L>(nt[q], owner);
#pragma unroll
        for (int q = 0; q < YD; ++q) ny[q] = group_get_dyn<L>(ny[q], owner);
        nlw = group_get_dyn<L>(nlw, owner);
        npr = group_get_dyn<L>(npr, owner);
        nkk = group_get_dyn<L>(nkk, owner);
        if (ALGO == ALGO_GLMCMC) nw = group_get_dyn<L>(nw, owner);
        if (moved) {
#pragma unroll
            for (int q = 0; q < D; ++q) c.theta[q] = nt[q];
#pragma unroll
            for (int q = 0; q < YD; ++q) c.y[q] = ny[q];
            c.prior = npr;
            c.kern = nkk;
            c.q = dist_log_prob<D, GU, GM>(a.global, c.theta);
            if (ALGO == ALGO_GLMCMC) {
                c.lw_cur = nlw;
                c.w_cur = nw;
                if (is_global)
                    c.log_w = nlw;                                            // GLMCMC.py:86
                else
                    c.flags |= GLABC_FLAG_LOCAL;                              // GLMCMC.py:100
            }
        }
    }
    return moved;
}

}  // namespace glabc
