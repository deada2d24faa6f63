/*
 * glabc_numerics.h -- the numerical specification shared by every build of the
 * GL-ABC-MCMC hot path: the counter-based random stream (Philox4x32-10), the
 * u32 -> uniform / normal conversions, and the f32 elementary functions
 * (exp, log, sin/cos of 2*pi*u) used inside a chain step.
 *
 * Why this is a header and not "whatever libm does": the accept / resample
 * decisions of a chain must be bit-identical between the gfx950 kernel and the
 * CPU checker (oracle/), so both sides evaluate the SAME sequence of IEEE-754
 * operations: + - * / sqrt and fma, all correctly rounded on x86-64 and on
 * CDNA4.  No hardware transcendental (v_exp_f32, v_log_f32, v_sin_f32) and no
 * libm call appears below.  Every translation unit that includes this file
 * must be compiled with  -ffp-contract=off  (contraction is spelled explicitly
 * with fmaf where it is wanted).
 *
 * The reference (caofff/GL-ABC-MCMC) draws from torch's and NumPy's global
 * Mersenne twisters (GLMCMC.py:17,59,66; distribution.py:167); a batched GPU
 * sampler cannot share that stream, so the stream is re-specified here and
 * parity with the reference is established on replayed random tapes instead
 * (tests/golden/).
 *
 * Plain C99; also valid C++ and HIP device code.
 */
#ifndef GLABC_NUMERICS_H
#define GLABC_NUMERICS_H

#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif

#if defined(__HIPCC__)
#define GLABC_HD static __host__ __device__ __forceinline__
#else
#define GLABC_HD static inline
#endif

/* the 10 Philox rounds are straight-line code: independent calls then interleave (ILP) */
#if defined(__clang__)
#define GLABC_UNROLL_10 _Pragma("unroll")
#elif defined(__GNUC__)
#define GLABC_UNROLL_10 _Pragma("GCC unroll 10")
#else
#define GLABC_UNROLL_10
#endif

/* ---- bit casts ---------------------------------------------------------- */
GLABC_HD uint32_t glabc_f2u(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
GLABC_HD float glabc_u2f(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }
GLABC_HD uint64_t glabc_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
GLABC_HD double glabc_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 constants) ----------
 * Stream layout used by every sampler:
 *   key     = (seed_lo, seed_hi)                 -- uniform over a launch
 *   counter = (chain_lo, chain_hi, step, slot)   -- per chain, per step, per draw
 * so a chain's numbers depend only on (seed, global chain id, step, slot): not
 * on launch geometry, steps-per-launch, or how chains are sharded over GPUs. */
typedef struct { uint32_t v[4]; } glabc_u32x4;

#define GLABC_PHILOX_M0 0xD2511F53u
#define GLABC_PHILOX_M1 0xCD9E8D57u
#define GLABC_PHILOX_W0 0x9E3779B9u
#define GLABC_PHILOX_W1 0xBB67AE85u

GLABC_HD glabc_u32x4 glabc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1)
{
GLABC_UNROLL_10
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)GLABC_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)GLABC_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += GLABC_PHILOX_W0;
        k1 += GLABC_PHILOX_W1;
    }
    glabc_u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

/* ---- u32 -> uniform ------------------------------------------------------ */
/* f32 uniform on [0,1): 24 random bits, the same grid torch.rand(float32) uses. */
GLABC_HD float glabc_uniform_f32(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }
/* f32 uniform on (0,1]: all 32 bits, never 0 (argument of log in Box-Muller). */
GLABC_HD float glabc_uniform_pos_f32(uint32_t x) { return __builtin_fmaf((float)x, 0x1p-32f, 0x1p-33f); }
/* f64 uniform on [0,1): 53 random bits, the grid numpy.random.uniform uses
 * (the reference's resampling draw is a NumPy double, GLMCMC.py:17). */
GLABC_HD double glabc_uniform_f64(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}

/* ---- f32 log ---------------------------------------------------------------
 * x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log m = f - f^2/2 + f^3 P(f), f = m-1.
 * Max error < 1 ulp (measured exhaustively on (0,1] and sampled elsewhere:
 * tests/test_numerics.py).  glabc_logf_normal is the branch-free core for positive
 * NORMAL finite x; glabc_logf adds the IEEE special cases and subnormals around it
 * and returns the same bits wherever both apply. */
GLABC_HD float glabc_logf_core(uint32_t ix, float escale)
{
    ix += 0x3f800000u - 0x3f3504f3u;
    float e = (float)((int32_t)(ix >> 23) - 127) + escale;
    float m = glabc_u2f((ix & 0x007fffffu) + 0x3f3504f3u);
    float f = m - 1.0f;
    float p = -0x1.2d9544p-4f;
    p = __builtin_fmaf(p, f, 0x1.0276dcp-3f);
    p = __builtin_fmaf(p, f, -0x1.0e610cp-3f);
    p = __builtin_fmaf(p, f, 0x1.235f0ep-3f);
    p = __builtin_fmaf(p, f, -0x1.5467dap-3f);
    p = __builtin_fmaf(p, f, 0x1.9998bp-3f);
    p = __builtin_fmaf(p, f, -0x1.00023cp-2f);
    p = __builtin_fmaf(p, f, 0x1.555564p-2f);
    float f2 = f * f;
    float r = __builtin_fmaf(p * f, f2, e * 0x1.7f7d1cp-20f);   /* f^3 P + e*ln2_lo */
    r = __builtin_fmaf(-0.5f, f2, r);
    r = r + f;
    return __builtin_fmaf(e, 0x1.62e4p-1f, r);                  /* + e*ln2_hi (ln2_hi has 9 trailing zero bits: e*ln2_hi exact) */
}

GLABC_HD float glabc_logf_normal(float x) { return glabc_logf_core(glabc_f2u(x), 0.0f); }

GLABC_HD float glabc_logf(float x)
{
    /* branch-free: the core runs on whatever bits arrive (integer and IEEE arithmetic only) and the special cases
     * are selected afterwards, in the order  +-0 -> -inf,  sign bit -> nan,  +inf / +nan -> x */
    const uint32_t ix = glabc_f2u(x);
    const int sub = ix < 0x00800000u;                                    /* +0 or positive subnormal */
    const uint32_t jx = sub ? glabc_f2u(x * 0x1p23f) : ix;
    float r = glabc_logf_core(jx, sub ? -23.0f : 0.0f);
    r = (ix >= 0x7f800000u) ? x : r;                                     /* +inf, nan */
    r = (ix >> 31) ? __builtin_nanf("") : r;                             /* log(<0) = nan (nan with sign bit too) */
    r = ((ix << 1) == 0u) ? -__builtin_inff() : r;                       /* log(+-0) = -inf */
    return r;
}

/* ---- f32 sqrt of x that is +-0 or finite with x >= 2^-64 ------------------------
 * Correctly rounded (== IEEE sqrtf, which is what the host build calls).  On gfx950 the
 * compiler's sqrtf expansion spends a third of its instructions on the rescaling of tiny
 * arguments (whose residuals would underflow) and on class checks; Box-Muller's argument is
 * 0 or in [1e-7, 46], so the device build keeps only the hardware estimate and the one-ulp
 * correction step.  Checked against the exactly rounded (float)sqrt((double)x) for EVERY
 * float in [2^-64, FLT_MAX] and for +-0 on the GPU (tests/test_hip_numerics.py). */
GLABC_HD float glabc_sqrtf_normal(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = glabc_u2f(glabc_f2u(s) - 1u);
    float s_up = glabc_u2f(glabc_f2u(s) + 1u);
    float r_dn = __builtin_fmaf(-s_dn, s, x);
    float r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
#else
    return __builtin_sqrtf(x);
#endif
}

/* ---- f32 exp ---------------------------------------------------------------
 * x = k ln2 + r, |r| <= ln2/2; exp r = 1 + r + r^2 Q(r); result scaled by 2^k in
 * two exact-or-once-rounded steps so the subnormal range rounds once. */
GLABC_HD float glabc_expf_core(float x)            /* x in [-104, 88.7228...] */
{
    float t = __builtin_fmaf(x, 0x1.715476p+0f, 12582912.0f);    /* round(x*log2e) in the low mantissa bits */
    float k = t - 12582912.0f;
    float r = __builtin_fmaf(k, -0x1.62e4p-1f, x);
    r = __builtin_fmaf(k, -0x1.7f7d1cp-20f, r);
    float q = 0x1.687bf6p-10f;
    q = __builtin_fmaf(q, r, 0x1.123b9ep-7f);
    q = __builtin_fmaf(q, r, 0x1.555b58p-5f);
    q = __builtin_fmaf(q, r, 0x1.55548ep-3f);
    q = __builtin_fmaf(q, r, 0x1.fffff8p-2f);
    float p = __builtin_fmaf(q * r, r, r) + 1.0f;
    int32_t ki = (int32_t)k;                            /* |k| <= 151 */
    int32_t k1 = ki / 2, k2 = ki - k1;
    float s1 = glabc_u2f((uint32_t)(k1 + 127) << 23);
    float s2 = glabc_u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

GLABC_HD float glabc_expf(float x0)
{
    /* Branch-free: the polynomial runs on the argument clamped to [-104, 88.7228...] (everything at or below -103.98
     * already gives 0) and +inf above the range / NaN are selected at the end -- six of these sit in every chain step,
     * and early returns would cut the step into small scheduling regions. */
    float x = __builtin_fminf(__builtin_fmaxf(x0, -104.0f), 88.72283935546875f);      /* NaN -> -104 */
    float e = glabc_expf_core(x);                       /* exp(-104) = 0 already: only overflow and NaN need a select */
    e = x0 > 88.72283935546875f ? __builtin_inff() : e;
    return x0 != x0 ? x0 : e;
}

/* The same function with early returns instead of selects (same bits for every input, tests/test_numerics.py): the
 * form for code whose registers are scarcer than its branches (the MFMA kernel of glabc_nf.hip). */
GLABC_HD float glabc_expf_b(float x)
{
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -104.0f) return 0.0f;
    return glabc_expf_core(x);
}

/* ---- sin, cos of 2*pi*u for u in [0,1) ---------------------------------------
 * Quadrant reduction is exact (t = 4u, q = round(t), f = t-q in [-1/2,1/2]);
 * then a = f*pi/2 and the usual degree-7 / degree-8 kernels on [-pi/4, pi/4]. */
GLABC_HD void glabc_sincos2pi(float u, float* s_out, float* c_out)
{
    float t = u * 4.0f;
    float qf = (t + 12582912.0f) - 12582912.0f;
    float f = t - qf;
    int32_t q = (int32_t)qf;
    float a = f * 0x1.921fb6p+0f;
    float z = a * a;
    float sp = -0x1.993c46p-13f;
    sp = __builtin_fmaf(sp, z, 0x1.11072p-7f);
    sp = __builtin_fmaf(sp, z, -0x1.555544p-3f);
    float s = __builtin_fmaf(sp * z, a, a);
    float cp = 0x1.99f6ep-16f;
    cp = __builtin_fmaf(cp, z, -0x1.6c0c5ep-10f);
    cp = __builtin_fmaf(cp, z, 0x1.55554ap-5f);
    float c = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    float ss = (q & 1) ? c : s;
    float cc = (q & 1) ? s : c;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *s_out = ss;
    *c_out = cc;
}

/* ---- two standard normals from two u32 (Box-Muller) ------------------------- */
GLABC_HD void glabc_normal_pair(uint32_t a, uint32_t b, float* z0, float* z1)
{
    float u1 = glabc_uniform_pos_f32(a);
    float rad = glabc_sqrtf_normal(-2.0f * glabc_logf_normal(u1));   /* u1 in [2^-33, 1]: argument is 0 or in [1e-7, 46] */
    float s, c;
    glabc_sincos2pi(glabc_uniform_f32(b), &s, &c);
    *z0 = rad * c;
    *z1 = rad * s;
}

/* ---- f64 log / exp (GLMALA only: its gradient estimate and, once a chain has switched to
 * float64, its iSIR weights are float64 in the reference, GLMALA.py:70-95,166-169) --------------
 * The classic fdlibm algorithms (Sun Microsystems, freely distributable) restated with their
 * published constants; < 1 ulp.  Plain IEEE double + - * / only, so host and device agree
 * bit for bit under -ffp-contract=off. */
#define GLABC_LOG_2PI 1.8378770664093453      /* numpy.log(2*numpy.pi), distribution.py:171 in float64 */

GLABC_HD double glabc_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix = glabc_d2u(x);
    int64_t k = 0;
    if ((ix << 1) == 0) return -__builtin_inf();                /* log(+-0) */
    if (ix >> 63) return __builtin_nan("");                      /* log(<0), nan with sign */
    if (ix >= 0x7ff0000000000000ull) return x;                   /* +inf, nan */
    if (ix < 0x0010000000000000ull) { x = x * 0x1p54; ix = glabc_d2u(x); k = -54; }
    /* normalise so that 1+f is in [sqrt(2)/2, sqrt(2)) */
    uint64_t hx = ix + (0x3ff0000000000000ull - 0x3fe6a09e667f3bcdull);
    k += (int64_t)(hx >> 52) - 1023;
    ix = (hx & 0x000fffffffffffffull) + 0x3fe6a09e667f3bcdull;
    double f = glabc_u2d(ix) - 1.0;
    double s = f / (2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

GLABC_HD double glabc_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_inf();
    if (x < -745.13321910194110842) return 0.0;
    double kf = invln2 * x + (x < 0.0 ? -0.5 : 0.5);
    int32_t k = (int32_t)kf;                                     /* truncation toward zero */
    double dk = (double)k;
    double hi = x - dk * ln2HI, lo = dk * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two steps (k in [-1075, 1024]) so the subnormal range rounds once */
    int32_t k1 = k / 2, k2 = k - k1;
    double s1 = glabc_u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = glabc_u2d((uint64_t)(k2 + 1023) << 52);
    return (y * s1) * s2;
}

/* ---- Gamma(shape, 1) variate in double (distribution.py:106-121 draws scipy.stats.gamma.rvs from NumPy's generator) ------
 * Marsaglia & Tsang (2000): d = a - 1/3, c = 1/sqrt(9 d); x ~ N(0,1), v = (1 + c x)^3 > 0, accept if
 * u < 1 - 0.0331 x^4 or log u < x^2/2 + d (1 - v + log v); result d v.  For shape < 1: Gamma(shape + 1) U^(1/shape).
 * The normal comes from Marsaglia's polar method (log, sqrt and division only -- no trigonometric function to specify in
 * double).  One Philox block per attempt, counter (c0, c1, c2, attempt): words 0, 1 -> the point in the square, words 2, 3
 * -> the 53-bit accept uniform; attempt 0xffffffff feeds the shape < 1 boost.  Acceptance is > 75 % per attempt; after 256
 * rejected attempts (probability < 1e-150) the mode-like value d is returned so that every lane terminates.
 * Plain IEEE double operations: host and device agree bit for bit under -ffp-contract=off. */
/* The general form: attempt t reads Philox(c0, c1, c2, slot0 + t), the shape < 1 boost Philox(c0, c1, c2, boost_slot). */
GLABC_HD double glabc_gamma_draw_at(double shape, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t slot0, uint32_t boost_slot,
                                    uint32_t k0, uint32_t k1)
{
    const double a = shape < 1.0 ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / __builtin_sqrt(9.0 * d);
    double g = d;
    for (uint32_t t = 0; t < 256u; ++t) {
        const glabc_u32x4 w = glabc_philox4x32_10(c0, c1, c2, slot0 + t, k0, k1);
        const double u1 = ((double)w.v[0] + 0.5) * 0x1p-31 - 1.0, u2 = ((double)w.v[1] + 0.5) * 0x1p-31 - 1.0;
        const double s = u1 * u1 + u2 * u2;
        if (!(s < 1.0) || s == 0.0) continue;
        const double x = u1 * __builtin_sqrt(-2.0 * glabc_log(s) / s);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = glabc_uniform_f64(w.v[2], w.v[3]);
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2) || glabc_log(u) < 0.5 * x2 + d * ((1.0 - v) + glabc_log(v))) {
            g = d * v;
            break;
        }
    }
    if (shape < 1.0) {
        const glabc_u32x4 w = glabc_philox4x32_10(c0, c1, c2, boost_slot, k0, k1);
        const double u = (((double)(w.v[0] >> 5) * 67108864.0 + (double)(w.v[1] >> 6)) + 0.5) * 0x1p-53;       /* in (0, 1) */
        g = g * glabc_exp(glabc_log(u) / shape);
    }
    return g;
}

/* glabc_gamma_forward (include/glabc.h): counter (row id lo, hi, coordinate, attempt), boost in attempt 0xffffffff */
GLABC_HD double glabc_gamma_draw(double shape, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1)
{
    return glabc_gamma_draw_at(shape, c0, c1, c2, 0u, 0xffffffffu, k0, k1);
}

/* Gamma as the importance / global proposal INSIDE a sampler iteration (glabc_dist kind GLABC_DIST_GAMMA): coordinate q of
 * candidate j of (chain, iteration) draws from counter (chain id lo, hi, iteration, GLABC_SLOT_GAMMA + ((j*8 + q) << 9) + attempt),
 * attempt < 256, the shape < 1 boost from attempt 256 -- a region of the slot word that neither the candidates' own blocks
 * (1 + j*spp + b < 2^15), GLMALA's gradient noise (< 2^22) nor the local move's redraws (GLABC_SLOT_REDRAW = 2^30) reach. */
#define GLABC_SLOT_GAMMA 0x20000000u
GLABC_HD double glabc_gamma_draw_candidate(double shape, uint32_t c0, uint32_t c1, uint32_t step, int j, int q, uint32_t k0,
                                           uint32_t k1)
{
    const uint32_t slot0 = GLABC_SLOT_GAMMA + (((uint32_t)j * 8u + (uint32_t)q) << 9);
    return glabc_gamma_draw_at(shape, c0, c1, step, slot0, slot0 + 256u, k0, k1);
}

/* one coordinate of Gamma.log_prob, distribution.py:133-136, float64: log(scipy.stats.gamma.pdf(z, shape, scale = 1/rate)) with
 * -inf where the pdf is 0 (it underflows earlier than a logpdf would -- reproduced);
 *   pdf = exp(xlogy(shape - 1, x) - x - gammaln(shape)) / scale,  x = z / scale,  0 for x < 0 */
GLABC_HD double glabc_gamma_log_pdf(double shape, double scale, double gammaln, double z)
{
    const double x = z / scale;
    if (x >= 0.0) {                                        /* scipy's support of gamma is closed at 0 */
        const double am1 = shape - 1.0;
        const double xl = am1 == 0.0 ? 0.0 : am1 * glabc_log(x);                       /* scipy.special.xlogy */
        const double p = glabc_exp((xl - x) - gammaln) / scale;                        /* gamma.pdf */
        return p > 0.0 ? glabc_log(p) : -__builtin_inf();                              /* distribution.py:136 */
    }
    return -__builtin_inf();
}

/* ---- exact, order-independent sums for GLMALA's gradient statistics (GLMALA.py:86-89) ------------------------
 * The mean and variance of the num_grad simulated discrepancies are accumulated on data shifted by a centre c
 * in FIXED POINT: d = x - c is an exact double (x, c are float32 values), q = rint(d * 2^40) is an exact integer for
 * every x, c in [2^-17, 2^7] (and a deterministic rounding outside), sum(q) fits int64 and sum(q^2) a 128-bit
 * integer for up to 65 536 terms.  Integer addition is associative, so any split of the simulations over lanes,
 * waves or threads gives the same bits -- which is what lets a wavefront share one chain's 2*d*num_grad
 * simulations.  (torch.mean / torch.var use their own float64 cascades; results agree to ~1e-16 relative.) */
typedef struct { int64_t s1; uint64_t s2_lo, s2_hi; } glabc_fxsum;

/* rint(d * 2^40) as an integer, for |d| < 2^11: adding 1.5 * 2^52 leaves the rounded-to-nearest-even integer in the low
 * mantissa bits (one fused multiply-add and one integer subtraction; a double -> int64 conversion is a dozen
 * instructions on gfx950).  Identical to (int64_t)rint(d * 0x1p40) on that domain (tests/test_numerics.py). */
GLABC_HD int64_t glabc_fx_quantize(double d)
{
    const double t = __builtin_fma(d, 0x1p40, 0x1.8p52);
    return (int64_t)glabc_d2u(t) - (int64_t)0x4338000000000000ll;
}

GLABC_HD void glabc_fx_add(glabc_fxsum* a, int64_t q)
{
    a->s1 += q;
    const uint64_t m = (uint64_t)(q < 0 ? -q : q);
    const uint64_t m_lo = m & 0xffffffffu, m_hi = m >> 32;
    /* m*m as a 128-bit number from 32-bit limbs */
    const uint64_t ll = m_lo * m_lo, lh = m_lo * m_hi, hh = m_hi * m_hi;
    const uint64_t mid = lh << 1;                       /* 2*lo*hi, bits 32.. ; lh < 2^47 so no overflow */
    uint64_t lo = ll + (mid << 32);
    uint64_t hi = hh + (mid >> 32) + (lo < ll ? 1u : 0u);
    const uint64_t old = a->s2_lo;
    a->s2_lo = old + lo;
    a->s2_hi = a->s2_hi + hi + (a->s2_lo < old ? 1u : 0u);
}

GLABC_HD void glabc_fx_merge(glabc_fxsum* a, const glabc_fxsum* b)
{
    a->s1 += b->s1;
    const uint64_t old = a->s2_lo;
    a->s2_lo = old + b->s2_lo;
    a->s2_hi = a->s2_hi + b->s2_hi + (a->s2_lo < old ? 1u : 0u);
}

/* The same sums with the square split in three 64-bit accumulators -- q = h 2^24 + l, q^2 = h^2 2^48 + h l 2^25 + l^2 --
 * so that one term costs three 32x32->64 multiply-accumulates instead of a 128-bit square: the form the gfx950 gradient
 * loop uses.  Exact for |q| < 2^47 (|d| < 128) and up to 65 536 terms, like glabc_fxsum; glabc_fxs_finish returns the
 * identical glabc_fxsum (integers: nothing is rounded), which tests/test_numerics.py checks on random streams. */
typedef struct { int64_t s1; uint64_t a; int64_t b; uint64_t c; } glabc_fxsplit;

GLABC_HD void glabc_fxs_add(glabc_fxsplit* x, int64_t q)
{
    const int32_t h = (int32_t)(q >> 24);                  /* floor split: l is the unsigned low part */
    uint32_t l = (uint32_t)q & 0xffffffu;
#if defined(__HIP_DEVICE_COMPILE__)
    /* ROCm 7.2's gfx950 backend turns the square of a value it knows to be 24 bits wide into a 24-bit-multiply node, drops
     * the mask as implied by that node, and then selects the full 32-bit v_mad_u64_u32 on the UNMASKED register
     * (tools/ubench/fx_check.hip shows it).  Hiding the value's width keeps the mask. */
    __asm__("" : "+v"(l));
#endif
    x->s1 += q;
    x->a += (uint64_t)((int64_t)h * (int64_t)h);
    x->b += (int64_t)h * (int64_t)l;
    x->c += (uint64_t)l * (uint64_t)l;
}

GLABC_HD void glabc_fxs_merge(glabc_fxsplit* x, const glabc_fxsplit* y)
{
    x->s1 += y->s1;
    x->a += y->a;
    x->b += y->b;
    x->c += y->c;
}

GLABC_HD glabc_fxsum glabc_fxs_finish(const glabc_fxsplit* x)
{
    glabc_fxsum r;
    r.s1 = x->s1;
    /* 128-bit  a 2^48 + b 2^25 + c  (b signed; the total is a sum of squares, so non-negative) */
    uint64_t lo = x->c, hi = 0;
    const uint64_t a_lo = x->a << 48, a_hi = x->a >> 16;
    lo += a_lo;
    hi += a_hi + (lo < a_lo ? 1u : 0u);
    const uint64_t b_lo = (uint64_t)x->b << 25, b_hi = (uint64_t)(x->b >> 39);      /* sign-extended high part */
    const uint64_t old = lo;
    lo += b_lo;
    hi += b_hi + (lo < old ? 1u : 0u);
    r.s2_lo = lo;
    r.s2_hi = hi;
    return r;
}

/* sum(d) and sum(d^2) as doubles */
GLABC_HD double glabc_fx_sum1(const glabc_fxsum* a) { return (double)a->s1 * 0x1p-40; }
GLABC_HD double glabc_fx_sum2(const glabc_fxsum* a)
{
    return ((double)a->s2_hi * 18446744073709551616.0 + (double)a->s2_lo) * 0x1p-80;
}

#endif /* GLABC_NUMERICS_H */
