// KernelDensity on gfx950 -- the adaptive proposal of AGLMCMC (reference: kernel_density.py:4-177).
//
//   kde_fit_kernel        one workgroup: normalised weights, weighted std -> bandwidth, log-weights, integer weights
//   kde_log_prob_kernel   one wavefront per evaluation point, lanes stride over the centres; two passes (max, then
//                         sum of exp) so that the sum can be taken exactly in fixed point -> independent of the lane
//                         split, bit-identical to the scalar checker
//   kde_sample_kernel     one thread per draw: inverse-CDF index on the integer prefix sums + bandwidth * normal
//
// The O(points x centres) log_prob is the only heavy piece: per pair D subtract/divide/square, 4 adds and -- in the
// second pass -- one exp.  Centres are read coalesced (dimension-major) and are shared by every wavefront, so after the
// first touch they come from L2; HBM traffic is ~ (points + centres) * (D + 1) * 4 bytes, the kernel is VALU-bound.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/glabc.h"
#include "../../include/glabc_numerics.h"
#include "glabc_device.h"

using namespace glabc;

namespace {

constexpr float KDE_LOG_2PI_F = 1.8378770351409912f;      // torch.log(torch.tensor(2*torch.pi)) in float32, kernel_density.py:121

struct FitArgs {
    const float* x;
    const float* w_raw;
    int64_t n;
    float h;
    int has_bw;
    float bw[GLABC_MAX_DIM];
    float* weights;
    float* log_w;
    int64_t* wq;
    float* consts;
};

// float64 sum over the workgroup in a fixed order: thread t adds its strided terms in index order, then a binary tree
__device__ __forceinline__ double block_sum(double v, double* red)
{
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) red[t] = red[t] + red[t + s];
        __syncthreads();
    }
    return red[0];
}

template <int D>
__global__ void __launch_bounds__(256) kde_fit_kernel(const FitArgs a)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int64_t n = a.n;
    // weights = w_raw / w_raw.sum()   (kernel_density.py:83-87)
    float wsum32 = 0.0f;
    if (a.w_raw) {
        double p = 0.0;
        for (int64_t i = t; i < n; i += 256) p = p + (double)a.w_raw[i];
        wsum32 = (float)block_sum(p, red);
    }
    const float uniform_w = 1.0f / (float)n;
    for (int64_t i = t; i < n; i += 256) {
        const float w = a.w_raw ? a.w_raw[i] / wsum32 : uniform_w;
        a.weights[i] = w;
        a.log_w[i] = glabc_logf(w + 1e-10f);                                         // :125
        a.wq[i] = glabc_fx_quantize((double)w);
    }
    __syncthreads();
    float bw[D];
    if (a.has_bw) {
#pragma unroll
        for (int d = 0; d < D; ++d) bw[d] = a.bw[d];
    } else {
        // weighted_std(X, weights, unbiased=True), kernel_density.py:40-68
        double p = 0.0;
        for (int64_t i = t; i < n; i += 256) p = p + (double)a.weights[i];
        const float w2 = (float)block_sum(p, red);                                    // :54
        double q = 0.0;
        for (int64_t i = t; i < n; i += 256) {
            const float w = a.weights[i] / w2;
            q = q + (double)(w * w);
        }
        float corr = 1.0f - (float)block_sum(q, red);                                 // :64
        corr = corr < 1e-10f ? 1e-10f : corr;                                         // :65
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double m = 0.0;
            for (int64_t i = t; i < n; i += 256) m = m + (double)((a.weights[i] / w2) * a.x[d * n + i]);
            const float mean = (float)block_sum(m, red);                              // :57
            double v = 0.0;
            for (int64_t i = t; i < n; i += 256) {
                const float diff = a.x[d * n + i] - mean;                             // :60
                v = v + (double)((a.weights[i] / w2) * (diff * diff));
            }
            const float var = (float)block_sum(v, red) / corr;                        // :62-65
            bw[d] = a.h * __builtin_sqrtf(var);                                       // :67, :36
        }
    }
    if (t == 0) {
        float lb[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a.consts[d] = bw[d];
            lb[d] = glabc_logf(bw[d]);
        }
        a.consts[D] = aten_rowsum<D>(lb);                                             // :122
        a.consts[D + 1] = (float)(0.5 * D) * KDE_LOG_2PI_F;                           // :121
    }
}

template <int D>
struct KdeArgs {
    const float* x;
    const float* log_w;
    const int64_t* cum_q;
    int64_t n_samples;
    float bw[D];
    float sum_log_bw, c_2pi;
    const float* pts;
    int64_t n_points;
    float* out;
    uint32_t seed_lo, seed_hi;
    int64_t row0;
    // indexed log_prob (glabc_kde_log_prob_indexed): the points are pts[d*pts_stride + idx[i]], i < *n_dev, results to out[idx[i]]
    const int32_t* idx;
    const int32_t* n_dev;
    int64_t pts_stride;
};

// inv[d] = 1/bandwidth_d (one IEEE division per wavefront): the reference divides every difference by the bandwidth
// (kernel_density.py:117); multiplying by the correctly rounded reciprocal differs by <= 1 ulp per term -- far inside the
// float32 summation-order tolerance this path is pinned at -- and halves the instructions of the (points x centres) loop
template <int D>
GLABC_DEV float kde_log_term(const KdeArgs<D>& a, const float (&pt)[D], const float (&inv)[D], int64_t s)
{
    float t[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float e = (pt[d] - a.x[d * a.n_samples + s]) * inv[d];                  // kernel_density.py:117
        t[d] = e * e;
    }
    float lk = -0.5f * aten_rowsum<D>(t);                                             // :118
    lk = lk - a.c_2pi;                                                                // :121
    lk = lk - a.sum_log_bw;                                                           // :122
    return lk + a.log_w[s];                                                           // :125
}

template <int D>
__global__ void __launch_bounds__(256) kde_log_prob_kernel(const KdeArgs<D> a)
{
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= (a.n_dev ? (int64_t)*a.n_dev : a.n_points)) return;   // whole wavefront leaves together
    const int64_t p = a.idx ? (int64_t)a.idx[slot] : slot;
    float pt[D], inv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        pt[d] = a.pts[d * a.pts_stride + p];
        inv[d] = 1.0f / a.bw[d];
    }
    // pass 1: max and NaN flag (torch.logsumexp: amax, kernel_density.py:126)
    float m = -__builtin_inff();
    int nan = 0;
    for (int64_t s = lane; s < a.n_samples; s += 64) {
        const float lk = kde_log_term<D>(a, pt, inv, s);
        nan |= (lk != lk);
        m = (lk > m) ? lk : m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(m, off);
        nan |= __shfl_xor(nan, off);
        m = (o > m) ? o : m;
    }
    const float m0 = (__builtin_fabsf(m) == __builtin_inff()) ? 0.0f : m;
    float res;
    if (nan) {
        res = __builtin_nanf("");
    } else if (m == __builtin_inff()) {
        res = m;
    } else {
        // pass 2: exact fixed-point sum of exp(lk - max); every term is in [0, 1]
        int64_t acc = 0;
        for (int64_t s = lane; s < a.n_samples; s += 64) {
            // the argument is <= 0 and not NaN here: glabc_expf reduces to its core behind the clamp (same bits)
            const float e = glabc_expf_core(__builtin_fmaxf(kde_log_term<D>(a, pt, inv, s) - m0, -104.0f));
            acc += glabc_fx_quantize((double)e);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        const float sum = (float)((double)acc * 0x1p-40);
        res = glabc_logf(sum) + m0;
    }
    if (lane == 0) a.out[p] = res;
}

template <int D>
__global__ void __launch_bounds__(256) kde_sample_kernel(const KdeArgs<D> a)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_points) return;
    const uint64_t gid = (uint64_t)(a.row0 + r);
    constexpr int NB = (D + 2 + 3) / 4;
    float nrm[4 * NB];
    double u = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, (uint32_t)b, a.seed_lo, a.seed_hi);
        if (b == 0) {
            u = glabc_uniform_f64(w.v[0], w.v[1]);
            glabc_normal_pair(w.v[2], w.v[3], &nrm[0], &nrm[1]);
        } else {
            glabc_normal_pair(w.v[0], w.v[1], &nrm[4 * b - 2], &nrm[4 * b - 1]);
            glabc_normal_pair(w.v[2], w.v[3], &nrm[4 * b], &nrm[4 * b + 1]);
        }
    }
    const int64_t S = a.n_samples;
    const int64_t target = (int64_t)(u * (double)a.cum_q[S - 1]);
    int64_t lo = 0, hi = S - 1;                            // first j with cum_q[j] > target (clamped to S-1)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.cum_q[mid] > target) hi = mid; else lo = mid + 1;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) a.out[d * a.n_points + r] = a.x[d * S + lo] + nrm[d] * a.bw[d];   // kernel_density.py:147-148
}

int kde_check(const glabc_kde* k)
{
    if (!k) return GLABC_ERR_NULL;
    if (k->dim < 1 || k->dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (k->n_samples < 1) return GLABC_ERR_ARG;
    if (!k->x || !k->log_w) return GLABC_ERR_NULL;
    for (int d = 0; d < k->dim; ++d)
        if (!(k->bandwidth[d] > 0.0f) || k->bandwidth[d] == __builtin_inff()) return GLABC_ERR_ARG;
    return GLABC_OK;
}

template <int D>
KdeArgs<D> kde_pack(const glabc_kde* k)
{
    KdeArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.x = k->x; a.log_w = k->log_w; a.cum_q = k->cum_q; a.n_samples = k->n_samples;
    for (int d = 0; d < D; ++d) a.bw[d] = k->bandwidth[d];
    a.sum_log_bw = k->sum_log_bw; a.c_2pi = k->c_2pi;
    return a;
}

int launched() { return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH; }

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int glabc_kde_fit(const float* x, const float* w_raw, int64_t n_samples, int32_t dim, double h,
                                                         const float* bw_fixed, float* weights_out, float* log_w_out,
                                                         int64_t* wq_out, float* consts_out, void* stream)
{
    if (!x || !weights_out || !log_w_out || !wq_out || !consts_out) return GLABC_ERR_NULL;
    if (dim < 1 || dim > GLABC_MAX_DIM) return GLABC_ERR_DIM;
    if (n_samples < 1 || n_samples > (int64_t)1 << 22) return GLABC_ERR_ARG;
    FitArgs a;
    std::memset(&a, 0, sizeof a);
    a.x = x; a.w_raw = w_raw; a.n = n_samples; a.h = (float)h;
    a.weights = weights_out; a.log_w = log_w_out; a.wq = wq_out; a.consts = consts_out;
    if (bw_fixed) {
        a.has_bw = 1;
        for (int d = 0; d < dim; ++d) {
            if (!(bw_fixed[d] > 0.0f)) return GLABC_ERR_ARG;
            a.bw[d] = bw_fixed[d];
        }
    } else if (!(h > 0.0)) {
        return GLABC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dim) {
    case 1: hipLaunchKernelGGL((kde_fit_kernel<1>), dim3(1), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((kde_fit_kernel<2>), dim3(1), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((kde_fit_kernel<3>), dim3(1), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((kde_fit_kernel<4>), dim3(1), dim3(256), 0, s, a); break;
    }
    return launched();
}

__attribute__((visibility("default"))) int glabc_kde_log_prob(const glabc_kde* kde, const float* pts, int64_t n_points, float* out,
                                                              void* stream)
{
    int rc = kde_check(kde);
    if (rc) return rc;
    if (n_points < 0 || n_points > (int64_t)1 << 31) return GLABC_ERR_ARG;
    if (n_points == 0) return GLABC_OK;
    if (!pts || !out) return GLABC_ERR_NULL;
    if (kde->n_samples > (int64_t)1 << 22) return GLABC_ERR_ARG;         // fixed-point sum headroom: 2^22 terms <= 1
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n_points + 3) / 4)), block(256);
#define GLABC_CASE(d)                                                              \
    case d: {                                                                      \
        KdeArgs<d> a = kde_pack<d>(kde);                                           \
        a.pts = pts; a.n_points = n_points; a.out = out; a.pts_stride = n_points;  \
        hipLaunchKernelGGL((kde_log_prob_kernel<d>), grid, block, 0, s, a);        \
        break;                                                                     \
    }
    switch (kde->dim) { GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) }
#undef GLABC_CASE
    return launched();
}

__attribute__((visibility("default"))) int glabc_kde_log_prob_indexed(const glabc_kde* kde, const float* pts, int64_t stride,
                                                                      const int32_t* idx, const int32_t* n_dev, int64_t max_points,
                                                                      float* out, void* stream)
{
    int rc = kde_check(kde);
    if (rc) return rc;
    if (max_points < 0 || max_points > (int64_t)1 << 31 || stride < max_points) return GLABC_ERR_ARG;
    if (max_points == 0) return GLABC_OK;
    if (!pts || !out || !idx || !n_dev) return GLABC_ERR_NULL;
    if (kde->n_samples > (int64_t)1 << 22) return GLABC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((max_points + 3) / 4)), block(256);       // wavefronts past the device-side count leave at once
#define GLABC_CASE(d)                                                              \
    case d: {                                                                      \
        KdeArgs<d> a = kde_pack<d>(kde);                                           \
        a.pts = pts; a.n_points = max_points; a.out = out; a.pts_stride = stride;  \
        a.idx = idx; a.n_dev = n_dev;                                              \
        hipLaunchKernelGGL((kde_log_prob_kernel<d>), grid, block, 0, s, a);        \
        break;                                                                     \
    }
    switch (kde->dim) { GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) }
#undef GLABC_CASE
    return launched();
}

__attribute__((visibility("default"))) int glabc_kde_sample(const glabc_kde* kde, int64_t n, uint64_t seed, int64_t row0, float* out,
                                                            void* stream)
{
    int rc = kde_check(kde);
    if (rc) return rc;
    if (n < 0 || row0 < 0) return GLABC_ERR_ARG;
    if (n == 0) return GLABC_OK;
    if (!out || !kde->cum_q) return GLABC_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define GLABC_CASE(d)                                                              \
    case d: {                                                                      \
        KdeArgs<d> a = kde_pack<d>(kde);                                           \
        a.n_points = n; a.out = out; a.row0 = row0;                                \
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);            \
        hipLaunchKernelGGL((kde_sample_kernel<d>), grid, block, 0, s, a);          \
        break;                                                                     \
    }
    switch (kde->dim) { GLABC_CASE(1) GLABC_CASE(2) GLABC_CASE(3) GLABC_CASE(4) }
#undef GLABC_CASE
    return launched();
}

}  // extern "C"
