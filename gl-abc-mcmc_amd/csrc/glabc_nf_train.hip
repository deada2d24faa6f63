// glabc_nf_train.hip -- the training step of GLMCMC_NF's flow on the gfx950 matrix cores: loss, gradient, Adam.
//
// Reference: GLMCMC_NFs.py:63 builds torch.optim.Adam(NF_model.parameters(), lr=5e-4, weight_decay=1e-5) and, each time a pool
// is used up (at most Train_step times, :112-124), takes ONE step on loss = NF_model.forward_kld(Train_t) =
// -mean(NF_model.log_prob(Train_t)) over the systematically resampled pool; autograd does the differentiation there.
// Here the gradient is written out by hand.  A coupling is invertible and leaves its conditioner input untouched, so
// nothing has to be stored on the way down: glabc_nf_inverse (glabc_nf.hip) pulls the rows back to the base space, then one
// launch per coupling -- in the order the flow would push them forward again -- recomputes the coupling's activations from
// its OUTPUT state, un-does it, and back-propagates.  Per row (z0 = conditioner input, z1' = transformed coordinate, g = dL/d.):
//     h1 = relu(W1 z0 + b1);  a2 = W2 h1 + b2;  h2 = relu(a2);  (shift, log_s) = W3 h2 + b3;  z1 = z1' exp(log_s) + shift
//     d z1 = g1' exp(-log_s);  d shift = -d z1;  d log_s = -g1' z1' - dL/dlog_q          (log_q -= log_s;  dL/dlog_q = -1/n)
//     d a2 = (a2 > 0) (W3^T dp);     d a1 = (a1 > 0) (W2^T d a2);     d z0 = g0 + W1^T d a1
//     dW3 += dp (x) h2,  db3 += dp,  dW2 += d a2 (x) h1,  db2 += d a2,  dW1 += d a1 z0,  db1 += d a1      (sums over rows)
//
// Mapping (v_mfma_f32_32x32x2_f32, D[m][n] = sum_k A[m][k] B[k][n]; lane l supplies A[l&31][l>>5] and B[l>>5][l&31] and holds
// D[(r&3) + 8(r>>2) + 4(l>>5)][l&31], r = 0..15).  A workgroup = 4 waves = a batch of 128 rows, one 32-row tile per wave:
//   (1) a2[i][row]   = sum_k W2[i][k] h1[k][row]        A = W2 (LDS, lanes <-> i),  B = h1 made on the fly (lanes <-> row)
//   (2) dh1[row][kk] = sum_i da2[i][row] W2[i][kk]      A = da2 -- the accumulators of (1) ARE in A's layout when the steps
//                                                       run over i in accumulator order (lane half h supplies i0 + 4h);
//                                                       B = W2 (LDS, lanes <-> kk).  The result has lanes <-> kk and
//                                                       registers <-> rows: dW1 / db1 are lane-local sums, d z0 is a
//                                                       32-lane butterfly of 16 values.
//   (3) dW2[i][k]    = sum_row da2[i][row] h1[k][row]   contraction over ROWS: a2 of the whole batch is staged in LDS
//                                                       row-major, wave w takes the 32 neurons i = 32w.. for all 128 rows:
//                                                       A = da2 rebuilt from staged a2 (lanes <-> i), B = h1 on the fly
//                                                       (lanes <-> k); db2 / dW3 are lane-local sums on the way.
// One LDS image of W2 serves (1) and (2): row stride 129 floats makes both "fixed k, lanes over i" and "fixed i, lanes over k"
// conflict-free.  LDS: 66 KiB weights + 66 KiB staging + 5 KiB vectors.  Per wave and 32 rows: 3 x 256 MFMAs (the forward
// pass has 256).  Workgroups write partial sums in the parameter-block layout; one reduction kernel adds them in a fixed
// order in double, so a step is reproducible to the bit from run to run.
//
// Parity: floating point.  tests/test_nf_train.py holds loss and every gradient to the CPU checker's double-precision
// restatement (oracle_nf_grad), itself checked against torch autograd in float64, within 2e-4 of each tensor's largest entry.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/glabc.h"
#include "../../include/glabc_numerics.h"
#include "glabc_nf_layout.h"

namespace glabc {

constexpr int BW_WAVES = 4;
constexpr int BW_ROWS = 32 * BW_WAVES;           // rows of a workgroup's batch
constexpr int WS = NF_H + 1;                     // padded row stride of the two LDS matrices
constexpr int B_W2 = 0;                          // W2[i][k] at i*WS + k
constexpr int B_S = B_W2 + NF_H * WS;            // staged a2[row][i] at row*WS + i
constexpr int B_W1 = B_S + BW_ROWS * WS;
constexpr int B_B1 = B_W1 + NF_H;
constexpr int B_B2 = B_B1 + NF_H;
constexpr int B_W30 = B_B2 + NF_H;
constexpr int B_W31 = B_W30 + NF_H;
constexpr int B_Z0 = B_W31 + NF_H;               // per row of the batch
constexpr int B_DSH = B_Z0 + BW_ROWS;
constexpr int B_DLS = B_DSH + BW_ROWS;
constexpr int B_DZ0 = B_DLS + BW_ROWS;
constexpr int B_B3 = B_DZ0 + BW_ROWS;
constexpr int B_FLOATS = B_B3 + 4;
constexpr int BW_MAX_WGS = 256;                  // one workgroup per CU

struct BwArgs {
    const float* block;       // this coupling's parameters
    float* z;                 // [2][n]: in = state after this coupling in the log_prob direction, out = the state before it
    float* g;                 // [2][n]: dL/dstate, same convention
    float* partial;           // [gridDim.x][NF_BLOCK_FLOATS] gradient sums of this coupling, one block per workgroup
    int64_t n_rows;
    int32_t rows_per_wg;      // multiple of BW_ROWS
    float gl;                 // dL/dlog_q of a row = -1/n
};

__global__ void __launch_bounds__(64 * BW_WAVES) nf_backward_kernel(const BwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    for (int idx = threadIdx.x; idx < NF_H * NF_H; idx += 64 * BW_WAVES)                      // block holds W2^T [k][i]
        lds[B_W2 + (idx & (NF_H - 1)) * WS + (idx >> 7)] = a.block[NF_W2_OFF + idx];
    for (int i = threadIdx.x; i < NF_H; i += 64 * BW_WAVES) {
        lds[B_W1 + i] = a.block[NF_W1_OFF + i];
        lds[B_B1 + i] = a.block[NF_B1_OFF + i];
        lds[B_B2 + i] = a.block[NF_V4_OFF + 4 * i];
        lds[B_W30 + i] = a.block[NF_V4_OFF + 4 * i + 1];
        lds[B_W31 + i] = a.block[NF_V4_OFF + 4 * i + 2];
    }
    if (threadIdx.x < 2) lds[B_B3 + threadIdx.x] = a.block[NF_B3_OFF + threadIdx.x];
    __syncthreads();

    f32x16 gw0, gw1, gw2, gw3;                    // dW2[32 wave + m][32 t + col], t = 0..3
#pragma unroll
    for (int r = 0; r < 16; ++r) gw0[r] = gw1[r] = gw2[r] = gw3[r] = 0.0f;
    float gb2 = 0.0f, gw30 = 0.0f, gw31 = 0.0f;   // neuron 32 wave + col, the rows 2s + half
    float gw1a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gb1a[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // unit 32 t + col, this wave's tiles
    float gb30 = 0.0f, gb31 = 0.0f;               // this lane's rows
    float w1c[4], b1c[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        w1c[t] = lds[B_W1 + 32 * t + col];
        b1c[t] = lds[B_B1 + 32 * t + col];
    }
    const float w30i = lds[B_W30 + 32 * wave + col], w31i = lds[B_W31 + 32 * wave + col];
    const float b30 = lds[B_B3], b31 = lds[B_B3 + 1];

    const int64_t wg_row0 = (int64_t)blockIdx.x * a.rows_per_wg;
    // the rows of a tile are loaded one batch ahead (issued before phase 2, consumed after the next barrier pair)
    float nz0, nz1p, ng0, ng1p;
    auto fetch = [&](int64_t first) {
        const int64_t r = first + 32 * wave + col, rc = r < a.n_rows ? r : a.n_rows - 1;
        nz0 = a.z[rc];
        nz1p = a.z[a.n_rows + rc];
        ng0 = a.g[rc];
        ng1p = a.g[a.n_rows + rc];
    };
    fetch(wg_row0 < a.n_rows ? wg_row0 : 0);
    for (int64_t row0 = wg_row0; row0 < wg_row0 + a.rows_per_wg && row0 < a.n_rows; row0 += BW_ROWS) {
        // ---------------------------------------------------------------- phase 1: this wave's tile, rows on lanes
        const int64_t row = row0 + 32 * wave + col;
        const bool valid = row < a.n_rows;
        const float z0 = nz0, z1p = nz1p;
        const float g0 = valid ? ng0 : 0.0f, g1p = valid ? ng1p : 0.0f;
        const float gl_row = valid ? a.gl : 0.0f;
        f32x16 a0, a1, a2, a3;                                                                // (1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            a0[r] = lds[B_B2 + i];
            a1[r] = lds[B_B2 + i + 32];
            a2[r] = lds[B_B2 + i + 64];
            a3[r] = lds[B_B2 + i + 96];
        }
        {
            const float* w1 = lds + B_W1 + half;
            const float* bb1 = lds + B_B1 + half;
            const float* wi = lds + B_W2 + col * WS + half;                                   // W2[32 t + col][2 s + half]
#pragma unroll 8
            for (int s = 0; s < 64; ++s) {
                const float h1 = __builtin_fmaxf(__builtin_fmaf(w1[2 * s], z0, bb1[2 * s]), 0.0f);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s], h1, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 32 * WS], h1, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 64 * WS], h1, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 96 * WS], h1, a3, 0, 0, 0);
            }
        }
        float p0 = 0.0f, p1 = 0.0f;
        float* srow = lds + B_S + (32 * wave + col) * WS + 4 * half;
        auto head = [&](const f32x16& acc, int t) {                  // W3 h2 for this lane's 16 units of tile t; a2 -> staging
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i0 = 32 * t + (r & 3) + 8 * (r >> 2);
                const float h2 = __builtin_fmaxf(acc[r], 0.0f);
                p0 = __builtin_fmaf(lds[B_W30 + 4 * half + i0], h2, p0);
                p1 = __builtin_fmaf(lds[B_W31 + 4 * half + i0], h2, p1);
                srow[i0] = acc[r];
            }
        };
        head(a0, 0);
        head(a1, 1);
        head(a2, 2);
        head(a3, 3);
        const float shift = (p0 + __shfl_xor(p0, 32, 64)) + b30;
        const float log_s = (p1 + __shfl_xor(p1, 32, 64)) + b31;
        const float z1 = z1p * glabc_expf_b(log_s) + shift;          // the coupling un-done
        const float dz1 = g1p * glabc_expf_b(-log_s);
        const float dsh = -dz1;
        const float dls = -(g1p * z1p) - gl_row;
        if (half == 0) {
            lds[B_Z0 + 32 * wave + col] = z0;
            lds[B_DSH + 32 * wave + col] = dsh;
            lds[B_DLS + 32 * wave + col] = dls;
            gb30 += dsh;
            gb31 += dls;
        }
        f32x16 d0, d1, d2, d3;                                                                // (2)
#pragma unroll
        for (int r = 0; r < 16; ++r) d0[r] = d1[r] = d2[r] = d3[r] = 0.0f;
        const float* wk = lds + B_W2 + 4 * half * WS + col;                                   // W2[i0 + 4 half][32 t + col]
        auto back = [&](const f32x16& acc, int t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i0 = 32 * t + (r & 3) + 8 * (r >> 2);
                const float dh2 = __builtin_fmaf(lds[B_W30 + 4 * half + i0], dsh, lds[B_W31 + 4 * half + i0] * dls);
                const float da2 = acc[r] > 0.0f ? dh2 : 0.0f;
                const float* wr = wk + i0 * WS;
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[0], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[32], d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[64], d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[96], d3, 0, 0, 0);
            }
        };
        back(a0, 0);
        back(a1, 1);
        back(a2, 2);
        back(a3, 3);
        // d_t[r] = dh1 of unit 32 t + col for the tile's row m_r = (r&3) + 8(r>>2) + 4 half
        float mine = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float zr = __shfl(z0, m, 64);
            float part = 0.0f;
            auto unit = [&](float dh1, int t) {
                const float da1 = __builtin_fmaf(w1c[t], zr, b1c[t]) > 0.0f ? dh1 : 0.0f;
                gw1a[t] = __builtin_fmaf(da1, zr, gw1a[t]);
                gb1a[t] += da1;
                part = __builtin_fmaf(w1c[t], da1, part);
            };
            unit(d0[r], 0);
            unit(d1[r], 1);
            unit(d2[r], 2);
            unit(d3[r], 3);
            part += __shfl_xor(part, 1, 64);                         // over the 32 lanes of this half (the 128 units)
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 8, 64);
            part += __shfl_xor(part, 16, 64);
            mine = (col == r) ? part : mine;
        }
        if (col < 16) lds[B_DZ0 + 32 * wave + (col & 3) + 8 * (col >> 2) + 4 * half] = mine;
        __syncthreads();
        if (half == 0 && valid) {
            a.z[row] = z1;                                           // state before the coupling: (z1, z0)
            a.z[a.n_rows + row] = z0;
            a.g[row] = dz1;
            a.g[a.n_rows + row] = g0 + lds[B_DZ0 + 32 * wave + col];
        }
        if (row0 + BW_ROWS < wg_row0 + a.rows_per_wg && row0 + BW_ROWS < a.n_rows) fetch(row0 + BW_ROWS);
        // ---------------------------------------------------------------- phase 2: neurons 32 wave.., all rows of the batch
        {
            const float* sa = lds + B_S + half * WS + 32 * wave + col;                        // a2[32 wave + col][row 2 s + half]
            const float* zs = lds + B_Z0 + half;
            const float* ds = lds + B_DSH + half;
            const float* dl = lds + B_DLS + half;
#pragma unroll 8
            for (int s = 0; s < 64; ++s) {
                const float a2v = sa[2 * s * WS];
                const float dsr = ds[2 * s], dlr = dl[2 * s], zr = zs[2 * s];
                const float dh2 = __builtin_fmaf(w30i, dsr, w31i * dlr);
                const float da2 = a2v > 0.0f ? dh2 : 0.0f;
                const float h2 = __builtin_fmaxf(a2v, 0.0f);
                gb2 += da2;
                gw30 = __builtin_fmaf(h2, dsr, gw30);
                gw31 = __builtin_fmaf(h2, dlr, gw31);
                gw0 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, __builtin_fmaxf(__builtin_fmaf(w1c[0], zr, b1c[0]), 0.0f), gw0, 0, 0, 0);
                gw1 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, __builtin_fmaxf(__builtin_fmaf(w1c[1], zr, b1c[1]), 0.0f), gw1, 0, 0, 0);
                gw2 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, __builtin_fmaxf(__builtin_fmaf(w1c[2], zr, b1c[2]), 0.0f), gw2, 0, 0, 0);
                gw3 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, __builtin_fmaxf(__builtin_fmaf(w1c[3], zr, b1c[3]), 0.0f), gw3, 0, 0, 0);
            }
        }
        __syncthreads();                                             // the staging area is free again
    }

    // ------------------------------------------------------------------------ this workgroup's sums, block layout
    float* out = a.partial + (int64_t)blockIdx.x * NF_BLOCK_FLOATS;
    auto put = [&](const f32x16& acc, int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half, k = 32 * t + col;
            out[NF_W2_OFF + k * NF_H + i] = acc[r];
        }
    };
    put(gw0, 0);
    put(gw1, 1);
    put(gw2, 2);
    put(gw3, 3);
    {
        const float s2 = gb2 + __shfl_xor(gb2, 32, 64), s30 = gw30 + __shfl_xor(gw30, 32, 64), s31 = gw31 + __shfl_xor(gw31, 32, 64);
        if (half == 0) {
            const int i = 32 * wave + col;
            *reinterpret_cast<float4*>(out + NF_V4_OFF + 4 * i) = make_float4(s2, s30, s31, 0.0f);
        }
    }
    // dW1 / db1 / db3: sums over the waves' tiles, through the (now free) staging area in a fixed order
    float* sc = lds + B_S;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float sw = gw1a[t] + __shfl_xor(gw1a[t], 32, 64), sb = gb1a[t] + __shfl_xor(gb1a[t], 32, 64);
        if (half == 0) {
            sc[wave * 256 + 32 * t + col] = sw;
            sc[wave * 256 + 128 + 32 * t + col] = sb;
        }
    }
    if (half == 0) {
        sc[1024 + wave * 64 + col] = gb30;
        sc[1024 + wave * 64 + 32 + col] = gb31;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int j = threadIdx.x;                                   // 0..127: dW1, 128..255: db1
        const float v = ((sc[j] + sc[256 + j]) + sc[512 + j]) + sc[768 + j];
        out[(j < 128 ? NF_W1_OFF : NF_B1_OFF - 128) + j] = v;
    }
    if (threadIdx.x < 2) {
        float v = 0.0f;
        for (int w = 0; w < BW_WAVES; ++w)
            for (int c = 0; c < 32; ++c) v += sc[1024 + w * 64 + 32 * threadIdx.x + c];
        out[NF_B3_OFF + threadIdx.x] = v;
        out[NF_B3_OFF + 2 + threadIdx.x] = 0.0f;
    }
}

// Start of the backward sweep.  z = the base-space points (glabc_nf_inverse), lq = log_prob of the rows:
//   g = dL/dz = gl * d base.log_prob / dz = gl * (-(z - loc) / scale^2)
//   per-block partial sums (double): sum lq, sum e_j / scale_j, sum (e_j^2 - 1), e = (z - loc) / scale
struct BaseArgs {
    const float* z;
    const float* lq;
    float* g;
    double* partial;          // [gridDim.x][5]
    int64_t n_rows;
    float loc[2], scale[2], gl;
};

__global__ void __launch_bounds__(256) nf_base_grad_kernel(const BaseArgs a)
{
    __shared__ double red[5][256];
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n_rows; r += (int64_t)gridDim.x * 256) {
        const float e0 = (a.z[r] - a.loc[0]) / a.scale[0], e1 = (a.z[a.n_rows + r] - a.loc[1]) / a.scale[1];
        a.g[r] = a.gl * (-(e0 / a.scale[0]));
        a.g[a.n_rows + r] = a.gl * (-(e1 / a.scale[1]));
        s[0] += (double)a.lq[r];
        s[1] += (double)(e0 / a.scale[0]);
        s[2] += (double)(e1 / a.scale[1]);
        s[3] += (double)e0 * (double)e0 - 1.0;
        s[4] += (double)e1 * (double)e1 - 1.0;
    }
    for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 5) a.partial[(int64_t)blockIdx.x * 5 + threadIdx.x] = red[threadIdx.x][0];
}

// grads[c][j] = sum over workgroups of partial[c][wg][j], in workgroup order, in double; base gradients and the loss
struct ReduceArgs {
    const float* partial;     // [n_couplings][n_wgs][NF_BLOCK_FLOATS]
    const double* base_partial;   // [n_base_blocks][5]
    float* grad_params;       // [n_couplings][NF_BLOCK_FLOATS]
    float* grad_base;         // loc0, loc1, log_scale0, log_scale1
    float* loss;
    int64_t total;            // n_couplings * NF_BLOCK_FLOATS
    int32_t n_wgs, n_base_blocks;
    double gl, n_rows;
};

__global__ void __launch_bounds__(256) nf_grad_reduce_kernel(const ReduceArgs a)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < a.total) {
        const int64_t c = j / NF_BLOCK_FLOATS, e = j % NF_BLOCK_FLOATS;
        const float* p = a.partial + c * (int64_t)a.n_wgs * NF_BLOCK_FLOATS + e;
        double s = 0.0;
        for (int w = 0; w < a.n_wgs; ++w) s += (double)p[(int64_t)w * NF_BLOCK_FLOATS];
        a.grad_params[j] = (float)s;
    }
    if (blockIdx.x == 0 && threadIdx.x < 5) {
        double s = 0.0;
        for (int b = 0; b < a.n_base_blocks; ++b) s += a.base_partial[(int64_t)b * 5 + threadIdx.x];
        if (threadIdx.x == 0)
            *a.loss = (float)(-s / a.n_rows);                        // forward_kld = -mean(log_prob)
        else
            a.grad_base[threadIdx.x - 1] = (float)(a.gl * s);        // d/dloc_j = gl sum e_j/scale_j, d/dlog_scale_j = gl sum (e_j^2 - 1)
    }
}

// torch.optim.Adam (GLMCMC_NFs.py:63; L2 weight decay folded into the gradient, no amsgrad), one element per work-item, the
// operation order of torch's _single_tensor_adam
struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float beta1, one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay, step_size, bias2_sqrt;
};

__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n) return;
    const float p = a.p[j];
    const float g = a.weight_decay != 0.0f ? a.g[j] + a.weight_decay * p : a.g[j];          // grad.add(param, alpha=weight_decay)
    const float m = a.m[j] + (g - a.m[j]) * a.one_minus_beta1;                             // exp_avg.lerp_(grad, 1 - beta1)
    const float v = a.v[j] * a.beta2 + a.one_minus_beta2 * (g * g);                        // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = __builtin_sqrtf(v) / a.bias2_sqrt + a.eps;
    a.m[j] = m;
    a.v[j] = v;
    a.p[j] = p - a.step_size * (m / denom);                                               // param.addcdiv_(exp_avg, denom, value=-step_size)
}

static int wgs_for(int64_t n_rows, int* rows_per_wg)
{
    const int64_t batches = (n_rows + BW_ROWS - 1) / BW_ROWS;
    const int64_t wgs = batches < BW_MAX_WGS ? batches : BW_MAX_WGS;
    const int64_t per = (batches + wgs - 1) / wgs;
    *rows_per_wg = (int)(per * BW_ROWS);
    return (int)((batches + per - 1) / per);
}

constexpr int BASE_BLOCKS = 256;

struct Workspace {
    float *z, *g, *lq, *partial;
    double* base_partial;
};

static int64_t carve(int32_t n_couplings, int64_t n_rows, char* base, Workspace* w)
{
    int rows_per_wg;
    const int wgs = wgs_for(n_rows, &rows_per_wg);
    int64_t at = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + at : nullptr;
        at += (bytes + 255) / 256 * 256;
        return p;
    };
    w->z = (float*)take(2 * n_rows * 4);
    w->g = (float*)take(2 * n_rows * 4);
    w->lq = (float*)take(n_rows * 4);
    w->partial = (float*)take((int64_t)n_couplings * wgs * NF_BLOCK_FLOATS * 4);
    w->base_partial = (double*)take(BASE_BLOCKS * 5 * 8);
    return at;
}

}  // namespace glabc

using namespace glabc;

extern "C" {

int glabc_nf_inverse(const glabc_flow* flow, const float* x, int64_t n_rows, float* z_out, float* log_q, void* stream);

__attribute__((visibility("default"))) int glabc_nf_grad_workspace(int32_t n_couplings, int64_t n_rows, int64_t* bytes)
{
    if (!bytes) return GLABC_ERR_NULL;
    if (n_couplings < 1 || n_couplings > 4096 || n_rows < 1) return GLABC_ERR_ARG;
    Workspace w;
    *bytes = carve(n_couplings, n_rows, nullptr, &w);
    return GLABC_OK;
}

__attribute__((visibility("default"))) int glabc_nf_grad(const glabc_flow* flow, const float* x, int64_t n_rows, void* workspace,
                                                         int64_t workspace_bytes, float* grad_params, float* grad_base, float* loss,
                                                         void* stream)
{
    if (!flow || !flow->params || !x || !workspace || !grad_params || !grad_base || !loss) return GLABC_ERR_NULL;
    if (flow->n_couplings < 1 || flow->n_couplings > 4096 || flow->hidden != NF_H || n_rows < 1) return GLABC_ERR_ARG;
    Workspace w;
    if (carve(flow->n_couplings, n_rows, (char*)workspace, &w) > workspace_bytes) return GLABC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc = glabc_nf_inverse(flow, x, n_rows, w.z, w.lq, stream);                            // rows -> base space, log_prob
    if (rc) return rc;
    const double gl = -1.0 / (double)n_rows;
    BaseArgs b;
    b.z = w.z;
    b.lq = w.lq;
    b.g = w.g;
    b.partial = w.base_partial;
    b.n_rows = n_rows;
    for (int j = 0; j < 2; ++j) {
        b.loc[j] = flow->base_loc[j];
        b.scale[j] = flow->base_scale[j];
    }
    b.gl = (float)gl;
    hipLaunchKernelGGL(nf_base_grad_kernel, dim3(BASE_BLOCKS), dim3(256), 0, s, b);
    static bool lds_ok = false;
    if (!lds_ok) {
        if (hipFuncSetAttribute((const void*)nf_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B_FLOATS * 4) != hipSuccess)
            return GLABC_ERR_LAUNCH;
        lds_ok = true;
    }
    int rows_per_wg;
    const int wgs = wgs_for(n_rows, &rows_per_wg);
    for (int c = 0; c < flow->n_couplings; ++c) {                     // log_prob applied n-1 .. 0: the sweep back runs 0 .. n-1
        BwArgs a;
        a.block = flow->params + (int64_t)c * NF_BLOCK_FLOATS;
        a.z = w.z;
        a.g = w.g;
        a.partial = w.partial + (int64_t)c * wgs * NF_BLOCK_FLOATS;
        a.n_rows = n_rows;
        a.rows_per_wg = rows_per_wg;
        a.gl = (float)gl;
        hipLaunchKernelGGL(nf_backward_kernel, dim3(wgs), dim3(64 * BW_WAVES), B_FLOATS * 4, s, a);
    }
    ReduceArgs r;
    r.partial = w.partial;
    r.base_partial = w.base_partial;
    r.grad_params = grad_params;
    r.grad_base = grad_base;
    r.loss = loss;
    r.total = (int64_t)flow->n_couplings * NF_BLOCK_FLOATS;
    r.n_wgs = wgs;
    r.n_base_blocks = BASE_BLOCKS;
    r.gl = gl;
    r.n_rows = (double)n_rows;
    hipLaunchKernelGGL(nf_grad_reduce_kernel, dim3((unsigned)((r.total + 255) / 256)), dim3(256), 0, s, r);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

__attribute__((visibility("default"))) int glabc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                                           int64_t count, double lr, double beta1, double beta2, double eps,
                                                           double weight_decay, int32_t step, void* stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq) return GLABC_ERR_NULL;
    if (count < 0 || step < 1 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) ||
        !(weight_decay >= 0.0))
        return GLABC_ERR_ARG;
    if (count == 0) return GLABC_OK;
    AdamArgs a;
    a.p = params;
    a.g = grads;
    a.m = exp_avg;
    a.v = exp_avg_sq;
    a.n = count;
    a.beta1 = (float)beta1;
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.weight_decay = (float)weight_decay;
    a.step_size = (float)(lr / (1.0 - std::pow(beta1, (double)step)));
    a.bias2_sqrt = (float)std::sqrt(1.0 - std::pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

}  // extern "C"
