// glabc_nf_train.hip -- the training step of GLMCMC_NF's flow on the gfx950 matrix cores: loss, gradient, Adam.
//
// Reference: GLMCMC_NFs.py:63 builds torch.optim.Adam(NF_model.parameters(), lr=5e-4, weight_decay=1e-5) and, each time a pool
// is used up (at most Train_step times, :112-124), takes ONE step on loss = NF_model.forward_kld(Train_t) =
// -mean(NF_model.log_prob(Train_t)) over the systematically resampled pool; autograd does the differentiation there.
// Here the gradient is written out by hand.  A coupling leaves its conditioner input untouched and its output is the next
// coupling's input, so almost nothing has to be stored on the way down: glabc_nf_inverse (glabc_nf.hip) pulls the rows back to
// the base space and keeps ONE float per row and coupling -- the conditioner input it saw -- and then one launch per coupling,
// in the order the flow would push the rows forward again, recomputes the coupling's activations from that input and
// back-propagates (its transformed coordinate z1' is the NEXT coupling's kept input, so no coupling has to be un-done).  The recomputed a2 is the downward pass's own fmaf chain on the downward pass's own input, so the sweep
// opens exactly the ReLU gates the float32 evaluation opened (what autograd would differentiate).
// Per row (z0 = conditioner input, z1' = transformed coordinate, g = dL/d.):
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
// conflict-free.  Per wave and 32 rows: 3 x 256 MFMAs (the forward pass has 256).  Two forms: nf_backward_kernel stages a2
// itself (66 KiB, one wave per SIMD); nf_backward_kernel2 -- the default -- stages only its sign bits (see there), which lets
// eight waves share a CU.  Workgroups write partial sums in the parameter-block layout; one reduction kernel adds them in a fixed
// order in double, so a step is reproducible to the bit from run to run.
//
// Parity: floating point.  tests/test_nf_train.py holds loss and every gradient to the CPU checker's double-precision
// restatement (oracle_nf_grad), itself checked against torch autograd in float64, within 2e-4 of each tensor's largest entry.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/glabc.h"
#include "../../include/glabc_numerics.h"
#include "glabc_lds_grant.h"
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
constexpr int BW_MAX_WGS = 256;                  // workgroups per CU-round: one per CU (staged kernel), two (sign-bit kernel)

struct BwArgs {
    const float* block;       // this coupling's parameters
    const float* z1p;         // [n]: the transformed coordinate this coupling put out on the way down (= the next coupling's
                              //      conditioner input; the base-space point's second coordinate for coupling 0)
    float* g;                 // [2][n]: in = dL/d(state after this coupling in the log_prob direction), out = dL/d(state before it)
    float* partial;           // [gridDim.x][NF_BLOCK_FLOATS] gradient sums of this coupling, one block per workgroup
    int64_t n_rows;
    int32_t rows_per_wg;      // multiple of BW_ROWS
    float gl;                 // dL/dlog_q of a row = -1/n
    const float* z0_trace;    // [n]: this coupling's conditioner inputs as the downward pass saw them (exact gates)
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
    const float b31 = lds[B_B3 + 1];

    const int64_t wg_row0 = (int64_t)blockIdx.x * a.rows_per_wg;
    // the rows of a tile are loaded one batch ahead (issued before phase 2, consumed after the next barrier pair)
    float nz0, nz1p, ng0, ng1p;
    auto fetch = [&](int64_t first) {
        const int64_t r = first + 32 * wave + col, rc = r < a.n_rows ? r : a.n_rows - 1;
        nz0 = a.z0_trace[rc];
        nz1p = a.z1p[rc];
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
        float p1 = 0.0f;                                             // log_s = W3[1] h2 + b3[1] (the shift is not needed going back)
        float* srow = lds + B_S + (32 * wave + col) * WS + 4 * half;
        auto head = [&](const f32x16& acc, int t) {                  // W3 h2 for this lane's 16 units of tile t; a2 -> staging
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i0 = 32 * t + (r & 3) + 8 * (r >> 2);
                const float h2 = __builtin_fmaxf(acc[r], 0.0f);
                p1 = __builtin_fmaf(lds[B_W31 + 4 * half + i0], h2, p1);
                srow[i0] = acc[r];
            }
        };
        head(a0, 0);
        head(a1, 1);
        head(a2, 2);
        head(a3, 3);
        const float log_s = (p1 + __shfl_xor(p1, 32, 64)) + b31;
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
            a.g[row] = dz1;                                          // state before the coupling: (z1, z0)
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

// The same sweep with the batch's a2 NOT staged: phase 2 only needs the SIGN of a2 (d a2 = [a2 > 0] W3^T dp), which is one
// wave-wide compare per accumulator register (v_cmp -> 64-bit mask = two LDS words), 2 KiB per batch instead of 66 KiB; the
// one thing that needs a2's magnitude, dW3 += dp (x) relu(a2), is summed over the rows in phase 1 by a reduce-scatter over the
// 32 lanes of a half (128 values -> 4 per lane in 124 exchanges).  LDS drops to 71 KiB, so TWO workgroups share a CU = two
// waves per SIMD (the register budget is held to 256 by the launch bounds), and one workgroup's epilogue / butterflies /
// barriers are covered by the other's MFMAs.
// (the small arrays first: every one of their reads is then base register + 16-bit immediate; behind the 66 KiB matrix each
// read needed its own address register -- the round-1 lesson of glabc_nf.hip)
// WAVES = 4: two workgroups share a CU.  WAVES = 8: one workgroup of eight waves -- a batch is 256 rows, the weights are
// staged once per CU, and in phase 2 wave w takes hidden units 32 (w & 3) .. for the input units 64 (w >> 2) .. 64 (w >> 2) + 63
// only: 32 persistent accumulator registers instead of 64, which is what lets the kernel fit 256 registers without scratch.
template <int WAVES>
struct BwLds {
    static constexpr int ROWS = 32 * WAVES;
    static constexpr int W1 = 0, B1 = W1 + NF_H, B2 = B1 + NF_H, W30 = B2 + NF_H, W31 = W30 + NF_H;
    static constexpr int Z0 = W31 + NF_H, DSH = Z0 + ROWS, DLS = DSH + ROWS, DZ0 = DLS + ROWS, B3 = DZ0 + ROWS;
    static constexpr int MASK = B3 + 4;              // uint32 [128 units][WAVES tiles]: bit r = a2[i][row 32 tile + r] > 0
    static constexpr int W2 = MASK + NF_H * WAVES;   // W2[i][k] at i*WS + k
    static constexpr int FLOATS = W2 + NF_H * WS;
};
static_assert(2 * BwLds<4>::FLOATS * 4 <= 160 * 1024 && BwLds<8>::FLOATS * 4 <= 160 * 1024, "LDS budget");

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) nf_backward_kernel2(const BwArgs a)
{
    using L = BwLds<WAVES>;
    constexpr int C_W1 = L::W1, C_B1 = L::B1, C_B2 = L::B2, C_W30 = L::W30, C_W31 = L::W31, C_Z0 = L::Z0, C_DSH = L::DSH, C_DLS = L::DLS,
                  C_DZ0 = L::DZ0, C_B3 = L::B3, C_MASK = L::MASK, C_W2 = L::W2;
    constexpr int KSPLIT = WAVES / 4;                 // phase 2: waves per 32-unit block of hidden units (each takes 128 / KSPLIT input units)
    constexpr int NACC = 4 / KSPLIT;                  // ... and so many 32 x 32 accumulators of dW2
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    for (int idx = threadIdx.x; idx < NF_H * NF_H; idx += 64 * WAVES)                      // block holds W2^T [k][i]
        lds[C_W2 + (idx & (NF_H - 1)) * WS + (idx >> 7)] = a.block[NF_W2_OFF + idx];
    for (int i = threadIdx.x; i < NF_H; i += 64 * WAVES) {
        lds[C_W1 + i] = a.block[NF_W1_OFF + i];
        lds[C_B1 + i] = a.block[NF_B1_OFF + i];
        lds[C_B2 + i] = a.block[NF_V4_OFF + 4 * i];
        lds[C_W30 + i] = a.block[NF_V4_OFF + 4 * i + 1];
        lds[C_W31 + i] = a.block[NF_V4_OFF + 4 * i + 2];
    }
    if (threadIdx.x < 2) lds[C_B3 + threadIdx.x] = a.block[NF_B3_OFF + threadIdx.x];
    __syncthreads();

    const int iblk = wave & 3, kpart = wave >> 2;   // phase 2: this wave's block of hidden units / part of the input units
    f32x16 gw[NACC];                              // dW2[32 iblk + m][32 (NACC kpart + t) + col], t = 0..NACC-1
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gw[t][r] = 0.0f;
    float gb2 = 0.0f;                             // neuron 32 wave + col, the rows 2s + half
    float gw3acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // dW3[col >> 4][unit (col & 15) of tile t, this half], this wave's tiles
    float gw1a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gb1a[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // unit 32 t + col, this wave's tiles
    float gb30 = 0.0f, gb31 = 0.0f;               // this lane's rows

    const int64_t wg_row0 = (int64_t)blockIdx.x * a.rows_per_wg;
    // the rows of a tile are loaded one batch ahead (issued before phase 2, consumed after the next barrier pair)
    float nz0, nz1p, ng0, ng1p;
    auto fetch = [&](int64_t first) {
        const int64_t r = first + 32 * wave + col, rc = r < a.n_rows ? r : a.n_rows - 1;
        nz0 = a.z0_trace[rc];
        nz1p = a.z1p[rc];
        ng0 = a.g[rc];
        ng1p = a.g[a.n_rows + rc];
    };
    fetch(wg_row0 < a.n_rows ? wg_row0 : 0);
    for (int64_t row0 = wg_row0; row0 < wg_row0 + a.rows_per_wg && row0 < a.n_rows; row0 += L::ROWS) {
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
            a0[r] = lds[C_B2 + i];
            a1[r] = lds[C_B2 + i + 32];
            a2[r] = lds[C_B2 + i + 64];
            a3[r] = lds[C_B2 + i + 96];
        }
        {
            const float* w1 = lds + C_W1 + half;
            const float* bb1 = lds + C_B1 + half;
            const float* wi = lds + C_W2 + col * WS + half;                                   // W2[32 t + col][2 s + half]
#pragma unroll 8
            for (int s = 0; s < 64; ++s) {
                const float h1 = __builtin_fmaxf(__builtin_fmaf(w1[2 * s], z0, bb1[2 * s]), 0.0f);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s], h1, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 32 * WS], h1, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 64 * WS], h1, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wi[2 * s + 96 * WS], h1, a3, 0, 0, 0);
            }
        }
        float p1 = 0.0f;                                             // log_s = W3[1] h2 + b3[1] (the shift is not needed going back)
        uint32_t* mask = reinterpret_cast<uint32_t*>(lds) + C_MASK;
        auto head = [&](const f32x16& acc, int t) {                  // W3 h2 for this lane's 16 units of tile t; sign bits -> LDS
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i0 = 32 * t + (r & 3) + 8 * (r >> 2);
                const float h2 = __builtin_fmaxf(acc[r], 0.0f);
                p1 = __builtin_fmaf(lds[C_W31 + 4 * half + i0], h2, p1);
                const unsigned long long m = __ballot(acc[r] > 0.0f);     // lanes 0..31: unit i0, 32..63: unit i0 + 4; bit = row
                if (lane == 0) {
                    mask[i0 * WAVES + wave] = (uint32_t)m;
                    mask[(i0 + 4) * WAVES + wave] = (uint32_t)(m >> 32);
                }
            }
        };
        head(a0, 0);
        head(a1, 1);
        head(a2, 2);
        head(a3, 3);
        const float log_s = (p1 + __shfl_xor(p1, 32, 64)) + lds[C_B3 + 1];
        const float dz1 = g1p * glabc_expf_b(-log_s);
        const float dsh = -dz1;
        const float dls = -(g1p * z1p) - gl_row;
        if (half == 0) {
            lds[C_Z0 + 32 * wave + col] = z0;
            lds[C_DSH + 32 * wave + col] = dsh;
            lds[C_DLS + 32 * wave + col] = dls;
            gb30 += dsh;
            gb31 += dls;
        }
        {
            // dW3[c][i] += sum over the tile's rows of relu(a2[i][row]) dp_c[row]: per 32-unit tile t 32 values per lane
            // (c, r), summed over the 32 lanes of this half by a reduce-scatter -- lane col ends with the total of
            // (c, r) = (col >> 4, col & 15), one per tile
            auto rows_sum = [&](const f32x16& acc, int t) {
                float v[32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h2 = __builtin_fmaxf(acc[r], 0.0f);
                    v[r] = h2 * dsh;
                    v[16 + r] = h2 * dls;
                }
#pragma unroll
                for (int n = 16, off = 16; n >= 1; n >>= 1, off >>= 1) {
                    const bool up = (col & off) != 0;
#pragma unroll
                    for (int k = 0; k < n; ++k) {
                        const float keep = up ? v[k + n] : v[k], send = up ? v[k] : v[k + n];
                        v[k] = keep + __shfl_xor(send, off, 64);
                    }
                }
                gw3acc[t] += v[0];
            };
            rows_sum(a0, 0);
            rows_sum(a1, 1);
            rows_sum(a2, 2);
            rows_sum(a3, 3);
        }
        // (2) in two passes of two 32-unit output tiles each (32 accumulator registers instead of 64: the kernel has to fit
        // 256 registers for two waves per SIMD): d_u[r] = dh1 of unit 32 (2 pass + u) + col for the tile's row
        // m_r = (r&3) + 8(r>>2) + 4 half
        const float* wk = lds + C_W2 + 4 * half * WS + col;                                   // W2[i0 + 4 half][32 t + col]
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.0f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            f32x16 d0, d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) d0[r] = d1[r] = 0.0f;
            auto back = [&](const f32x16& acc, int t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i0 = 32 * t + (r & 3) + 8 * (r >> 2);
                    const float dh2 = __builtin_fmaf(lds[C_W30 + 4 * half + i0], dsh, lds[C_W31 + 4 * half + i0] * dls);
                    const float da2 = acc[r] > 0.0f ? dh2 : 0.0f;
                    const float* wr = wk + i0 * WS + 64 * pass;
                    d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[0], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, wr[32], d1, 0, 0, 0);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // bound the operand prefetch (register budget)
                }
            };
            back(a0, 0);
            back(a1, 1);
            back(a2, 2);
            back(a3, 3);
            // W1 / b1 of this pass's two 32-unit tiles, read here rather than held in eight registers through all three phases
            // (the kernel is at its 256-register budget: 16 registers used to be spilled)
            const float w1c[2] = {lds[C_W1 + 64 * pass + col], lds[C_W1 + 64 * pass + 32 + col]};
            const float b1c[2] = {lds[C_B1 + 64 * pass + col], lds[C_B1 + 64 * pass + 32 + col]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float zr = __shfl(z0, m, 64);
                auto unit = [&](float dh1, int u) {
                    const int t = 2 * pass + u;
                    const float da1 = __builtin_fmaf(w1c[u], zr, b1c[u]) > 0.0f ? dh1 : 0.0f;
                    gw1a[t] = __builtin_fmaf(da1, zr, gw1a[t]);
                    gb1a[t] += da1;
                    part[r] = __builtin_fmaf(w1c[u], da1, part[r]);
                };
                unit(d0[r], 0);
                unit(d1[r], 1);
            }
        }
        float mine = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float q = part[r];
            q += __shfl_xor(q, 1, 64);                               // over the 32 lanes of this half (the 128 units)
            q += __shfl_xor(q, 2, 64);
            q += __shfl_xor(q, 4, 64);
            q += __shfl_xor(q, 8, 64);
            q += __shfl_xor(q, 16, 64);
            mine = (col == r) ? q : mine;
        }
        if (col < 16) lds[C_DZ0 + 32 * wave + (col & 3) + 8 * (col >> 2) + 4 * half] = mine;
        __syncthreads();
        if (half == 0 && valid) {
            a.g[row] = dz1;                                          // state before the coupling: (z1, z0)
            a.g[a.n_rows + row] = g0 + lds[C_DZ0 + 32 * wave + col];
        }
        if (row0 + L::ROWS < wg_row0 + a.rows_per_wg && row0 + L::ROWS < a.n_rows) fetch(row0 + L::ROWS);
        // ---------------------------------------------------------------- phase 2: neurons 32 wave.., all rows of the batch
        {
            const float* zs = lds + C_Z0 + half;
            const float* ds = lds + C_DSH + half;
            const float* dl = lds + C_DLS + half;
            const float w30i = lds[C_W30 + 32 * iblk + col], w31i = lds[C_W31 + 32 * iblk + col];   // (read per batch: registers)
            float w1p[NACC], b1p[NACC];                              // this wave's input units 32 (NACC kpart + t) + col
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                w1p[t] = lds[C_W1 + 32 * (NACC * kpart + t) + col];
                b1p[t] = lds[C_B1 + 32 * (NACC * kpart + t) + col];
            }
            for (int tile = 0; tile < WAVES; ++tile) {               // rows 32 tile .. of the batch: wave `tile`'s sign bits
                // this half's sign bits moved to the even positions once per tile: the bit tests below then use immediate masks (with
                // `mw >> (2 ss + half)` the compiler kept the sixteen lane-dependent masks 1 << (2 ss + half) in registers)
                const uint32_t mw = mask[(32 * iblk + col) * WAVES + tile] >> half;
#pragma unroll
                for (int ss = 0; ss < 16; ++ss) {
                    const int s = 16 * tile + ss;                    // row 2 s + half = 32 tile + (2 ss + half)
                    const float dsr = ds[2 * s], dlr = dl[2 * s], zr = zs[2 * s];
                    const float dh2 = __builtin_fmaf(w30i, dsr, w31i * dlr);
                    const float da2 = ((mw >> (2 * ss)) & 1u) ? dh2 : 0.0f;
                    gb2 += da2;
#pragma unroll
                    for (int t = 0; t < NACC; ++t)
                        gw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(da2, __builtin_fmaxf(__builtin_fmaf(w1p[t], zr, b1p[t]), 0.0f), gw[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                             // the batch's LDS rows / sign bits are free again
    }

    // ------------------------------------------------------------------------ this workgroup's sums, block layout
    float* out = a.partial + (int64_t)blockIdx.x * NF_BLOCK_FLOATS;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * iblk + (r & 3) + 8 * (r >> 2) + 4 * half, kk = 32 * (NACC * kpart + t) + col;
            out[NF_W2_OFF + kk * NF_H + i] = gw[t][r];
        }
    {
        const float s2 = gb2 + __shfl_xor(gb2, 32, 64);
        if (half == 0 && kpart == 0) {                               // (the waves of the other input-unit parts saw the same d a2)
            const int i = 32 * iblk + col;
            out[NF_V4_OFF + 4 * i] = s2;
            out[NF_V4_OFF + 4 * i + 3] = 0.0f;
        }
    }
    // dW1 / db1 / db3 / dW3: sums over the waves' tiles, through LDS (the weights are no longer needed) in a fixed order
    float* sc = lds + C_W2;                      // [wave][256] dW1 | db1, then [wave][64] db3 parts, then [wave][256] dW3
    constexpr int SC_B3 = 256 * WAVES, SC_W3 = SC_B3 + 64 * WAVES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = col >> 4, r = col & 15;
        sc[SC_W3 + wave * 256 + c * 128 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half] = gw3acc[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float sw = gw1a[t] + __shfl_xor(gw1a[t], 32, 64), sb = gb1a[t] + __shfl_xor(gb1a[t], 32, 64);
        if (half == 0) {
            sc[wave * 256 + 32 * t + col] = sw;
            sc[wave * 256 + 128 + 32 * t + col] = sb;
        }
    }
    if (half == 0) {
        sc[SC_B3 + wave * 64 + col] = gb30;
        sc[SC_B3 + wave * 64 + 32 + col] = gb31;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int j = threadIdx.x;                                   // 0..127: dW1, 128..255: db1
        float v = 0.0f, v3 = 0.0f;
        for (int w = 0; w < WAVES; ++w) {
            v += sc[w * 256 + j];
            v3 += sc[SC_W3 + w * 256 + j];
        }
        out[(j < 128 ? NF_W1_OFF : NF_B1_OFF - 128) + j] = v;
        out[NF_V4_OFF + 4 * (j & 127) + 1 + (j >> 7)] = v3;          // dW3[c = j >> 7][i = j & 127]
    }
    if (threadIdx.x < 2) {
        float v = 0.0f;
        for (int w = 0; w < WAVES; ++w)
            for (int c = 0; c < 32; ++c) v += sc[SC_B3 + w * 64 + 32 * threadIdx.x + c];
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

static int wgs_for(int64_t n_rows, int* rows_per_wg, int max_wgs = 2 * BW_MAX_WGS, int batch_rows = BW_ROWS)
{
    const int64_t batches = (n_rows + batch_rows - 1) / batch_rows;
    const int64_t wgs = batches < max_wgs ? batches : max_wgs;
    const int64_t per = (batches + wgs - 1) / wgs;
    *rows_per_wg = (int)(per * batch_rows);
    return (int)((batches + per - 1) / per);
}

constexpr int BASE_BLOCKS = 256;

struct Workspace {
    float *z, *g, *lq, *trace, *partial;
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
    w->trace = (float*)take((int64_t)n_couplings * n_rows * 4);
    w->partial = (float*)take((int64_t)n_couplings * wgs * NF_BLOCK_FLOATS * 4);
    w->base_partial = (double*)take(BASE_BLOCKS * 5 * 8);
    return at;
}

}  // namespace glabc

using namespace glabc;

extern "C" {

int glabc_nf_inverse(const glabc_flow* flow, const float* x, int64_t n_rows, float* z_out, float* log_q, float* trace, void* stream);

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
    int rc = glabc_nf_inverse(flow, x, n_rows, w.z, w.lq, w.trace, stream);                   // rows -> base space, log_prob, conditioner inputs
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
    // kernel variants: 8 = sign-bit kernel, one workgroup of eight waves per CU (default); 4 = sign-bit kernel, two workgroups of
    // four waves per CU; 1 = the staged kernel (GLABC_NF_BW = 8 | 4 | 1, a measuring knob: same results up to summation order)
    static int variant = 0;
    if (!variant) {
        const char* e = std::getenv("GLABC_NF_BW");
        variant = (e && e[0] == '1') ? 1 : (e && e[0] == '4') ? 4 : 8;
    }
    {
        static LdsGrant grant;                           // per device: a second GPU driven by the same process needs its own grant
        const void* fn = variant == 1 ? (const void*)nf_backward_kernel : variant == 4 ? (const void*)nf_backward_kernel2<4> : (const void*)nf_backward_kernel2<8>;
        const int bytes = (variant == 1 ? B_FLOATS : variant == 4 ? BwLds<4>::FLOATS : BwLds<8>::FLOATS) * 4;
        if (!grant_dynamic_lds(grant, fn, (size_t)bytes, 0)) return GLABC_ERR_LAUNCH;
    }
    int rows_per_wg;
    const int wgs = wgs_for(n_rows, &rows_per_wg, variant == 4 ? 2 * BW_MAX_WGS : BW_MAX_WGS, variant == 8 ? 256 : BW_ROWS);
    for (int c = 0; c < flow->n_couplings; ++c) {                     // log_prob applied n-1 .. 0: the sweep back runs 0 .. n-1
        BwArgs a;
        a.block = flow->params + (int64_t)c * NF_BLOCK_FLOATS;
        a.z1p = c == 0 ? w.z + n_rows : w.trace + (int64_t)(c - 1) * n_rows;
        a.g = w.g;
        a.partial = w.partial + (int64_t)c * wgs * NF_BLOCK_FLOATS;
        a.n_rows = n_rows;
        a.rows_per_wg = rows_per_wg;
        a.gl = (float)gl;
        a.z0_trace = w.trace + (int64_t)c * n_rows;
        if (variant == 1)
            hipLaunchKernelGGL(nf_backward_kernel, dim3(wgs), dim3(64 * BW_WAVES), B_FLOATS * 4, s, a);
        else if (variant == 4)
            hipLaunchKernelGGL(nf_backward_kernel2<4>, dim3(wgs), dim3(256), BwLds<4>::FLOATS * 4, s, a);
        else
            hipLaunchKernelGGL(nf_backward_kernel2<8>, dim3(wgs), dim3(512), BwLds<8>::FLOATS * 4, s, a);
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
