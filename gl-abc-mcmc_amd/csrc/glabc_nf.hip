// glabc_nf.hip -- RealNVP affine-coupling stack of GLMCMC_NF on the gfx950 matrix cores.
//
// Reference: GLMCMC_NFs.py:51-61 builds num_layers x [AffineCouplingBlock(MLP([1,128,128,2])), Permute(2,'swap')]
// with the third-party normflows package and uses NF_model.sample(n) (forward through every coupling,
// GLMCMC_NFs.py:70-72,125-127) and NF_model.log_prob(x) (inverse through every coupling, :96-98).
// normflows is not vendored in the reference tree; its published semantics are restated in DESIGN.md /
// SURVEY.md appendix C.  For theta_dim = 2 one coupling is, per row z = (z0, z1):
//     h1 = relu(W1 z0 + b1)            1 -> 128
//     h2 = relu(W2 h1 + b2)            128 x 128   <-- the only dense contraction on the whole hot path
//     (shift, log_s) = W3 h2 + b3      128 -> 2
//     forward: z1 <- z1*exp(log_s) + shift, log_det = +log_s ; inverse: z1 <- (z1 - shift)*exp(-log_s), log_det = -log_s
//     swap:    z <- (z1, z0)
//
// Mapping.  One wavefront owns a tile of 32 rows.  The 128x128 layer runs on v_mfma_f32_32x32x2_f32 with
// D[i][j] = sum_k A[i][k] B[k][j]:  i = output neuron (4 tiles of 32), j = row, k = input neuron, so
//     A: lane l holds W2[32t + (l&31)][2s + (l>>5)]   one float from the LDS image of W2^T ([k][i], conflict-free)
//     B: lane l holds h1[row l&31][2s + (l>>5)]        one fma + max, made on the fly
//     D: lane l holds, for ITS row (l&31), the 16 neurons i = 32t + (r&3) + 8(r>>2) + 4(l>>5), r = 0..15
// i.e. the two lanes l and l+32 of a row each end up with half of the hidden units, reduce them against W3
// in registers (64-term fmaf chains) and exchange the two partial sums with one ds_swizzle.  f32-input MFMA is
// exact float32 (bit-for-bit a k-ordered fmaf chain, MI355X_MICROARCH.md), so the CPU checker reproduces every
// output bit.  Weights of the current coupling live in LDS (W2^T 64 KiB + 3 KiB of vectors); a workgroup of
// 8 waves -- two per SIMD, so one wave's LDS / VALU gaps are filled by the other's MFMAs (85 -> 100 TFLOP/s) -- pushes
// its rows through a coupling before the next coupling's weights are staged.
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

// The LDS image of a block puts the 3 KiB of vectors FIRST and W2^T behind them: every vector element is then within the
// 16-bit immediate offset of a ds_read from one base register.  (With the global layout the vectors sit at byte 66 560, past
// that range, and the compiler kept one address VGPR per accumulator-init / epilogue read -- the spills of round 1.)
constexpr int L_W1 = 0, L_B1 = NF_H, L_V4 = 2 * NF_H, L_B3 = L_V4 + 4 * NF_H, L_W2 = L_B3 + 4;
static_assert((L_W2 * 4) % 16 == 0 && L_W2 + NF_H * NF_H == NF_BLOCK_FLOATS, "LDS image layout");

#ifndef GLABC_NF_WAVES
#define GLABC_NF_WAVES 12
#endif
#ifndef GLABC_NF_GROUPED
#define GLABC_NF_GROUPED 1     // form 2: a group's eight MFMAs issued back to back behind their operands' vector instructions
#endif
#ifndef GLABC_NF_FORM
#define GLABC_NF_FORM 2        // how a wavefront walks a pair-coupling: 0 k outermost, 1 output tile outermost, 2 = 1 + fenced prefetch
#endif
// waves per workgroup (one workgroup per CU).  12 = three per SIMD: with the rows' state in LDS and the vectors addressed
// from one base register the pair kernel needs 168 VGPRs (no scratch), so three waves fit a SIMD's 512 and one wave's
// LDS / VALU / epilogue gaps are filled by two others' MFMAs: 8 waves 104, 12 waves 112 TFLOP/s (profiles/r02_nf_*)
constexpr int NF_WAVES = GLABC_NF_WAVES;
constexpr int NF_MAX_PAIRS = 5;                               // 64-row pairs a wave keeps in registers, at most
constexpr int NF_CUS = 256;

struct NfArgs {
    const float* params;          // [n_couplings][NF_BLOCK_FLOATS]
    int32_t n_couplings;
    float base_loc[2], base_log_scale[2], base_scale[2], base_c0;
    // inputs / outputs, chain-major [2][n]
    const float* in;              // forward: base noise eps (NULL = Philox);  inverse: x
    float* z_out;                 // forward: samples ; inverse: unused (NULL)
    float* log_q;                 // [n]
    int64_t n_rows, row0;
    uint32_t seed_lo, seed_hi;
    int32_t rows_per_wg;          // multiple of 64 (pairs), of 32 in tile mode
    // indexed inverse (glabc_nf_log_prob_indexed): rows are idx[0 .. *n_dev), inputs in[idx] / in[in_stride + idx]
    const int32_t* idx;
    const int32_t* n_dev;
    int64_t in_stride;
    int32_t state_floats_per_wave; // pair mode: 192 floats (2 tiles x 3 x 32) per pair slot of a wave
    float* trace;                 // inverse only, may be NULL: [n_couplings][n_rows] the conditioner input of every coupling (training)
};

// (shift, log_s) of one coupling for this lane's two rows (one in tile a, one in tile b), conditioner inputs z0a / z0b;
// lds = the staged parameter block.  k runs outermost: h1[k] is made on the fly (one fma + max per tile) and every
// W2^T operand fetched from LDS feeds two MFMAs (tile a and tile b), for each of the four 32-neuron output tiles.  The
// accumulators start at the bias b2, so the pre-activation of hidden unit i is the k-ascending fmaf chain
//   fma(W2[i][127], h1[127], ... fma(W2[i][0], h1[0], b2[i])).
// (Explicit operand prefetch rings and other unroll factors were measured no better than the compiler's schedule.)
__device__ __forceinline__ void coupling_params2(const float* __restrict__ lds, float z0a, float z0b, int lane, float& shift_a,
                                                 float& log_s_a, float& shift_b, float& log_s_b)
{
    const int half = lane >> 5, col = lane & 31;
    f32x16 a0, a1, a2, a3, b0, b1, b2, b3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        a0[r] = b0[r] = lds[L_V4 + 4 * i];
        a1[r] = b1[r] = lds[L_V4 + 4 * (i + 32)];
        a2[r] = b2[r] = lds[L_V4 + 4 * (i + 64)];
        a3[r] = b3[r] = lds[L_V4 + 4 * (i + 96)];
    }
    const float* w1 = lds + L_W1 + half;
    const float* bb1 = lds + L_B1 + half;
    const float* wt = lds + L_W2 + half * NF_H + col;
#pragma unroll 4
    for (int s = 0; s < 64; ++s) {
        const float w = w1[2 * s], bia = bb1[2 * s];
        const float ha = __builtin_fmaxf(__builtin_fmaf(w, z0a, bia), 0.0f);
        const float hb = __builtin_fmaxf(__builtin_fmaf(w, z0b, bia), 0.0f);
        const float* row = wt + 2 * s * NF_H;
        const float w0 = row[0], w1v = row[32], w2v = row[64], w3v = row[96];
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, ha, a0, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, hb, b0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1v, ha, a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1v, hb, b1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2v, ha, a2, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2v, hb, b2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3v, ha, a3, 0, 0, 0);
        b3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3v, hb, b3, 0, 0, 0);
    }
    float pa0 = 0.0f, pa1 = 0.0f, pb0 = 0.0f, pb1 = 0.0f;
    auto epilogue = [&](const f32x16& xa, const f32x16& xb, int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
            const float ha = __builtin_fmaxf(xa[r], 0.0f), hb = __builtin_fmaxf(xb[r], 0.0f);
            pa0 = __builtin_fmaf(v.x, ha, pa0);
            pa1 = __builtin_fmaf(v.y, ha, pa1);
            pb0 = __builtin_fmaf(v.x, hb, pb0);
            pb1 = __builtin_fmaf(v.y, hb, pb1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    epilogue(a0, b0, 0);
    epilogue(a1, b1, 1);
    epilogue(a2, b2, 2);
    epilogue(a3, b3, 3);
    const float qa0 = __shfl_xor(pa0, 32, 64), qa1 = __shfl_xor(pa1, 32, 64);
    const float qb0 = __shfl_xor(pb0, 32, 64), qb1 = __shfl_xor(pb1, 32, 64);
    const float b30 = lds[L_B3 + 0], b31 = lds[L_B3 + 1];
    shift_a = ((half ? qa0 : pa0) + (half ? pa0 : qa0)) + b30;
    log_s_a = ((half ? qa1 : pa1) + (half ? pa1 : qa1)) + b31;
    shift_b = ((half ? qb0 : pb0) + (half ? pb0 : qb0)) + b30;
    log_s_b = ((half ? qb1 : pb1) + (half ? pb1 : qb1)) + b31;
}

// The same arithmetic with the OUTPUT TILE outermost (the default since round 3; 103.4 -> 108.6 TFLOP/s): tile t's 128 MFMAs (k ascending, two rows' tiles
// a / b per A operand) run while the W3 reduction of tile t-1 -- whose accumulators are complete -- is spread over them, one
// hidden unit every fourth k step.  Only the last tile's reduction is exposed (a quarter of the epilogue), and two tiles of
// accumulators are alive instead of four (64 registers instead of 128).  h1 is made four times (once per tile): 2 VALU per
// MFMA pair, in the shadow of the matrix pipe.  Same bits: every accumulator is the same k-ordered chain, the W3 partial
// sums visit the hidden units in the same tile-then-register order.
template <int T>
__device__ __forceinline__ void nf_tile_pass(const float* __restrict__ lds, float z0a, float z0b, int half, int col, f32x16& xa,
                                             f32x16& xb, const f32x16& ya, const f32x16& yb, float& pa0, float& pa1, float& pb0,
                                             float& pb1)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 32 * T + (r & 3) + 8 * (r >> 2) + 4 * half;
        xa[r] = xb[r] = lds[L_V4 + 4 * i];
    }
    const float* w1 = lds + L_W1 + half;
    const float* bb1 = lds + L_B1 + half;
    const float* wt = lds + L_W2 + half * NF_H + col + 32 * T;
    // groups of four k steps (8 MFMAs = 512 cycles of the matrix pipe); the operands of group g + 1 are fetched from LDS at the top
    // of group g, and a scheduling barrier per group keeps the compiler from hoisting all 192 loads of a tile to its head
    float w[4], bi[4], wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w[q] = w1[2 * q];
        bi[q] = bb1[2 * q];
        wv[q] = wt[2 * q * NF_H];
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        float nw[4], nbi[4], nwv[4];
        if (g < 15) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sn = 4 * (g + 1) + q;
                nw[q] = w1[2 * sn];
                nbi[q] = bb1[2 * sn];
                nwv[q] = wt[2 * sn * NF_H];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ha = __builtin_fmaxf(__builtin_fmaf(w[q], z0a, bi[q]), 0.0f);
            const float hb = __builtin_fmaxf(__builtin_fmaf(w[q], z0b, bi[q]), 0.0f);
            xa = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q], ha, xa, 0, 0, 0);
            xb = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[q], hb, xb, 0, 0, 0);
        }
        if constexpr (T > 0) {                                         // hidden unit g of the PREVIOUS tile, in register order
            const int i = 32 * (T - 1) + (g & 3) + 8 * (g >> 2) + 4 * half;
            const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
            const float qa = __builtin_fmaxf(ya[g], 0.0f), qb = __builtin_fmaxf(yb[g], 0.0f);
            pa0 = __builtin_fmaf(v.x, qa, pa0);
            pa1 = __builtin_fmaf(v.y, qa, pa1);
            pb0 = __builtin_fmaf(v.x, qb, pb0);
            pb1 = __builtin_fmaf(v.y, qb, pb1);
        }
        if (g < 15) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = nw[q];
                bi[q] = nbi[q];
                wv[q] = nwv[q];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void coupling_params2_tile_outer(const float* __restrict__ lds, float z0a, float z0b, int lane, float& shift_a,
                                                            float& log_s_a, float& shift_b, float& log_s_b)
{
    const int half = lane >> 5, col = lane & 31;
    f32x16 a0, b0, a1, b1, a2, b2, a3, b3;
    float pa0 = 0.0f, pa1 = 0.0f, pb0 = 0.0f, pb1 = 0.0f;
    nf_tile_pass<0>(lds, z0a, z0b, half, col, a0, b0, a0, b0, pa0, pa1, pb0, pb1);
    nf_tile_pass<1>(lds, z0a, z0b, half, col, a1, b1, a0, b0, pa0, pa1, pb0, pb1);
    nf_tile_pass<2>(lds, z0a, z0b, half, col, a2, b2, a1, b1, pa0, pa1, pb0, pb1);
    nf_tile_pass<3>(lds, z0a, z0b, half, col, a3, b3, a2, b2, pa0, pa1, pb0, pb1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                     // the last tile's reduction
        const int i = 96 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
        const float qa = __builtin_fmaxf(a3[r], 0.0f), qb = __builtin_fmaxf(b3[r], 0.0f);
        pa0 = __builtin_fmaf(v.x, qa, pa0);
        pa1 = __builtin_fmaf(v.y, qa, pa1);
        pb0 = __builtin_fmaf(v.x, qb, pb0);
        pb1 = __builtin_fmaf(v.y, qb, pb1);
    }
    const float qa0 = __shfl_xor(pa0, 32, 64), qa1 = __shfl_xor(pa1, 32, 64);
    const float qb0 = __shfl_xor(pb0, 32, 64), qb1 = __shfl_xor(pb1, 32, 64);
    const float b30 = lds[L_B3 + 0], b31 = lds[L_B3 + 1];
    shift_a = ((half ? qa0 : pa0) + (half ? pa0 : qa0)) + b30;
    log_s_a = ((half ? qa1 : pa1) + (half ? pa1 : qa1)) + b31;
    shift_b = ((half ? qb0 : pb0) + (half ? pb0 : qb0)) + b30;
    log_s_b = ((half ? qb1 : pb1) + (half ? pb1 : qb1)) + b31;
}

// The tile-outer form with the operand fetch REALLY one group ahead (the default since round 3b).  In nf_tile_pass the source
// asks for group g + 1's operands at the top of group g, but nothing holds the compiler to it: it sinks the ds_reads to the end of
// the group, next to their first use, and the wavefront then sits in s_waitcnt for an LDS round trip at the head of EVERY group
// of 8 MFMAs (19 % of the wave cycles parked, profiles/r03b_nf_wait.txt) -- covered only as far as the SIMD's other wavefronts
// happen to have MFMAs queued.  Here the fetch is fenced in by scheduling barriers on both sides, and everything a group needs
// from LDS comes that way: the four k steps' W1 / b1 / W2^T operands, the W3 pair of the hidden unit the group reduces, and
// -- one register per group, into the registers the first MFMAs of the tile have just freed -- the next tile's b2, which
// initialises the accumulators as the C operand of that tile's first MFMA pair instead of through sixteen exposed reads and
// moves.  The last group of a tile fetches the first group of the next one; after tile 3 that is tile 0 of the wavefront's NEXT
// pair (same coupling, same weights), so a wavefront fetches operands on the critical path once per coupling, not per group.
// Same bits: every accumulator is the same k-ordered chain from b2, the W3 sums visit the hidden units in the same order.
struct NfOps {
    float w[4], bi[4], wv[4];     // W1[k], b1[k], W2^T[k][this lane's output unit] of the group's four k steps (k = 2 sn + half)
    float2 v;                     // (W3[0][i], W3[1][i]) of the previous tile's hidden unit this group reduces
};

__device__ __forceinline__ void nf_fetch_ops(const float* __restrict__ lds, int half, int col, int t, int g, NfOps& o)
{
    const float* w1 = lds + L_W1 + half;
    const float* bb1 = lds + L_B1 + half;
    const float* wt = lds + L_W2 + half * NF_H + col + 32 * t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sn = 4 * g + q;
        o.w[q] = w1[2 * sn];
        o.bi[q] = bb1[2 * sn];
        o.wv[q] = wt[2 * sn * NF_H];
    }
    if (t > 0) {
        const int i = 32 * (t - 1) + (g & 3) + 8 * (g >> 2) + 4 * half;
        o.v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
    }
}

__device__ __forceinline__ void nf_fetch_bias(const float* __restrict__ lds, int half, int t, f32x16& c)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = lds[L_V4 + 4 * (32 * t + (r & 3) + 8 * (r >> 2) + 4 * half)];
}

template <int T>
__device__ __forceinline__ void nf_tile_pass_ahead(const float* __restrict__ lds, float z0a, float z0b, int half, int col, f32x16& xa,
                                                   f32x16& xb, const f32x16& ya, const f32x16& yb, NfOps& cur, f32x16& c, float& pa0,
                                                   float& pa1, float& pb0, float& pb1)
{
    constexpr int TN = (T + 1) & 3;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        NfOps nxt;
        if (g < 15) nf_fetch_ops(lds, half, col, T, g + 1, nxt);
        else nf_fetch_ops(lds, half, col, TN, 0, nxt);
        f32x16 cn = c;
        if (g >= 1) {                                                  // the next tile's b2: registers 0 and 15 in group 1, g in group g
            const int r = g == 1 ? 0 : g;
            cn[r] = lds[L_V4 + 4 * (32 * TN + (r & 3) + 8 * (r >> 2) + 4 * half)];
            if (g == 1) cn[1] = lds[L_V4 + 4 * (32 * TN + 1 + 4 * half)];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the group's eight B operands first, then its eight MFMAs back to back: a vector instruction between two MFMAs of one
        // wavefront costs the matrix pipe ~12 cycles for the switch, whatever it is (tools/ubench/mfma_valu_fill.hip: 8 MFMAs then
        // their 24 v_fma 133 TFLOP/s against 129 interleaved one to three, two wavefronts per SIMD)
        float ha[4], hb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ha[q] = __builtin_fmaxf(__builtin_fmaf(cur.w[q], z0a, cur.bi[q]), 0.0f);
            hb[q] = __builtin_fmaxf(__builtin_fmaf(cur.w[q], z0b, cur.bi[q]), 0.0f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (g == 0 && q == 0) {
                xa = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], ha[q], c, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], hb[q], c, 0, 0, 0);
            } else {
                xa = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], ha[q], xa, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], hb[q], xb, 0, 0, 0);
            }
        }
#if GLABC_NF_GROUPED
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);            // VALU: the eight fma + max
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);             // MFMA
#endif
        if constexpr (T > 0) {                                         // hidden unit g of the PREVIOUS tile, in register order
            const float qa = __builtin_fmaxf(ya[g], 0.0f), qb = __builtin_fmaxf(yb[g], 0.0f);
            pa0 = __builtin_fmaf(cur.v.x, qa, pa0);
            pa1 = __builtin_fmaf(cur.v.y, qa, pa1);
            pb0 = __builtin_fmaf(cur.v.x, qb, pb0);
            pb1 = __builtin_fmaf(cur.v.y, qb, pb1);
        }
        cur = nxt;
        c = cn;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ops / c: in = group 0 of tile 0 and b2 of tile 0 (nf_fetch_ops / nf_fetch_bias at the head of a coupling); out = the same again, for
// the wavefront's next pair of this coupling
__device__ __forceinline__ void coupling_params2_ahead(const float* __restrict__ lds, float z0a, float z0b, int lane, NfOps& ops, f32x16& c,
                                                       float& shift_a, float& log_s_a, float& shift_b, float& log_s_b)
{
    const int half = lane >> 5, col = lane & 31;
    f32x16 a0, b0, a1, b1, a2, b2, a3, b3;
    float pa0 = 0.0f, pa1 = 0.0f, pb0 = 0.0f, pb1 = 0.0f;
    nf_tile_pass_ahead<0>(lds, z0a, z0b, half, col, a0, b0, a0, b0, ops, c, pa0, pa1, pb0, pb1);
    nf_tile_pass_ahead<1>(lds, z0a, z0b, half, col, a1, b1, a0, b0, ops, c, pa0, pa1, pb0, pb1);
    nf_tile_pass_ahead<2>(lds, z0a, z0b, half, col, a2, b2, a1, b1, ops, c, pa0, pa1, pb0, pb1);
    nf_tile_pass_ahead<3>(lds, z0a, z0b, half, col, a3, b3, a2, b2, ops, c, pa0, pa1, pb0, pb1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                     // the last tile's reduction
        const int i = 96 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
        const float qa = __builtin_fmaxf(a3[r], 0.0f), qb = __builtin_fmaxf(b3[r], 0.0f);
        pa0 = __builtin_fmaf(v.x, qa, pa0);
        pa1 = __builtin_fmaf(v.y, qa, pa1);
        pb0 = __builtin_fmaf(v.x, qb, pb0);
        pb1 = __builtin_fmaf(v.y, qb, pb1);
    }
    const float qa0 = __shfl_xor(pa0, 32, 64), qa1 = __shfl_xor(pa1, 32, 64);
    const float qb0 = __shfl_xor(pb0, 32, 64), qb1 = __shfl_xor(pb1, 32, 64);
    const float b30 = lds[L_B3 + 0], b31 = lds[L_B3 + 1];
    shift_a = ((half ? qa0 : pa0) + (half ? pa0 : qa0)) + b30;
    log_s_a = ((half ? qa1 : pa1) + (half ? pa1 : qa1)) + b31;
    shift_b = ((half ? qb0 : pb0) + (half ? pb0 : qb0)) + b30;
    log_s_b = ((half ? qb1 : pb1) + (half ? pb1 : qb1)) + b31;
}

// The same for ONE row tile per wavefront (tile mode: inputs of at most eight tiles per workgroup): groups of four k steps = four
// MFMAs behind their four B operands, operands fetched one group ahead, b2 as the C operand of a tile's first MFMA.
template <int T>
__device__ __forceinline__ void nf_tile_pass_ahead1(const float* __restrict__ lds, float z0, int half, int col, f32x16& x, const f32x16& y,
                                                    NfOps& cur, f32x16& c, float& p0, float& p1)
{
    constexpr int TN = (T + 1) & 3;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        NfOps nxt;
        if (g < 15) nf_fetch_ops(lds, half, col, T, g + 1, nxt);
        else nf_fetch_ops(lds, half, col, TN, 0, nxt);
        f32x16 cn = c;
        if (g >= 1) {
            const int r = g == 1 ? 0 : g;
            cn[r] = lds[L_V4 + 4 * (32 * TN + (r & 3) + 8 * (r >> 2) + 4 * half)];
            if (g == 1) cn[1] = lds[L_V4 + 4 * (32 * TN + 1 + 4 * half)];
        }
        __builtin_amdgcn_sched_barrier(0);
        float h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = __builtin_fmaxf(__builtin_fmaf(cur.w[q], z0, cur.bi[q]), 0.0f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (g == 0 && q == 0) x = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], h[q], c, 0, 0, 0);
            else x = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.wv[q], h[q], x, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if constexpr (T > 0) {
            const float q2 = __builtin_fmaxf(y[g], 0.0f);
            p0 = __builtin_fmaf(cur.v.x, q2, p0);
            p1 = __builtin_fmaf(cur.v.y, q2, p1);
        }
        cur = nxt;
        c = cn;
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void coupling_params_ahead(const float* __restrict__ lds, float z0, int lane, float& shift, float& log_s)
{
    const int half = lane >> 5, col = lane & 31;
    NfOps ops;
    f32x16 c;
    nf_fetch_ops(lds, half, col, 0, 0, ops);
    nf_fetch_bias(lds, half, 0, c);
    f32x16 a0, a1, a2, a3;
    float p0 = 0.0f, p1 = 0.0f;
    nf_tile_pass_ahead1<0>(lds, z0, half, col, a0, a0, ops, c, p0, p1);
    nf_tile_pass_ahead1<1>(lds, z0, half, col, a1, a0, ops, c, p0, p1);
    nf_tile_pass_ahead1<2>(lds, z0, half, col, a2, a1, ops, c, p0, p1);
    nf_tile_pass_ahead1<3>(lds, z0, half, col, a3, a2, ops, c, p0, p1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 96 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);
        const float h2 = __builtin_fmaxf(a3[r], 0.0f);
        p0 = __builtin_fmaf(v.x, h2, p0);
        p1 = __builtin_fmaf(v.y, h2, p1);
    }
    const float q0 = __shfl_xor(p0, 32, 64), q1 = __shfl_xor(p1, 32, 64);
    const float lo0 = half ? q0 : p0, hi0 = half ? p0 : q0;
    const float lo1 = half ? q1 : p1, hi1 = half ? p1 : q1;
    shift = (lo0 + hi0) + lds[L_B3 + 0];
    log_s = (lo1 + hi1) + lds[L_B3 + 1];
}

// One row tile per wavefront: the form for small inputs (at most one tile per CU), where the launch is a latency chain
// and half the MFMAs per coupling beat operand reuse.  Same arithmetic per row as coupling_params2.
__device__ __forceinline__ void coupling_params(const float* __restrict__ lds, float z0, int lane, float& shift, float& log_s)
{
    const int half = lane >> 5, col = lane & 31;
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        acc0[r] = lds[L_V4 + 4 * i];
        acc1[r] = lds[L_V4 + 4 * (i + 32)];
        acc2[r] = lds[L_V4 + 4 * (i + 64)];
        acc3[r] = lds[L_V4 + 4 * (i + 96)];
    }
    const float* w1 = lds + L_W1 + half;
    const float* b1 = lds + L_B1 + half;
    const float* wt = lds + L_W2 + half * NF_H + col;                // W2^T[2s + half][32t + col]
#pragma unroll 8
    for (int s = 0; s < 64; ++s) {
        const float h1 = __builtin_fmaxf(__builtin_fmaf(w1[2 * s], z0, b1[2 * s]), 0.0f);
        const float* row = wt + 2 * s * NF_H;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[0], h1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32], h1, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[64], h1, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[96], h1, acc3, 0, 0, 0);
    }
    float p0 = 0.0f, p1 = 0.0f;
    auto epilogue = [&](const f32x16& acc, int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float2 v = *reinterpret_cast<const float2*>(lds + L_V4 + 4 * i + 1);     // (W3[0][i], W3[1][i])
            const float h2 = __builtin_fmaxf(acc[r], 0.0f);
            p0 = __builtin_fmaf(v.x, h2, p0);
            p1 = __builtin_fmaf(v.y, h2, p1);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep the next tile's LDS reads from being hoisted above (VGPR pressure)
    };
    epilogue(acc0, 0);
    epilogue(acc1, 1);
    epilogue(acc2, 2);
    epilogue(acc3, 3);
    // partial sums of the two halves of the row: lane l <-> l + 32
    const float q0 = __shfl_xor(p0, 32, 64), q1 = __shfl_xor(p1, 32, 64);
    const float lo0 = half ? q0 : p0, hi0 = half ? p0 : q0;
    const float lo1 = half ? q1 : p1, hi1 = half ? p1 : q1;
    shift = (lo0 + hi0) + lds[L_B3 + 0];
    log_s = (lo1 + hi1) + lds[L_B3 + 1];
}

// Initial state of one row: forward = the base distribution's draw (nf.distributions.base.DiagGaussian.forward: z = loc +
// exp(log_scale)*eps, log_p = C - sum(log_scale + 0.5 eps^2)), inverse = the point itself with log_q = 0.
template <bool INVERSE>
__device__ __forceinline__ void nf_row_init(const NfArgs& a, int64_t n_rows, int64_t row, float& z0, float& z1, float& lq)
{
    const int64_t rr = row < n_rows ? row : n_rows - 1;
    if (INVERSE) {
        const int64_t src = a.idx ? (int64_t)a.idx[rr] : rr;
        z0 = a.in[src];
        z1 = a.in[a.in_stride + src];
        lq = 0.0f;
        return;
    }
    float e0, e1;
    if (a.in) {
        e0 = a.in[rr];
        e1 = a.in[a.n_rows + rr];
    } else {
        const uint64_t gid = (uint64_t)(a.row0 + rr);
        glabc_u32x4 w = glabc_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), 0u, 0u, a.seed_lo, a.seed_hi);
        glabc_normal_pair(w.v[0], w.v[1], &e0, &e1);
    }
    z0 = a.base_loc[0] + a.base_scale[0] * e0;
    z1 = a.base_loc[1] + a.base_scale[1] * e1;
    lq = a.base_c0 - ((a.base_log_scale[0] + 0.5f * (e0 * e0)) + (a.base_log_scale[1] + 0.5f * (e1 * e1)));
}

template <bool INVERSE>
__device__ __forceinline__ void nf_row_apply(float shift, float log_s, float& z0, float& z1, float& lq)
{
    if (INVERSE) {                             // flows reversed: Permute^-1 (the swap again) first, conditioner input z1
        const float t0 = z1, t1 = z0;
        z0 = t0;
        z1 = (t1 - shift) * glabc_expf_b(-log_s);
        lq = lq + (-log_s);
    } else {
        const float nz = z1 * glabc_expf_b(log_s) + shift;
        lq = lq - log_s;                       // log_q -= log_det
        z1 = z0;                               // Permute(2, 'swap')
        z0 = nz;
    }
}

template <bool INVERSE>
__device__ __forceinline__ void nf_row_store(const NfArgs& a, int64_t row, float z0, float z1, float lq)
{
    if (INVERSE) {
        // + q0.log_prob(z): C - sum(log_scale + 0.5 ((z - loc)/exp(log_scale))^2)
        const float e0 = (z0 - a.base_loc[0]) / a.base_scale[0], e1 = (z1 - a.base_loc[1]) / a.base_scale[1];
        const float lp = a.base_c0 - ((a.base_log_scale[0] + 0.5f * (e0 * e0)) + (a.base_log_scale[1] + 0.5f * (e1 * e1)));
        a.log_q[a.idx ? (int64_t)a.idx[row] : row] = lq + lp;
        if (a.z_out) {                         // glabc_nf_inverse: the base-space point as well (training, glabc_nf_train.hip)
            a.z_out[row] = z0;
            a.z_out[a.n_rows + row] = z1;
        }
    } else {
        a.z_out[row] = z0;
        a.z_out[a.n_rows + row] = z1;
        a.log_q[row] = lq;
    }
}

// PAIR mode (TILE_MODE false): the workgroup's 64-row pairs are dealt round-robin to its waves (pair p -> wave p % NF_WAVES)
// and a wave walks its pairs coupling by coupling.  The rows' running state (z0, z1, log_q: 3 floats per row) lives in LDS
// behind the parameter image -- the 128 accumulator registers + operands of coupling_params2 leave no room for it in the 256
// VGPRs of a wave at two waves per SIMD (it used to be spilled to scratch), and six ds_read / ds_write per 512 MFMAs are free.
// TILE mode: one 32-row tile per wave, state in registers (the form for inputs of at most one tile per CU).
template <bool INVERSE, bool TILE_MODE>
__global__ void __launch_bounds__(64 * NF_WAVES) nf_kernel(const NfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31;
    const int64_t wg_row0 = (int64_t)blockIdx.x * a.rows_per_wg;
    const int64_t n_rows = a.n_dev ? (int64_t)*a.n_dev : a.n_rows;     // indexed mode: the row count lives on the device
    if (wg_row0 >= n_rows) return;                                     // (workgroup-uniform, before any barrier)
    const int n_pairs = a.rows_per_wg / 64;                            // pair mode: 64-row pairs in this workgroup
    float* st = lds + NF_BLOCK_FLOATS;                                 // pair mode: [pair][tile a/b][z0, z1, lq][32 rows]
    // Pair mode: which pairs are this wavefront's.  The matrix pipe belongs to a SIMD, so the pairs are dealt to the SIMDs first
    // (pair p -> SIMD p % 4) and only then to the wavefronts that happen to sit on that SIMD (HW_ID), whatever order the
    // dispatcher placed them in: with 20 pairs on 12 or 16 wavefronts a deal by wavefront index leaves one SIMD with 6 .. 8
    // pairs and another with 3 .. 4, and the workgroup waits for the fullest.  Falls back to the wavefront index if a SIMD
    // holds none of the workgroup's wavefronts.
    int p_first = wave, p_stride = NF_WAVES;
    if constexpr (!TILE_MODE) {
        __shared__ int simd_waves[4];
        if (threadIdx.x < 4) simd_waves[threadIdx.x] = 0;
        __syncthreads();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int simd = (int)((hw >> 4) & 3u);
        int rank = 0;
        if (lane == 0) rank = atomicAdd(&simd_waves[simd], 1);
        rank = __shfl(rank, 0, 64);
        __syncthreads();
        const int n0 = simd_waves[0], n1 = simd_waves[1], n2 = simd_waves[2], n3 = simd_waves[3];
        if (n0 > 0 && n1 > 0 && n2 > 0 && n3 > 0) {
            const int mine = simd == 0 ? n0 : simd == 1 ? n1 : simd == 2 ? n2 : n3;
            p_first = simd + 4 * rank;
            p_stride = 4 * mine;
        }
    }

    float z0 = 0.0f, z1 = 0.0f, lq = 0.0f;                             // tile mode only
    const bool tile_active = wave * 32 < a.rows_per_wg;                // wave-uniform
    if constexpr (TILE_MODE) {
        if (tile_active) nf_row_init<INVERSE>(a, n_rows, wg_row0 + (int64_t)wave * 32 + col, z0, z1, lq);
    } else {
        for (int p = p_first; p < n_pairs; p += p_stride) {
            // lane l initialises row 64p + l of the pair: tile a = lanes 0..31, tile b = lanes 32..63
            float i0, i1, il;
            nf_row_init<INVERSE>(a, n_rows, wg_row0 + (int64_t)p * 64 + lane, i0, i1, il);
            float* s = st + p * 192 + (lane >> 5) * 96 + col;
            s[0] = i0;
            s[32] = i1;
            s[64] = il;
        }
    }

    // A coupling's parameter block reaches LDS through registers, one coupling AHEAD: each work-item holds its six float4 of the
    // NEXT block while the workgroup computes on the current one (the loads are issued right after the barrier that publishes the
    // current block and have a whole coupling to arrive), so the boundary between two couplings costs two barriers and 67 KiB of
    // ds_write instead of an L2 round trip with every wavefront idle.  (A second LDS image filled by LDS-DMA was tried in round 2:
    // no faster then, and it takes the LDS a deeper pair queue needs.)
    constexpr int W2_VEC = NF_H * NF_H / 4, BLOCK_VEC = NF_BLOCK_FLOATS / 4, SMALL_VEC = BLOCK_VEC - W2_VEC;
    constexpr int PRE = (BLOCK_VEC + 64 * NF_WAVES - 1) / (64 * NF_WAVES);
    static_assert(PRE <= 9, "prefetch registers");
    float4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7, pre8;        // named, not an array: an array indexed in (unrolled) loops stayed in scratch
    (void)pre6; (void)pre7; (void)pre8;
#define GLABC_NF_PRE_ONE(J_, R_)                                                    \
    if constexpr (PRE > J_) {                                                       \
        const int i_ = (int)threadIdx.x + J_ * 64 * NF_WAVES;                       \
        R_ = src_[i_ < BLOCK_VEC ? i_ : BLOCK_VEC - 1];                             \
    }
#define GLABC_NF_PREFETCH_BLOCK(C_)                                                                         \
    {                                                                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(a.params + (int64_t)(C_) * NF_BLOCK_FLOATS);   \
        GLABC_NF_PRE_ONE(0, pre0) GLABC_NF_PRE_ONE(1, pre1) GLABC_NF_PRE_ONE(2, pre2) GLABC_NF_PRE_ONE(3, pre3) GLABC_NF_PRE_ONE(4, pre4) \
        GLABC_NF_PRE_ONE(5, pre5) GLABC_NF_PRE_ONE(6, pre6) GLABC_NF_PRE_ONE(7, pre7) GLABC_NF_PRE_ONE(8, pre8)                     \
    }
#define GLABC_NF_COMMIT_ONE(J_, R_)                                                                         \
    if constexpr (PRE > J_) {                                                                               \
        const int i_ = (int)threadIdx.x + J_ * 64 * NF_WAVES; /* global order: W2^T | W1 | b1 | V4 | b3; the LDS image has W2^T last */ \
        if (i_ < BLOCK_VEC) dst_[i_ < W2_VEC ? L_W2 / 4 + i_ : i_ - W2_VEC] = R_;                           \
    }
    GLABC_NF_PREFETCH_BLOCK(INVERSE ? a.n_couplings - 1 : 0)
    for (int cc = 0; cc < a.n_couplings; ++cc) {
        const int c = INVERSE ? (a.n_couplings - 1 - cc) : cc;
        __syncthreads();                                               // everyone is done with the previous block
        {
            float4* dst_ = reinterpret_cast<float4*>(lds);
            GLABC_NF_COMMIT_ONE(0, pre0) GLABC_NF_COMMIT_ONE(1, pre1) GLABC_NF_COMMIT_ONE(2, pre2) GLABC_NF_COMMIT_ONE(3, pre3)
            GLABC_NF_COMMIT_ONE(4, pre4) GLABC_NF_COMMIT_ONE(5, pre5) GLABC_NF_COMMIT_ONE(6, pre6) GLABC_NF_COMMIT_ONE(7, pre7)
            GLABC_NF_COMMIT_ONE(8, pre8)
        }
        __syncthreads();
        if (cc + 1 < a.n_couplings) GLABC_NF_PREFETCH_BLOCK(INVERSE ? c - 1 : c + 1)
        if constexpr (TILE_MODE) {
            if (tile_active) {
                float sh, ls;
#if GLABC_NF_FORM == 2
                coupling_params_ahead(lds, INVERSE ? z1 : z0, lane, sh, ls);
#else
                coupling_params(lds, INVERSE ? z1 : z0, lane, sh, ls);
#endif
                if (INVERSE && a.trace && lane < 32) {
                    const int64_t row = wg_row0 + (int64_t)wave * 32 + col;
                    if (row < n_rows) a.trace[(int64_t)c * a.n_rows + row] = z1;
                }
                nf_row_apply<INVERSE>(sh, ls, z0, z1, lq);
            }
        } else {
#if GLABC_NF_FORM == 2
            NfOps ops;                                                 // group 0 of tile 0 and its b2: fetched once per coupling,
            f32x16 cinit;                                              // handed from pair to pair (coupling_params2_ahead)
            if (p_first < n_pairs) {
                nf_fetch_ops(lds, lane >> 5, col, 0, 0, ops);
                nf_fetch_bias(lds, lane >> 5, 0, cinit);
            }
#endif
            for (int p = p_first; p < n_pairs; p += p_stride) {
                float* s = st + p * 192 + col;                         // both lanes of a row read the same words
                float a0 = s[0], a1 = s[32], al = s[64], b0 = s[96], b1 = s[128], bl = s[160];
                float sa, la, sb, lb;
#if GLABC_NF_FORM == 0         // the round-2 form (k outermost, all four tiles' accumulators alive): an A/B knob
                coupling_params2(lds, INVERSE ? a1 : a0, INVERSE ? b1 : b0, lane, sa, la, sb, lb);
#elif GLABC_NF_FORM == 1       // output tile outermost, operand fetch left to the compiler (round 3a)
                coupling_params2_tile_outer(lds, INVERSE ? a1 : a0, INVERSE ? b1 : b0, lane, sa, la, sb, lb);
#else
                coupling_params2_ahead(lds, INVERSE ? a1 : a0, INVERSE ? b1 : b0, lane, ops, cinit, sa, la, sb, lb);
#endif
                if (INVERSE && a.trace) {                              // lane l: row 64 p + l of the pair (tile a | tile b)
                    const int64_t row = wg_row0 + (int64_t)p * 64 + lane;
                    if (row < n_rows) a.trace[(int64_t)c * a.n_rows + row] = lane < 32 ? a1 : b1;
                }
                nf_row_apply<INVERSE>(sa, la, a0, a1, al);
                nf_row_apply<INVERSE>(sb, lb, b0, b1, bl);
                if (lane < 32) {
                    s[0] = a0;
                    s[32] = a1;
                    s[64] = al;
                    s[96] = b0;
                    s[128] = b1;
                    s[160] = bl;
                }
            }
        }
    }

    if constexpr (TILE_MODE) {
        const int64_t row = wg_row0 + (int64_t)wave * 32 + col;
        if (tile_active && row < n_rows && lane < 32) nf_row_store<INVERSE>(a, row, z0, z1, lq);
    } else {
        // the lane-derived offsets are recomputed from the thread id here (laundered through an empty asm) instead of
        // being kept alive -- or spilled -- across the coupling loop
        int tid = threadIdx.x;
        __asm__ volatile("" : "+v"(tid));
        const int lane2 = tid & 63;
        const float* st2 = lds + NF_BLOCK_FLOATS;
        for (int p = p_first; p < n_pairs; p += p_stride) {
            const int64_t row = wg_row0 + (int64_t)p * 64 + lane2;
            const float* s = st2 + p * 192 + (lane2 >> 5) * 96 + (lane2 & 31);
            if (row < n_rows) nf_row_store<INVERSE>(a, row, s[0], s[32], s[64]);
        }
    }
}

}  // namespace glabc

using namespace glabc;

static int nf_check(const glabc_flow* f, const float* io, float* log_q, int64_t n)
{
    if (!f || !f->params || !log_q) return GLABC_ERR_NULL;
    (void)io;
    if (f->n_couplings < 1 || f->n_couplings > 4096 || f->hidden != NF_H) return GLABC_ERR_ARG;
    if (n < 0) return GLABC_ERR_ARG;
    for (int j = 0; j < 2; ++j)
        if (!std::isfinite(f->base_loc[j]) || !std::isfinite(f->base_log_scale[j]) || !(f->base_scale[j] > 0.0f)) return GLABC_ERR_ARG;
    return GLABC_OK;
}

static NfArgs nf_pack(const glabc_flow* f, const float* in, float* z, float* log_q, int64_t n, uint64_t seed, int64_t row0)
{
    NfArgs a;
    std::memset(&a, 0, sizeof a);
    a.params = f->params;
    a.n_couplings = f->n_couplings;
    for (int j = 0; j < 2; ++j) {
        a.base_loc[j] = f->base_loc[j];
        a.base_log_scale[j] = f->base_log_scale[j];
        a.base_scale[j] = f->base_scale[j];
    }
    a.base_c0 = f->base_c0;
    a.in = in;
    a.z_out = z;
    a.log_q = log_q;
    a.n_rows = n;
    a.in_stride = n;
    a.row0 = row0;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    return a;
}

template <bool INV, bool TILE>
static int nf_launch_mode(NfArgs a, int rows_per_wg, int slots_per_wave, hipStream_t s)
{
    a.rows_per_wg = rows_per_wg;
    a.state_floats_per_wave = 192 * slots_per_wave;
    const size_t lds_bytes = sizeof(float) * ((size_t)NF_BLOCK_FLOATS + (size_t)NF_WAVES * a.state_floats_per_wave);
    static LdsGrant grant;                               // the largest dynamic LDS size this instantiation was allowed so far, per device
    if (!grant_dynamic_lds(grant, (const void*)nf_kernel<INV, TILE>, lds_bytes, 0)) return GLABC_ERR_LAUNCH;
    const unsigned grid = (unsigned)((a.n_rows + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL((nf_kernel<INV, TILE>), dim3(grid), dim3(64 * NF_WAVES), lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? GLABC_OK : GLABC_ERR_LAUNCH;
}

// Rows per workgroup: the rows are spread over (a multiple of) the 256 CUs in 64-row pairs, one workgroup per CU,
// so the chip is filled evenly and each workgroup stages every coupling once; a wave holds at most NF_MAX_PAIRS pairs.
template <bool INV>
static int nf_launch(const NfArgs& a, hipStream_t s)
{
    const int64_t tiles = (a.n_rows + 31) / 32;
    if (const char* force = std::getenv("GLABC_NF_TILE_WAVES")) {          // experiment knob: tile mode, this many tiles per workgroup
        const int w = std::atoi(force);
        if (w >= 1 && w <= NF_WAVES) return nf_launch_mode<INV, true>(a, 32 * w, 0, s);
    }
    // Up to four tiles per CU (32 768 rows) the launch is a latency chain of couplings and what counts is the longest queue of
    // MFMAs on any SIMD: one tile per wavefront, one wavefront per SIMD (256 MFMAs per coupling) against one or two 64-row pairs
    // on one or two of the four SIMDs (512 each) -- 0.105 against 0.16 ms at 16 384 and 32 768 rows.  From three pairs per CU on
    // the pair form wins again (its A operands feed two MFMAs): 49 152 rows 0.166 against 0.185, 65 536 rows 0.168 against 0.186
    // (profiles/r03c_nf_tile_ab.txt).
    if (tiles <= 4 * (int64_t)NF_CUS) return nf_launch_mode<INV, true>(a, 32 * (int)((tiles + NF_CUS - 1) / NF_CUS), 0, s);
    const int64_t pairs = (a.n_rows + 63) / 64;
    int64_t pairs_per_wg = (pairs + NF_CUS - 1) / NF_CUS;
    const int64_t cap = (int64_t)NF_WAVES * NF_MAX_PAIRS;
    if (pairs_per_wg > cap) {                                         // several rounds of workgroups per CU
        const int64_t rounds = (pairs_per_wg + cap - 1) / cap;
        pairs_per_wg = (pairs + NF_CUS * rounds - 1) / (NF_CUS * rounds);
    }
    const int slots = (int)((pairs_per_wg + NF_WAVES - 1) / NF_WAVES);
    return nf_launch_mode<INV, false>(a, (int)pairs_per_wg * 64, slots, s);
}

extern "C" {

__attribute__((visibility("default"))) int glabc_nf_sample(const glabc_flow* flow, const float* eps, uint64_t seed, int64_t row0,
                                                           int64_t n_rows, float* z_out, float* log_q, void* stream)
{
    int rc = nf_check(flow, eps, log_q, n_rows);
    if (rc) return rc;
    if (!z_out) return GLABC_ERR_NULL;
    if (n_rows == 0) return GLABC_OK;
    return nf_launch<false>(nf_pack(flow, eps, z_out, log_q, n_rows, seed, row0), (hipStream_t)stream);
}

__attribute__((visibility("default"))) int glabc_nf_log_prob(const glabc_flow* flow, const float* x, int64_t n_rows, float* log_q,
                                                             void* stream)
{
    int rc = nf_check(flow, x, log_q, n_rows);
    if (rc) return rc;
    if (!x) return GLABC_ERR_NULL;
    if (n_rows == 0) return GLABC_OK;
    return nf_launch<true>(nf_pack(flow, x, nullptr, log_q, n_rows, 0, 0), (hipStream_t)stream);
}

__attribute__((visibility("default"))) int glabc_nf_inverse(const glabc_flow* flow, const float* x, int64_t n_rows, float* z_out,
                                                            float* log_q, float* trace, void* stream)
{
    int rc = nf_check(flow, x, log_q, n_rows);
    if (rc) return rc;
    if (!x || !z_out) return GLABC_ERR_NULL;
    if (n_rows == 0) return GLABC_OK;
    NfArgs a = nf_pack(flow, x, z_out, log_q, n_rows, 0, 0);
    a.trace = trace;
    return nf_launch<true>(a, (hipStream_t)stream);
}

__attribute__((visibility("default"))) int glabc_nf_log_prob_indexed(const glabc_flow* flow, const float* theta, int64_t stride,
                                                                     const int32_t* idx, const int32_t* n_dev, int64_t max_rows,
                                                                     float* log_q, void* stream)
{
    int rc = nf_check(flow, theta, log_q, max_rows);
    if (rc) return rc;
    if (!theta || !idx || !n_dev) return GLABC_ERR_NULL;
    if (stride < max_rows) return GLABC_ERR_ARG;
    if (max_rows == 0) return GLABC_OK;
    NfArgs a = nf_pack(flow, theta, nullptr, log_q, max_rows, 0, 0);
    a.in_stride = stride;
    a.idx = idx;
    a.n_dev = n_dev;
    // one 32-row tile per workgroup: the moved chains of an iteration are a few per cent of the chains, so the launch is a
    // latency chain of couplings on a few dozen CUs; workgroups past the device-side count return at once
    return nf_launch_mode<true, true>(a, 32, 0, (hipStream_t)stream);
}

}  // extern "C"
