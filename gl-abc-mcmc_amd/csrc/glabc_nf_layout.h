// glabc_nf_layout.h -- layout of one coupling's parameter block (include/glabc.h: GLABC_NF_COUPLING_FLOATS), shared by the
// sampling / density kernels (glabc_nf.hip) and the training kernels (glabc_nf_train.hip).
#pragma once

#include "../../include/glabc.h"

namespace glabc {

constexpr int NF_H = 128;                                     // hidden width, GLMCMC_NFs.py:56
constexpr int NF_W2_OFF = 0;                                  // layout of one coupling's parameter block (floats)
constexpr int NF_W1_OFF = NF_H * NF_H;                        //   W2^T [k][i] | W1 | b1 | (b2, W3[0], W3[1], 0)[i] | b3[2] pad[2]
constexpr int NF_B1_OFF = NF_W1_OFF + NF_H;
constexpr int NF_V4_OFF = NF_B1_OFF + NF_H;
constexpr int NF_B3_OFF = NF_V4_OFF + 4 * NF_H;
constexpr int NF_BLOCK_FLOATS = NF_B3_OFF + 4;                // = GLABC_NF_COUPLING_FLOATS
static_assert(NF_BLOCK_FLOATS == GLABC_NF_COUPLING_FLOATS, "parameter block layout");
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace glabc
