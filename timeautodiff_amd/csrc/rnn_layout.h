// Sizes of the images of a recurrent weight matrix W_hh[G*H, H] inside one packed forward buffer, shared by the
// packing kernels and the sequence kernels that read them.  All counts are 1-KiB blocks (64 lanes x 16 bytes).
#pragma once
#include "common.h"

// The cluster form of the forward sequence kernel exists for bf16 operands at H = 512: eight workgroups of 64 hidden
// units each keep one direction's whole W_hh in registers.
inline bool rnn_cluster_shape(int prec, int64_t H) { return prec == TAD_BF16 && H == 512; }

// blocks of the 32-row-form image, then (bf16 precision) of the 16-row-form image
inline int64_t rnn_fwd_image_blocks(int prec, int64_t H, int G) {
  const int64_t NJ = cdiv64(H, 32), nkf = cdiv64(H, prec == TAD_BF16 ? 16 : 8), nk32 = prec == TAD_BF16 ? cdiv64(H, 32) : 0;
  return NJ * nkf * G + NJ * nk32 * G * 2;
}
// blocks of the cluster-form image that follows them: four gate slots per unit (a GRU's fourth is zero)
inline int64_t rnn_cluster_image_blocks(int prec, int64_t H) { return rnn_cluster_shape(prec, H) ? 4 * H * H * 2 / 1024 : 0; }
