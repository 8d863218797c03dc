// Shared device helpers for the gfx950 (CDNA4) kernels of the TimeAutoDiff hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tad_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define TAD_WAVE 64

// error plumbing (host)
void tad_set_error(const char* fmt, ...);
void tad_note_kernel(const char* fmt, ...);   // records the launched kernel's name for tad_last_kernel()
#define TAD_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      tad_set_error(__VA_ARGS__);       \
      return TAD_ERR_INVALID;           \
    }                                   \
  } while (0)
#define TAD_LAUNCH_CHECK()                                                      \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      tad_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return TAD_ERR_LAUNCH;                                                    \
    }                                                                           \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// device math
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// Gate nonlinearities on the hardware exp2 / reciprocal units (v_exp_f32, v_rcp_f32):
// absolute error ~2e-7, saturating correctly at both ends.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Row of a 32x32 MFMA accumulator register: C/D layout of v_mfma_f32_32x32x*:
// column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int acc_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
