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

// Device words that a launch of this library needs to find zero (arrival counters, sums that atomics add to) are zeroed by
// this kernel and not by hipMemsetAsync.  Under stream capture the memset becomes a memset node, and replays of such a node
// were seen to fill the words with other values than the one captured (ROCm 7.2, the graph replayed from a process that
// had run other fills and small copies since the capture): counters of the replayed launch then started at arbitrary
// values.  A kernel node carries its arguments with it.  bytes: a multiple of 4, from a 4-byte-aligned address.
static __global__ void tad_zero_words_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static inline hipError_t tad_zero_async(void* p, size_t bytes, hipStream_t s) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(tad_zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

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

// Sum over the 64 lanes of a wave (all of them active), the same value in every lane.  Four DPP steps inside the
// rows of 16 lanes (quad swaps, then the half-row and row mirrors: no LDS crossbar round trip as a lane shuffle
// takes), then the four row sums are read as scalars and added in a fixed order.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  v = dpp_add<0x140>(v);     // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
