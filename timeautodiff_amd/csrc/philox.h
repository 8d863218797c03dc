// Philox4x32-10 (Salmon et al. 2011) and the normal draws built on it, shared by the kernels that draw on the device:
// a value is a function of (seed, counter, element) alone, so whoever owns an element draws it -- a rank, a chunk of the
// batch, or the kernel that consumes it.
#pragma once
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace tad_rng {

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = u4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += W0; k1 += W1;
  }
  return c;
}

// the streams of one draw: counter word 2 of the Philox input
enum { ST_PERM = 0, ST_T = 1, ST_I = 2, ST_NOISE = 3, ST_SAMPLER = 4 };

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }            // [0, 1)
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }   // (0, 1]

// Noise of reverse step `counter` of the sampler: values 2 p and 2 p + 1 of global row g of the [B*T, K] latent tensor
// (P = ceil(K / 2) pairs per row) come from draw g P + p of the step's counter.
__device__ __forceinline__ void sampler_noise_pair(uint32_t k0, uint32_t k1, uint32_t counter, int64_t g, int P, int p, float& even, float& odd) {
  const uint64_t ge = (uint64_t)(g * P + p);
  const u4 d = philox4x32_10(u4{(uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)ST_SAMPLER, counter}, k0, k1);
  const float rad = sqrtf(-2.0f * logf(u01_open(d.x)));
  float s, c;
  sincosf(6.283185307179586f * u01(d.y), &s, &c);
  even = rad * c;
  odd = rad * s;
}

}  // namespace tad_rng
