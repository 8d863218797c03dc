// The random draws of a training iteration and the small fills / copies around a step, as device kernels (gfx950):
// with these the timed step launches nothing but this library's kernels.
//
// tad_draw_batch: counter-based (Philox4x32-10, Salmon et al. 2011) so that every rank of a data-parallel job, and a
// replayed run, draws the same global batch from (seed, counter) alone -- no generator state lives on the device.
#include <math.h>

#include "common.h"
#include "philox.h"

namespace {

using namespace tad_rng;

struct DrawArgs {
  uint32_t k0, k1, counter;     // key = seed, counter word 3 = the iteration
  int64_t N, B, T, K;
  int n_steps;
  int half_bits;                // the index permutation works on 2 * half_bits bits
  int64_t* idx; float* t; int32_t* i; float* noise;
};

__device__ __forceinline__ u4 draw(const DrawArgs& a, int stream, uint64_t e) {
  return philox4x32_10(u4{(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)stream, a.counter}, a.k0, a.k1);
}

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
  return h;
}

// A bijection of [0, N): four Feistel rounds on 2 * half_bits bits (keys drawn from the ST_PERM stream), walked until
// the value falls inside [0, N) again (the domain is less than 4 N wide).  Distinct arguments give distinct indices:
// perm(0 .. B-1) is a sample of B windows without replacement, which is what random.sample draws (DIFF.py:306).
__device__ __forceinline__ uint64_t perm_index(const DrawArgs& a, const u4& keys, uint64_t x) {
  const uint32_t mask = (uint32_t)((1ull << a.half_bits) - 1);
  const uint32_t k[4] = {keys.x, keys.y, keys.z, keys.w};
  do {
    uint32_t L = (uint32_t)(x >> a.half_bits), R = (uint32_t)x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t F = mix32(R * 0x9E3779B1u + k[r]) & mask;
      const uint32_t nl = R;
      R = L ^ F; L = nl;
    }
    x = ((uint64_t)L << a.half_bits) | R;
  } while (x >= (uint64_t)a.N);
  return x;
}

// One workgroup of 64 lanes per sample b: its window index, its diffusion step, its T sorted time points (rank sort in
// LDS: position of a value = number of smaller ones, ties by index) and its T*K normals (Box-Muller, two per draw).
__global__ __launch_bounds__(64) void draw_batch_kernel(DrawArgs a) {
  extern __shared__ float tv[];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  if (lane == 0) {
    if (a.idx) a.idx[b] = (int64_t)perm_index(a, draw(a, ST_PERM, 0), (uint64_t)b);
    if (a.i) a.i[b] = (int32_t)(draw(a, ST_I, (uint64_t)b).x % (uint32_t)a.n_steps);
  }
  if (a.t) {
    const int T = (int)a.T;
    for (int j = lane; j < T; j += 64) tv[j] = u01(draw(a, ST_T, (uint64_t)(b * a.T + j)).x);
    __syncthreads();
    for (int j = lane; j < T; j += 64) {
      const float v = tv[j];
      int rank = 0;
      for (int k = 0; k < T; ++k) {
        const float w = tv[k];
        rank += (w < v || (w == v && k < j)) ? 1 : 0;
      }
      a.t[b * a.T + rank] = v;
    }
  }
  if (a.noise) {
    const int64_t n = a.T * a.K, base = b * n, pairs = (n + 1) >> 1;
    for (int64_t p = lane; p < pairs; p += 64) {
      const u4 d = draw(a, ST_NOISE, (uint64_t)(b * pairs + p));
      const float r = sqrtf(-2.0f * logf(u01_open(d.x)));
      float s, c;
      sincosf(6.283185307179586f * u01(d.y), &s, &c);
      a.noise[base + 2 * p] = r * c;
      if (2 * p + 1 < n) a.noise[base + 2 * p + 1] = r * s;
    }
  }
}

// Noise of one reverse step of the sampler for rows row0 .. row0 + rows of the GLOBAL [B*T, K] latent tensor: value k of
// global row g comes from draw (g * ceil(K/2) + k/2) of the step's counter -- a function of (seed, step, g, k) alone, so
// a rank or a chunk that owns any slice of the batch draws exactly its rows, one step at a time.
__global__ void sampler_noise_kernel(uint32_t k0, uint32_t k1, uint32_t counter, int64_t row0, int64_t rows, int K, float* __restrict__ out) {
  const int P = (K + 1) >> 1;
  const int64_t n = rows * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / P;
    const int p = (int)(e - r * P);
    float even, odd;
    sampler_noise_pair(k0, k1, counter, row0 + r, P, p, even, odd);
    out[r * K + 2 * p] = even;
    if (2 * p + 1 < K) out[r * K + 2 * p + 1] = odd;
  }
}

// ---------------------------------------------------------------------------------------------- fills and copies
// rows x row_words 32-bit words, row stride ld words (buffers of any 4-byte or 2 x 2-byte element type)
__global__ void zero_2d_kernel(int64_t rows, int64_t row_words, int64_t ld, uint32_t* __restrict__ p) {
  const int64_t n = rows * row_words;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / row_words, c = e - r * row_words;
    p[r * ld + c] = 0u;
  }
}
__global__ void zero_flat_kernel(int64_t n16, int64_t n_words, uint32_t* __restrict__ p) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(p)[e] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0) {
    const int64_t e = (n16 << 2) + threadIdx.x;
    if (e < n_words) p[e] = 0u;
  }
}
__global__ void copy_2d_kernel(int64_t rows, int64_t cols, const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd) {
  const int64_t n = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e - r * cols;
    dst[r * ldd + c] = src[r * lds_ + c];
  }
}
__global__ void add_vec_kernel(int64_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) out[e] = a[e] + b[e];
}

__global__ void fill_f32_kernel(int64_t n, float* __restrict__ p, float v) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

__global__ void sigmoid_f32_kernel(int64_t n, float* __restrict__ p) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = sigmoid_acc(p[e]);
}

inline unsigned grid_for(int64_t n) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g < 1 ? 1 : g > 2048 ? 2048 : g);
}

}  // namespace

extern "C" int tad_draw_batch(uint64_t seed, uint64_t counter, int64_t N, int64_t B, int64_t T, int64_t K, int n_steps,
                              int64_t* idx, float* t, int32_t* i_idx, float* noise, tad_stream_t stream) {
  TAD_CHECK(B >= 0 && T >= 0 && K >= 0 && counter < (1ull << 32), "tad_draw_batch: bad arguments B=%ld T=%ld K=%ld counter=%llu", (long)B, (long)T, (long)K,
            (unsigned long long)counter);
  TAD_CHECK(!idx || (N >= B && N >= 1 && N < (1ll << 40)), "tad_draw_batch: a sample of %ld windows needs at least as many (N = %ld)", (long)B, (long)N);
  TAD_CHECK(!i_idx || n_steps >= 1, "tad_draw_batch: n_steps must be positive");
  TAD_CHECK(T <= 8192, "tad_draw_batch: T = %ld exceeds 8192 (the time points of a sample are sorted in LDS)", (long)T);
  if (B == 0) return TAD_OK;
  DrawArgs a{};
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.counter = (uint32_t)counter;
  a.N = N; a.B = B; a.T = T; a.K = K; a.n_steps = n_steps;
  int bits = 2;
  while (bits < 62 && (1ll << bits) < N) ++bits;
  a.half_bits = (bits + 1) / 2;
  a.idx = idx; a.t = T > 0 ? t : nullptr; a.i = i_idx; a.noise = (T > 0 && K > 0) ? noise : nullptr;
  hipLaunchKernelGGL(draw_batch_kernel, dim3((unsigned)B), dim3(64), (size_t)(T > 0 ? T : 1) * 4, (hipStream_t)stream, a);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_sampler_noise(uint64_t seed, uint64_t counter, int64_t row0, int64_t rows, int K, float* out, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && row0 >= 0 && K >= 1 && counter < (1ull << 32), "tad_sampler_noise: bad arguments rows=%ld row0=%ld K=%d", (long)rows, (long)row0, K);
  if (rows == 0) return TAD_OK;
  TAD_CHECK(out, "tad_sampler_noise: null pointer");
  hipLaunchKernelGGL(sampler_noise_kernel, dim3(grid_for(rows * ((K + 1) / 2))), dim3(256), 0, (hipStream_t)stream, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (uint32_t)counter, row0, rows, K, out);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_zero_2d(int64_t rows, int64_t row_bytes, void* p, int64_t ld_bytes, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && row_bytes >= 0 && ld_bytes >= row_bytes, "tad_zero_2d: bad shape rows=%ld row_bytes=%ld ld_bytes=%ld", (long)rows, (long)row_bytes, (long)ld_bytes);
  if (rows == 0 || row_bytes == 0) return TAD_OK;
  TAD_CHECK(p && (reinterpret_cast<uintptr_t>(p) & 3) == 0 && row_bytes % 4 == 0 && ld_bytes % 4 == 0, "tad_zero_2d: 4-byte granularity");
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  if (rows == 1 || ld_bytes == row_bytes) {
    const int64_t words = rows * (row_bytes / 4);
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      hipLaunchKernelGGL(zero_flat_kernel, dim3(grid_for(words / 4 + 1)), dim3(256), 0, (hipStream_t)stream, words / 4, words, q);
      TAD_LAUNCH_CHECK();
      return TAD_OK;
    }
    hipLaunchKernelGGL(zero_2d_kernel, dim3(grid_for(words)), dim3(256), 0, (hipStream_t)stream, (int64_t)1, words, words, q);
    TAD_LAUNCH_CHECK();
    return TAD_OK;
  }
  hipLaunchKernelGGL(zero_2d_kernel, dim3(grid_for(rows * (row_bytes / 4))), dim3(256), 0, (hipStream_t)stream, rows, row_bytes / 4, ld_bytes / 4, q);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_copy_2d(int64_t rows, int64_t cols, const float* src, int64_t ld_src, float* dst, int64_t ld_dst, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, "tad_copy_2d: bad shape");
  if (rows == 0 || cols == 0) return TAD_OK;
  TAD_CHECK(src && dst, "tad_copy_2d: null pointer");
  hipLaunchKernelGGL(copy_2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, rows, cols, src, ld_src, dst, ld_dst);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_add_vec(int64_t n, const float* a, const float* b, float* out, tad_stream_t stream) {
  TAD_CHECK(n >= 0, "tad_add_vec: bad n");
  if (n == 0) return TAD_OK;
  TAD_CHECK(a && b && out, "tad_add_vec: null pointer");
  hipLaunchKernelGGL(add_vec_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, a, b, out);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_fill_f32(int64_t n, float* p, float value, tad_stream_t stream) {
  TAD_CHECK(n >= 0, "tad_fill_f32: bad n");
  if (n == 0) return TAD_OK;
  TAD_CHECK(p, "tad_fill_f32: null pointer");
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, p, value);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_sigmoid_f32(int64_t n, float* p, tad_stream_t stream) {
  TAD_CHECK(n >= 0, "tad_sigmoid_f32: bad n");
  if (n == 0) return TAD_OK;
  TAD_CHECK(p, "tad_sigmoid_f32: null pointer");
  hipLaunchKernelGGL(sigmoid_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, p);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
