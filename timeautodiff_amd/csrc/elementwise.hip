// Fused elementwise / reduction kernels of the TimeAutoDiff hot path (gfx950).
// All of these are HBM-bound: one pass over their operands, coalesced along the
// fastest dimension, wave-level reductions (64 lanes) followed by one atomic per wave
// or workgroup for the scalar / per-column results.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "philox.h"

namespace {

constexpr int EW_BLOCK = 256;
constexpr int MAX_DISC = 64;

inline unsigned ew_grid(int64_t n, int per_thread = 1) {
  int64_t g = cdiv64(n, (int64_t)EW_BLOCK * per_thread);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// every row of a [rows, ld] tensor of `esz`-byte elements starts at a multiple of `align` bytes
inline bool wide_rows_ok(const void* p, int64_t ld, int esz, int align) {
  return (reinterpret_cast<uintptr_t>(p) % (uintptr_t)align) == 0 && (ld * esz) % align == 0;
}

// block-wide sum of one value per thread -> atomicAdd(out) once per block
__device__ __forceinline__ void block_atomic_sum(float v, float* out) {
  __shared__ float part[EW_BLOCK / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
    atomicAdd(out, s);
  }
  __syncthreads();
}

// Sum of squares of a flat buffer: 16-byte loads, per-lane partial sums over a grid stride, one wave reduction and one
// atomic per workgroup (the buffer is the flat gradient: a few million floats, one pass at HBM rate).
__global__ void sumsq_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ out) {
  const int64_t n4 = n >> 2;
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[e];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
  block_atomic_sum(acc, out);
}

// ---------------------------------------------------------------------------
__global__ void add_noise_kernel(int64_t B, int64_t T, int64_t K, const float* __restrict__ x0,
                                 const float* __restrict__ noise, const int32_t* __restrict__ i_idx,
                                 const float* __restrict__ alphas, float* __restrict__ xn, float* __restrict__ i_f) {
  const int64_t row = T * K, n = B * row;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / row;
    const int i = i_idx[b];
    const float a = alphas[i];
    xn[e] = sqrtf(a) * x0[e] + sqrtf(1.f - a) * noise[e];
    if (i_f) {
      const int64_t w = e - b * row;
      if (w < T) i_f[b * T + w] = (float)i;
    }
  }
}

__global__ void sampler_update_kernel(int64_t n, float* __restrict__ x, const float* __restrict__ pred,
                                      const float* __restrict__ z, float c1, float c2, float c3, int last) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float p = pred[e];
    // same operation order as the reference expression: c1 * (x - beta*pred/sqrt(1-alpha)) + sqrt(beta)*z
    float v = __fadd_rn(__fmul_rn(c1, __fsub_rn(x[e], __fmul_rn(c2, p))), __fmul_rn(c3, z[e]));
    if (last) v = __fmul_rn(c1, __fsub_rn(v, __fmul_rn(c2, p)));
    x[e] = v;
  }
}

__global__ void mse_loss_kernel(int64_t n, const float* __restrict__ pred, const float* __restrict__ target,
                                float gscale, float* __restrict__ dpred, float* __restrict__ loss_sum) {
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[e] - target[e];
    acc += d * d;
    if (dpred) dpred[e] = gscale * d;
  }
  if (loss_sum) block_atomic_sum(acc, loss_sum);
}

// out[r, j*8+k] = sin(2^k pi x[r,j]);  out[r, 8n + j*8+k] = cos(...)
// OT: element type of the result (float, or bf16_t where only matrix products read it)
template <typename OT>
__global__ void sincos_kernel(int64_t rows, int nn, const float* __restrict__ x, int64_t ldx, OT* __restrict__ out,
                              int64_t ldo) {
  const int64_t n = rows * nn * 8;
  const float pi = 3.14159265358979323846f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / (nn * 8);
    const int c = (int)(e - r * (nn * 8));
    const int j = c >> 3, k = c & 7;
    const float freq = __fmul_rn((float)(1 << k), pi);    // exact power-of-two scaling of float(pi)
    const float ang = __fmul_rn(freq, x[r * ldx + j]);
    float s, co;
    sincosf(ang, &s, &co);
    out[r * ldo + c] = (OT)s;
    out[r * ldo + nn * 8 + c] = (OT)co;
  }
}

// 8 consecutive results as one 16-byte (bf16) or two 16-byte (fp32) stores; dst is 16-byte aligned
template <typename OT>
__device__ __forceinline__ void store8(OT* dst, const float (&v)[8]) {
  if constexpr (sizeof(OT) == 2) {
    OT t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (OT)v[k];
    *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, t);
  } else {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// The same values, one thread per (row, numeric column): x is read once, the 8 sines and the 8 cosines of the column
// leave as 16-byte stores (rows of `out` 16-byte aligned), and no thread loops -- the grid covers the elements, so
// the load -> sincos -> store chains of a wave's threads are all in flight together.
template <typename OT>
__global__ void sincos8_kernel(int64_t rows, int nn, const float* __restrict__ x, int64_t ldx, OT* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * nn) return;
  const int64_t r = e / nn;
  const int j = (int)(e - r * nn);
  const float pi = 3.14159265358979323846f;
  const float xv = x[r * ldx + j];
  float sn[8], cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float freq = __fmul_rn((float)(1 << k), pi);
    sincosf(__fmul_rn(freq, xv), &sn[k], &cs[k]);
  }
  store8(out + r * ldo + 8 * j, sn);
  store8(out + r * ldo + 8 * (nn + j), cs);
}

struct DiscMeta {
  int32_t off[MAX_DISC];
  int32_t card[MAX_DISC];
};

template <typename OT>
__global__ void embed_fwd_kernel(int64_t rows, int nd, int E, const float* __restrict__ x, int64_t ldx,
                                 const float* __restrict__ tables, DiscMeta m, OT* __restrict__ out, int64_t ldo) {
  const int64_t n = rows * nd * E;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / (nd * E);
    const int c = (int)(e - r * (nd * E));
    const int d = c / E, q = c - d * E;
    int code = (int)x[r * ldx + d];
    code = min(max(code, 0), m.card[d] - 1);
    out[r * ldo + c] = (OT)tables[((int64_t)m.off[d] + code) * E + q];
  }
}

// E a multiple of 8, rows of `out` 16-byte aligned: one thread per 8 consecutive entries of a table row
template <typename OT>
__global__ void embed_fwd8_kernel(int64_t rows, int nd, int E, const float* __restrict__ x, int64_t ldx,
                                  const float* __restrict__ tables, DiscMeta m, OT* __restrict__ out, int64_t ldo) {
  const int e8 = E / 8;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * nd * e8) return;
  const int64_t r = e / (nd * e8);
  const int c = (int)(e - r * (nd * e8));
  const int d = c / e8, q = 8 * (c - d * e8);
  int code = (int)x[r * ldx + d];
  code = min(max(code, 0), m.card[d] - 1);
  const float* src = tables + ((int64_t)m.off[d] + code) * E + q;
  const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  store8(out + r * ldo + (int64_t)d * E + q, v);
}

// One workgroup = one discrete column and one slab of rows; the column's table gradient is
// accumulated in LDS (ds_add_f32) and flushed with one global atomic per touched entry.
__global__ void embed_bwd_lds_kernel(int64_t rows, int nd, int E, const float* __restrict__ x, int64_t ldx,
                                     const float* __restrict__ dout, int64_t ldo, float* __restrict__ dtables,
                                     DiscMeta m, int64_t rows_per_block) {
  extern __shared__ float tab[];
  const int d = blockIdx.y;
  const int card = m.card[d];
  for (int i = threadIdx.x; i < card * E; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  const int64_t n = (r1 - r0) * E;
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
    const int64_t r = r0 + e / E;
    const int q = (int)(e % E);
    int code = (int)x[r * ldx + d];
    code = min(max(code, 0), card - 1);
    atomicAdd(&tab[code * E + q], dout[r * ldo + (int64_t)d * E + q]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < card * E; i += blockDim.x) {
    const float v = tab[i];
    if (v != 0.f) atomicAdd(&dtables[(int64_t)m.off[d] * E + i], v);
  }
}

// Small tables (card <= 16, E <= 128): the sums stay in registers.  A wave owns rows (stride = waves of the launch), a
// lane two columns of E; the row's code is wave-uniform, so the add goes to the registers of that code through a uniform
// branch -- no LDS atomics in the loop (a float add in LDS costs about four cycles per lane: 245 cycles per wave
// instruction, which is what bounded embed_bwd_lds_kernel at 90 us for 98 304 rows x 128).  The waves of a workgroup
// combine in LDS, one global atomic per table entry and workgroup.
constexpr int EB_CARD = 16;
__global__ __launch_bounds__(256) void embed_bwd_reg_kernel(int64_t rows, int nd, int E, const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ dout, int64_t ldo, float* __restrict__ dtables, DiscMeta m) {
  __shared__ float part[4][EB_CARD * 128];
  const int d = blockIdx.y;
  const int card = m.card[d];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  float acc[EB_CARD][2];
#pragma unroll
  for (int c = 0; c < EB_CARD; ++c) acc[c][0] = acc[c][1] = 0.f;
  const bool q0 = lane < E, q1 = lane + 64 < E;
  constexpr int U = 4;                                   // rows in flight per wave
  for (int64_t r0 = gw; r0 < rows; r0 += (int64_t)U * nw) {
    int code[U];
    float v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + (int64_t)u * nw;
      code[u] = -1; v0[u] = v1[u] = 0.f;
      if (r < rows) {
        int cd = (int)x[r * ldx + d];
        code[u] = __builtin_amdgcn_readfirstlane(min(max(cd, 0), card - 1));
        const float* dr = dout + r * ldo + (int64_t)d * E;
        v0[u] = q0 ? dr[lane] : 0.f;
        v1[u] = q1 ? dr[lane + 64] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < EB_CARD; ++c)
        if (code[u] == c) { acc[c][0] += v0[u]; acc[c][1] += v1[u]; }
    }
  }
#pragma unroll
  for (int c = 0; c < EB_CARD; ++c) { part[wave][c * 128 + lane] = acc[c][0]; part[wave][c * 128 + 64 + lane] = acc[c][1]; }
  __syncthreads();
  for (int i = threadIdx.x; i < card * 128; i += blockDim.x) {
    const int c = i >> 7, q = i & 127;
    const float v = part[0][i] + part[1][i] + part[2][i] + part[3][i];
    if (q < E && v != 0.f) atomicAdd(&dtables[((int64_t)m.off[d] + c) * E + q], v);
  }
}

__global__ void embed_bwd_direct_kernel(int64_t rows, int nd, int E, const float* __restrict__ x, int64_t ldx,
                                        const float* __restrict__ dout, int64_t ldo, float* __restrict__ dtables,
                                        DiscMeta m) {
  const int64_t n = rows * nd * E;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / (nd * E);
    const int c = (int)(e - r * (nd * E));
    const int d = c / E, q = c - d * E;
    int code = (int)x[r * ldx + d];
    code = min(max(code, 0), m.card[d] - 1);
    atomicAdd(&dtables[((int64_t)m.off[d] + code) * E + q], dout[r * ldo + c]);
  }
}

// IT: index type of the flat element counter (uint32_t when rows * H fits: a 64-bit division per element costs more
// than the rest of the kernel)
template <typename IT, typename OT>
__global__ void posenc_kernel(int64_t rows, int H, const float* __restrict__ t, int64_t ts, float inv_max,
                              const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ scale,
                              const float* __restrict__ shift, OT* __restrict__ out, int64_t ldo) {
  const int L = H / 2;
  const IT n = (IT)(rows * H);
  for (IT e = (IT)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (IT)gridDim.x * blockDim.x) {
    const IT rr = e / (IT)H;
    const int64_t r = (int64_t)rr;
    const int c = (int)(e - rr * (IT)H);
    const float tv = t[r * ts];
    float v;
    if (c < L) v = __fadd_rn(__fmul_rn(__fmul_rn(tv, inv_max), w[c]), b[c]);
    else v = sinf(__fadd_rn(__fmul_rn(tv, scale[c - L]), shift[c - L]));
    out[r * ldo + c] = (OT)v;
  }
}

// H a multiple of 8, rows of `out` 8-byte (bf16) or 16-byte (fp32) aligned: one thread per 4 columns, same arithmetic
// per element, no loop.  blockIdx.y picks the half of the row (0: the linear columns, 1: the sines), so a wave never
// holds both kinds: with them mixed every wave pays for the sine of all its lanes.
template <typename OT>
__global__ void posenc4_kernel(int64_t rows, int H, const float* __restrict__ t, int64_t ts, float inv_max,
                               const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ scale,
                               const float* __restrict__ shift, OT* __restrict__ out, int64_t ldo) {
  const int L = H / 2, l4 = L / 4;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * l4) return;
  const int64_t r = e / l4;
  const int c = 4 * (int)(e - r * l4);
  const float tv = t[r * ts];
  float v[4];
  // the four per-column tables are 16-byte aligned (checked at the launch): one load per table and thread
  if (blockIdx.y == 0) {
    const float tn = __fmul_rn(tv, inv_max);
    const float4 wv = *reinterpret_cast<const float4*>(w + c), bv = *reinterpret_cast<const float4*>(b + c);
    const float ws[4] = {wv.x, wv.y, wv.z, wv.w}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __fadd_rn(__fmul_rn(tn, ws[k]), bs[k]);
  } else {
    const float4 sv = *reinterpret_cast<const float4*>(scale + c), hv = *reinterpret_cast<const float4*>(shift + c);
    const float ss[4] = {sv.x, sv.y, sv.z, sv.w}, hs[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = sinf(__fadd_rn(__fmul_rn(tv, ss[k]), hs[k]));
  }
  OT* dst = out + r * ldo + c + (blockIdx.y ? L : 0);
  if constexpr (sizeof(OT) == 2) {
    OT q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (OT)v[k];
    *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, q);
  } else {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <typename OT>
__global__ void silu_fwd_kernel(int64_t n, const float* __restrict__ z, OT* __restrict__ a) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = z[e];
    a[e] = (OT)(v * sigmoid_acc(v));
  }
}

template <typename OT>
__global__ void silu_bwd_kernel(int64_t n, const float* __restrict__ z, const float* __restrict__ da, OT* __restrict__ dz) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = z[e], s = sigmoid_acc(v);
    dz[e] = (OT)(da[e] * s * (1.f + v * (1.f - s)));
  }
}

// One wave per row; the row lives in registers (NC = ceil(D / 64) elements per lane), so x is read once.
// Sums run over the lane's elements in column order, then across the wave: two-pass mean / variance.
template <int NC>
__global__ void layernorm_fwd_kernel(int64_t rows, int D, const float* __restrict__ x, int64_t ldx,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     float* __restrict__ y, int64_t ldy, float* __restrict__ mr) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: row addresses = scalar base + lane offset
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float gm[NC], bt[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    gm[i] = c < D ? gamma[c] : 0.f;
    bt[i] = c < D ? beta[c] : 0.f;
  }
  // the next row of the wave is fetched before this one is worked on: a row's chain (load, two wave reductions, K more,
  // stores) is long and the wave has nothing else to overlap it with
  float xn[NC];
  auto fetch = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xn[i] = (r < rows && c < D) ? x[r * ldx + c] : 0.f;
    }
  };
  fetch(wave);
  for (int64_t r = wave; r < rows; r += nwaves) {
    float xv[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      xv[i] = xn[i];
      s += xv[i];
    }
    fetch(r + nwaves);
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float dlt = xv[i] - mean;
      if (lane + 64 * i < D) v += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) y[r * ldy + c] = (xv[i] - mean) * rstd * gm[i] + bt[i];
    }
    if (mr && lane == 0) { mr[2 * r] = mean; mr[2 * r + 1] = rstd; }
  }
}

// Where the noise of a reverse step comes from when the tail draws it itself (tad_sampler_noise's values, bit for bit).
struct TailNoise {
  uint32_t k0, k1, counter;
  int on;
  int64_t row0;          // global row of the launch's first row
};

// Tail of one reverse-diffusion step, one wave per row (= one (sequence, time) pair): LayerNorm of the recurrent
// output row (kept in registers, as layernorm_fwd_kernel), the K-wide output projection, the sampler update of
// the row's K latent values, and the input projection of the UPDATED values for the next step.  round16: the
// operands of the two projections are rounded to bf16 where a bf16 matrix product would round them.
// A row costs its instructions, not its bytes (the first form ran 2400 per row, most of them lane masks and scalar
// guards of the unrolled loops): every load of a row is issued at the top of its iteration through clamped addresses
// (no branch around a load), the two weight matrices sit in LDS padded to 8 values per column, read 16 bytes at a time,
// loops over the K outputs have no guards where K is the template's KT (KT = 0: any K <= 8, guarded).  What a lane keeps
// per row is what it must (the row, double-buffered, and NP = ceil(H / 64) values of the next projection's step-invariant
// share): LayerNorm's weights are read from LDS too, so that four waves per SIMD stay resident at D = 1024.
template <int NC, int KT, int NP>
__global__ __launch_bounds__(EW_BLOCK) void sampler_tail_kernel(int64_t rows, int D, int K, int H, int round16, const float* __restrict__ h, int64_t ldh,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                    const float* __restrict__ fcw, const float* __restrict__ fcb, float* __restrict__ x,
                                    const float* __restrict__ z, TailNoise nz, float c1, float c2, float c3, int last,
                                    const float* __restrict__ w_in, const float* __restrict__ b_in,
                                    float* __restrict__ xh, int64_t ldxh, bf16_t* __restrict__ xh16,
                                    const float* __restrict__ vrow, const float* __restrict__ pinv, int64_t ldp) {
  constexpr int KMAX = 8;
  constexpr int KN = KT ? KT : KMAX;           // trip count of the loops over the outputs
  extern __shared__ __attribute__((aligned(16))) float tail_lds[];
  // LDS: w_in [H][8] (rounded, zeros beyond K), b_in [H], vrow [H], fc's weight [64 NC][8] (rounded, zeros beyond K and D),
  // LayerNorm's gamma and beta [64 NC] each (zeros beyond D).
  // pinv != NULL: w_in / b_in are the input projection folded into the x share of `proj` (prepared by the caller from
  // rounded factors: not rounded again) and the row leaves as relu(. + vrow + pinv[r]), the operand of the LSTM's input side
  float* const win = tail_lds;
  float* const bin = tail_lds + (size_t)H * KMAX;
  float* const vr = bin + H;
  float* const fw = vr + H + ((4 - ((2 * H) & 3)) & 3);      // 16-byte aligned
  float* const gml = fw + 64 * NC * KMAX;
  float* const btl = gml + 64 * NC;
  // (the fills keep eight loads per thread in flight through clamped addresses: a branch around each load made the
  // prologue of a workgroup a chain of sixteen round trips to L2, half of the launch's time)
  auto fill = [&](float* dst, int n, auto value) {
    for (int e0 = threadIdx.x; e0 < n; e0 += 8 * EW_BLOCK) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * EW_BLOCK;
        v[j] = value(e < n ? e : n - 1);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * EW_BLOCK;
        if (e < n) dst[e] = v[j];
      }
    }
  };
  const bool round_in = round16 && !pinv;
  fill(win, H * KMAX, [&](int e) {
    const int u = e >> 3, k = e & 7;
    const float w = w_in[u * K + (k < K ? k : K - 1)];
    const float wr = round_in ? (float)(bf16_t)w : w;
    return k < K ? wr : 0.f;
  });
  fill(bin, H, [&](int e) { return b_in[e]; });
  if (pinv) fill(vr, H, [&](int e) { return vrow[e]; });
  fill(fw, 64 * NC * KMAX, [&](int e) {
    const int c = e >> 3, k = e & 7;
    const float w = fcw[(int64_t)(k < K ? k : K - 1) * D + (c < D ? c : D - 1)];
    const float wr = round16 ? (float)(bf16_t)w : w;
    return (c < D && k < K) ? wr : 0.f;
  });
  fill(gml, 64 * NC, [&](int c) { const float v = gamma[c < D ? c : D - 1]; return c < D ? v : 0.f; });
  fill(btl, 64 * NC, [&](int c) { const float v = beta[c < D ? c : D - 1]; return c < D ? v : 0.f; });
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // (the wave's index through readfirstlane: row addresses are then scalar bases with 32-bit lane offsets, not 64-bit
  // per-lane pointers -- two registers and an add per load less)
  const int64_t wave = (int64_t)blockIdx.x * (EW_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t nwaves = (int64_t)gridDim.x * (EW_BLOCK / 64);
  const int nD = (D + 63) >> 6, nH = (H + 63) >> 6;        // 64-column blocks of a row that exist (nH <= NP)
  const int kl = lane < K ? lane : K - 1;
  const float fcb_l = fcb[kl];
  // The loads of row r + nwaves are issued before row r's stores: the wait for loaded values releases in issue order, so
  // loads issued behind the stores would wait for the stores' acknowledgements too (a row then took its store latency, its
  // load latency and its arithmetic one after the other: 5 us).
  float xq[NC], pq[NP], xo_q, z_q;
  auto fetch = [&](int64_t rr) {
    if (rr >= rows) rr = rows - 1;               // (uniform; the values are not used)
    const float* hr = h + rr * ldh;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xq[i] = i < nD ? hr[c < D ? c : D - 1] : 0.f;
    }
    xo_q = x[rr * K + kl];
    z_q = nz.on ? 0.f : z[rr * K + kl];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int u = lane + 64 * i;
      pq[i] = (pinv && i < nH) ? pinv[rr * ldp + (u < H ? u : H - 1)] : 0.f;
    }
  };
  if (wave < rows) fetch(wave);
  // In-launch noise: a draw costs some 350 instructions and a row has K values, so the wave draws for its next 64 / K rows
  // at once (lane = (row of the batch, k)) and each row picks its K values up from those lanes.
  const int rpb = 64 / K;                        // rows per batch of draws
  const int bl_row = lane / K, bl_k = lane - bl_row * K;
  float zbatch = 0.f;
  int jj = 0;                                    // the row's place in its batch
  for (int64_t r = wave; r < rows; r += nwaves) {
    float xv[NC], pv[NP];
#pragma unroll
    for (int i = 0; i < NC; ++i) xv[i] = xq[i];
#pragma unroll
    for (int i = 0; i < NP; ++i) pv[i] = pq[i];
    const float x_old = xo_q;
    float zv = z_q;
    fetch(r + nwaves);
    if (nz.on) {            // (arithmetic only: runs while the loads are in flight)
      if (jj == 0) {
        float even, odd;
        tad_rng::sampler_noise_pair(nz.k0, nz.k1, nz.counter, nz.row0 + r + (int64_t)bl_row * nwaves, (K + 1) >> 1, bl_k >> 1, even, odd);
        zbatch = (bl_k & 1) ? odd : even;
      }
      zv = __shfl(zbatch, jj * K + kl, 64);
      jj = jj + 1 == rpb ? 0 : jj + 1;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (lane + 64 * i >= D) xv[i] = 0.f;
      s += xv[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float dlt = xv[i] - mean;
      v += lane + 64 * i < D ? dlt * dlt : 0.f;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
    float acc[KN];
#pragma unroll
    for (int k = 0; k < KN; ++k) acc[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (i < nD) {
        float y = (xv[i] - mean) * rstd * gml[lane + 64 * i] + btl[lane + 64 * i];
        if (round16) y = (float)(bf16_t)y;
        if (lane + 64 * i >= D) y = 0.f;
        const float4 w0 = *reinterpret_cast<const float4*>(fw + (size_t)(lane + 64 * i) * KMAX);
        const float4 w1 = *reinterpret_cast<const float4*>(fw + (size_t)(lane + 64 * i) * KMAX + 4);
        const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int k = 0; k < KN; ++k) acc[k] += y * w8[k];
      }
      // (wide rows: the weights of four column blocks at a time, not of all sixteen, wait in registers)
      if constexpr (NC > 8) { if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
    }
    float pk = 0.f;                              // lane k < K: the row's prediction k (before the bias)
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      if (KT || k < K) {
        const float t = wave_sum(acc[k]);
        if (lane == k) pk = t;
      }
    }
    float mine;                                  // lane k < K: the updated x[r, k]
    {
      const float p = pk + fcb_l;
      // same operation order as sampler_update_kernel
      float nv = __fadd_rn(__fmul_rn(c1, __fsub_rn(x_old, __fmul_rn(c2, p))), __fmul_rn(c3, zv));
      if (last) nv = __fmul_rn(c1, __fsub_rn(nv, __fmul_rn(c2, p)));
      if (lane < K) x[r * K + lane] = nv;
      mine = nv;
    }
    float xn[KN];
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), k));
      xn[k] = round16 ? (float)(bf16_t)t : t;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i < nH) {
        const int u = lane + 64 * i, uc = u < H ? u : H - 1;
        const float4 w0 = *reinterpret_cast<const float4*>(win + (size_t)uc * KMAX);
        const float4 w1 = *reinterpret_cast<const float4*>(win + (size_t)uc * KMAX + 4);
        const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < KN; ++k)
          if (KT || k < K) a += xn[k] * w8[k];
        float o = a + bin[uc];
        if (pinv) o = fmaxf((o + vr[uc]) + pv[i], 0.f);
        if (u < H) {
          if (xh16) xh16[r * ldxh + u] = (bf16_t)o;            // the next step's `proj` reads it as a bf16 operand
          else xh[r * ldxh + u] = o;
        }
      }
    }
  }
}

// LayerNorm followed by a narrow output projection (K <= 8 columns), one wave per row: the normalised row stays in
// registers, the K outputs are K wave reductions.  Writes mean / rstd for the backward and, only if asked, the
// normalised row.  round16: the projection's operands are rounded to bf16 where a bf16 matrix product would.
template <int NC, int KMAX>
__global__ __launch_bounds__(EW_BLOCK) void ln_fc_fwd_kernel(int64_t rows, int D, int K, int round16, const float* __restrict__ x, int64_t ldx,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                 const float* __restrict__ fcw, const float* __restrict__ fcb, float* __restrict__ y, int64_t ldy,
                                 float* __restrict__ mr, float* __restrict__ pred) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: row addresses = scalar base + lane offset
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float gm[NC], bt[NC], wk[KMAX][NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    gm[i] = c < D ? gamma[c] : 0.f;
    bt[i] = c < D ? beta[c] : 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float w = (c < D && k < K) ? fcw[(int64_t)k * D + c] : 0.f;
      wk[k][i] = round16 ? (float)(bf16_t)w : w;
    }
  }
  float xn[NC];                       // the wave's next row, fetched before the current one is worked on
  auto fetch = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xn[i] = (r < rows && c < D) ? x[r * ldx + c] : 0.f;
    }
  };
  fetch(wave);
  for (int64_t r = wave; r < rows; r += nwaves) {
    float xv[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      xv[i] = xn[i];
      s += xv[i];
    }
    fetch(r + nwaves);
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float dlt = xv[i] - mean;
      if (lane + 64 * i < D) v += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      float yv = (xv[i] - mean) * rstd * gm[i] + bt[i];
      if (y && c < D) y[r * ldy + c] = yv;
      if (round16) yv = (float)(bf16_t)yv;
      if (c >= D) yv = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc[k] += yv * wk[k][i];
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const float p = wave_sum(acc[k]) + fcb[k];
        if (lane == k) pred[r * K + k] = p;
      }
    }
    if (lane == 0) { mr[2 * r] = mean; mr[2 * r + 1] = rstd; }
  }
}

// Backward of the pair above, the normalised row recomputed from x and mean / rstd, as two launches over the rows
// (one kernel, two roles; each keeps its column sums per lane over the wave's rows, combines them over the
// workgroup's waves in LDS and adds them with one atomic per entry and workgroup):
//   DW == false: d(row) = dpred W, LayerNorm's backward -> dx, dgamma, dbeta; many workgroups (it streams x in and
//                dx out);
//   DW == true:  the projection's weight gradient dW[k][c] += dpred[r][k] * y[r][c] and its bias gradient; few
//                workgroups (K * D atomics each).
// One launch doing both needs K * D / 64 more accumulators per lane than the register file gives at two waves per
// SIMD, and its K + 2 atomics per column and workgroup cost more than the second pass over x.
template <int NC, int KMAX, bool DW>
__global__ __launch_bounds__(EW_BLOCK) void ln_fc_bwd_kernel(int64_t rows, int D, int K, int round16, const float* __restrict__ dpred,
                                 const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ mr, const float* __restrict__ fcw,
                                 float* __restrict__ dx, int64_t lddx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                 float* __restrict__ dfcw, float* __restrict__ dfcb) {
  extern __shared__ float lnfc_red[];              // DW ? [KMAX][NC * 64] : [2][NC * 64]
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, nwb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: row addresses = scalar base + lane offset
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int NA = DW ? KMAX : 2;
  // (rows of more than 512 columns, first role: gamma, beta and the projection's weights in LDS behind the reduction
  // rows, as in the forward -- the role ran one wave per SIMD on 283 registers)
  constexpr bool WL = NC > 8 && !DW;
  constexpr int WW = NC * 64;
  float* const lw = lnfc_red + NA * WW;            // WL: [KMAX + 2][WW]
  float acc[NA][NC], gm[WL ? 1 : NC], bt[WL ? 1 : NC], wk[(DW || WL) ? 1 : KMAX][WL ? 1 : NC], dbk[KMAX];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a][i] = 0.f;
  }
  if constexpr (WL) {
    for (int e = threadIdx.x; e < (KMAX + 2) * WW; e += blockDim.x) {
      const int k = e / WW, c = e - k * WW;
      float v = 0.f;
      if (c < D) {
        if (k < K) { v = fcw[(int64_t)k * D + c]; if (round16) v = (float)(bf16_t)v; }
        else if (k == KMAX) v = gamma[c];
        else if (k == KMAX + 1) v = beta[c];
      }
      lw[e] = v;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      gm[i] = c < D ? gamma[c] : 0.f;
      bt[i] = c < D ? beta[c] : 0.f;
      if constexpr (!DW) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          const float w = (c < D && k < K) ? fcw[(int64_t)k * D + c] : 0.f;
          wk[k][i] = round16 ? (float)(bf16_t)w : w;
        }
      }
    }
  }
  auto GM = [&](int i) -> float { if constexpr (WL) return lw[KMAX * WW + lane + 64 * i]; else return gm[i]; };
  auto BT = [&](int i) -> float { if constexpr (WL) return lw[(KMAX + 1) * WW + lane + 64 * i]; else return bt[i]; };
  auto WK = [&](int k, int i) -> float { if constexpr (WL) return lw[k * WW + lane + 64 * i]; else return wk[k][i]; };
#pragma unroll
  for (int k = 0; k < KMAX; ++k) dbk[k] = 0.f;
  // operands of the wave's next row, fetched before the current row is worked on (see the forward)
  float xn[NC], mean_n = 0.f, rstd_n = 0.f, mine_n = 0.f;
  auto fetch = [&](int64_t r) {
    const bool live = r < rows;
    mean_n = live ? mr[2 * r] : 0.f;
    rstd_n = live ? mr[2 * r + 1] : 0.f;
    mine_n = (live && lane < K) ? dpred[r * K + lane] : 0.f;       // one coalesced load of the row, broadcast below
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xn[i] = (live && c < D) ? x[r * ldx + c] : 0.f;
    }
  };
  fetch(wave);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mean = mean_n, rstd = rstd_n;
    float dp[KMAX], dpq[KMAX];
    const float mine = mine_n;
    float xcur[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) xcur[i] = xn[i];
    fetch(r + nwaves);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      dp[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), k));      // (a scalar read: k is a constant here; a lane shuffle is a trip through the LDS crossbar)
      dpq[k] = round16 ? (float)(bf16_t)dp[k] : dp[k];
      dbk[k] += dp[k];
    }
    float xh[NC], dxh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xh[i] = 0.f; dxh[i] = 0.f;
      if (c < D) {
        xh[i] = (xcur[i] - mean) * rstd;
        if constexpr (DW) {
          float yv = xh[i] * GM(i) + BT(i);
          if (round16) yv = (float)(bf16_t)yv;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) acc[k][i] += dpq[k] * yv;
        } else {
          float g = 0.f;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) g += dpq[k] * WK(k, i);
          dxh[i] = g * GM(i);
          acc[0][i] += g * xh[i];
          acc[1][i] += g;
          s1 += dxh[i];
          s2 += dxh[i] * xh[i];
        }
      }
    }
    if constexpr (!DW) {
      const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < D) dx[r * lddx + c] = rstd * (dxh[i] - m1 - xh[i] * m2);
      }
    }
  }
  float* red = lnfc_red;
  const int W = NC * 64;
  for (int w = 0; w < nwb; ++w) {       // waves take turns adding their sums into the LDS rows
    if (wib == w) {
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int a = 0; a < NA; ++a) red[a * W + c] = (w ? red[a * W + c] : 0.f) + acc[a][i];
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    if constexpr (DW) {
      for (int k = 0; k < K; ++k) atomicAdd(&dfcw[(int64_t)k * D + c], red[k * W + c]);
    } else {
      atomicAdd(&dgamma[c], red[c]);
      atomicAdd(&dbeta[c], red[W + c]);
    }
  }
  if constexpr (DW) {
    // the projection's bias gradient: every lane of a wave holds the same sums; combined over the workgroup's waves,
    // one atomic per output and workgroup (the few workgroups of this role keep the traffic on these K addresses low)
    __syncthreads();
    if (lane == 0) {
      for (int k = 0; k < K; ++k) red[wib * KMAX + k] = dbk[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
      float sum = 0.f;
      for (int w = 0; w < nwb; ++w) sum += red[w * KMAX + threadIdx.x];
      atomicAdd(&dfcb[threadIdx.x], sum);
    }
  }
}

// One wave per row.  The column sums (dgamma, dbeta) are kept per lane over the wave's rows, combined over
// the workgroup's waves in LDS, and added to the result with one atomic per column and workgroup.
template <int NC>
__global__ void layernorm_bwd_kernel(int64_t rows, int D, const float* __restrict__ dy, int64_t lddy,
                                     const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                     const float* __restrict__ mr, float* __restrict__ dx, int64_t lddx,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][NC * 64];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, nwb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: row addresses = scalar base + lane offset
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float dg[NC], db[NC], gm[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    dg[i] = 0.f; db[i] = 0.f;
    const int c = lane + 64 * i;
    gm[i] = c < D ? gamma[c] : 0.f;
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mean = mr[2 * r], rstd = mr[2 * r + 1];
    float xh[NC], dxh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xh[i] = 0.f; dxh[i] = 0.f;
      if (c < D) {
        const float g = dy[r * lddy + c];
        xh[i] = (x[r * ldx + c] - mean) * rstd;
        dxh[i] = g * gm[i];
        dg[i] += g * xh[i];
        db[i] += g;
        s1 += dxh[i];
        s2 += dxh[i] * xh[i];
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) dx[r * lddx + c] = rstd * (dxh[i] - m1 - xh[i] * m2);
    }
  }
  for (int w = 0; w < nwb; ++w) {       // waves take turns adding their sums into the LDS rows
    if (wib == w) {
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        red[0][c] = (w ? red[0][c] : 0.f) + dg[i];
        red[1][c] = (w ? red[1][c] : 0.f) + db[i];
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    atomicAdd(&dgamma[c], red[0][c]);
    atomicAdd(&dbeta[c], red[1][c]);
  }
}

__global__ void gather_rows_kernel(int64_t rows, int64_t len, const float* __restrict__ src, const int64_t* __restrict__ idx,
                                   float* __restrict__ out) {
  const int64_t n = rows * len;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / len, c = e - r * len;
    out[e] = src[idx[r] * len + c];
  }
}

// Window gather: out[b, t, :] = table[idx[b] * stride + t, :] (windows of a [rows, F] table, never materialised).
// One thread per 16 bytes when F, ld and the bases allow it, else per element.
template <bool VEC>
__global__ void gather_windows_kernel(int64_t B, int T, int F, int64_t stride, const float* __restrict__ table, int64_t ld,
                                      const int64_t* __restrict__ idx, float* __restrict__ out) {
  const int fw = VEC ? F / 4 : F;
  const int64_t n = B * T * fw;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = e / fw;
    const int c = (int)(e - bt * fw);
    const int64_t b = bt / T;
    const int t = (int)(bt - b * T);
    const float* src = table + (idx[b] * stride + t) * ld;
    if constexpr (VEC) reinterpret_cast<float4*>(out)[e] = reinterpret_cast<const float4*>(src)[c];
    else out[e] = src[c];
  }
}

// coef (optional): device pointer to {step_size, inv_sqrt_bc2}, read instead of the two arguments -- a captured
// launch then takes the step-dependent bias corrections from memory the host refreshes before every replay
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float b1, float b2, float eps, float wd, float step_size,
                            float inv_sqrt_bc2, const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; inv_sqrt_bc2 = coef[1]; }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float pe = p[e];
    float ge = g[e];
    if (wd != 0.f) ge += wd * pe;
    float me = m[e], ve = v[e];
    me += (ge - me) * (1.f - b1);
    ve = ve * b2 + (1.f - b2) * ge * ge;
    m[e] = me; v[e] = ve;
    const float denom = sqrtf(ve) * inv_sqrt_bc2 + eps;
    p[e] = pe - step_size * (me / denom);
  }
}

// One thread: advance the step counter kept in coef[2] (int32 bits) and write the two step-dependent factors of Adam
// to coef[0..1], in double precision like the host (tad_adam_coefficients).
__global__ void adam_tick_kernel(float lr, float b1, float b2, float* __restrict__ coef) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int* cnt = reinterpret_cast<int*>(coef) + 2;
  const int step = *cnt + 1;
  *cnt = step;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  coef[0] = (float)((double)lr / bc1);
  coef[1] = (float)(1.0 / sqrt(bc2));
}

// out_sum[c] += sum_r dY[r,c]; out_wsum[c] += sum_r dY[r,c] * scale[r]
__global__ void colsum_kernel(int64_t M, int64_t N, const float* __restrict__ dY, int64_t ldy,
                              const float* __restrict__ rs, int64_t rs_stride, float mul, float* __restrict__ o1,
                              float* __restrict__ o2, float* __restrict__ ow, int64_t rows_per_block) {
  __shared__ float p1[4][64], p2[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  float a = 0.f, w = 0.f;
  if (c < N) {
    for (int64_t r = r0 + ty; r < r1; r += 4) {
      const float v = dY[r * ldy + c];
      a += v;
      if (ow) w += v * rs[r * rs_stride];
    }
  }
  p1[ty][tx] = a; p2[ty][tx] = w;
  __syncthreads();
  if (ty == 0 && c < N) {
    a = p1[0][tx] + p1[1][tx] + p1[2][tx] + p1[3][tx];
    w = p2[0][tx] + p2[1][tx] + p2[2][tx] + p2[3][tx];
    if (o1) atomicAdd(&o1[c], a);
    if (o2) atomicAdd(&o2[c], a);
    if (ow) atomicAdd(&ow[c], w * mul);
  }
}

// Column sums of a bf16 matrix, plain and weighted by two per-row scalars, 8 columns per lane (16-byte loads):
// U[0][c] += sum_r s1[r] A[r,c];  U[1][c] += sum_r s2[r] A[r,c];  U[2][c] += sum_r A[r,c].
// LPR lanes (a power of two >= N / 8) serve one row, a wave 64 / LPR rows at a time, four row groups in flight; the
// rows of a wave meet through lane shuffles, the waves of a workgroup in LDS, the workgroups in 3 N global atomics.
template <int LPR>
__global__ __launch_bounds__(256) void wcolsum16_kernel(int64_t M, int N, const bf16_t* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ s1, int64_t st1, const float* __restrict__ s2,
                                                        int64_t st2, float* __restrict__ U) {
  extern __shared__ float wc_part[];                     // [4 waves][3][N]
  constexpr int RPW = 64 / LPR, UN = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, c = 8 * (lane % LPR);
  const bool cok = c < N;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  float a0[8], a1[8], a2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a0[k] = a1[k] = a2[k] = 0.f;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub; r0 < M; r0 += UN * stride) {
    uint4 v[UN];
    float w1[UN], w2[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t r = r0 + u * stride;
      const bool ok = r < M;
      v[u] = (ok && cok) ? *reinterpret_cast<const uint4*>(A + r * lda + c) : make_uint4(0u, 0u, 0u, 0u);
      w1[u] = ok ? s1[r * st1] : 0.f;
      w2[u] = ok ? s2[r * st2] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const uint32_t q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(q[k] << 16), hi = __uint_as_float(q[k] & 0xffff0000u);
        a0[2 * k] += lo; a0[2 * k + 1] += hi;
        a1[2 * k] = fmaf(w1[u], lo, a1[2 * k]); a1[2 * k + 1] = fmaf(w1[u], hi, a1[2 * k + 1]);
        a2[2 * k] = fmaf(w2[u], lo, a2[2 * k]); a2[2 * k + 1] = fmaf(w2[u], hi, a2[2 * k + 1]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) {
      a0[k] += __shfl_xor(a0[k], m); a1[k] += __shfl_xor(a1[k], m); a2[k] += __shfl_xor(a2[k], m);
    }
  }
  if (sub == 0 && cok) {
    float* pw = wc_part + (size_t)wave * 3 * N;
#pragma unroll
    for (int k = 0; k < 8; ++k) { pw[c + k] = a1[k]; pw[N + c + k] = a2[k]; pw[2 * N + c + k] = a0[k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * N; i += 256) {
    const float t = wc_part[i] + wc_part[3 * N + i] + wc_part[6 * N + i] + wc_part[9 * N + i];
    if (t != 0.f) atomicAdd(&U[i], t);
  }
}

// Gradients of two Linear(1 -> L) maps whose outputs are columns [col1, col1 + L) and [col2, col2 + L) of the input of a
// Linear with weight W [N, ldw]: with g the gradient of that Linear's pre-activations, the bias gradient is
// W[:, col..]^T (sum_r g[r]) and the weight gradient W[:, col..]^T (sum_r s[r] g[r]) -- U holds those three row sums.
// blockIdx.x: the map; blockIdx.y: 8 rows of W (their loads are in flight together); one thread per output column.
__global__ void enc_grad_finish_kernel(int N, int L, const float* __restrict__ W, int64_t ldw, int col1, int col2,
                                       const float* __restrict__ U, float mul1, float mul2, int round16, float* __restrict__ gb1,
                                       float* __restrict__ gw1, float* __restrict__ gb2, float* __restrict__ gw2) {
  const int which = blockIdx.x, o0 = 8 * blockIdx.y;
  const int col = which ? col2 : col1;
  float* gb = which ? gb2 : gb1;
  float* gw = which ? gw2 : gw1;
  const float mul = which ? mul2 : mul1;
  float us[8], uw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { us[k] = U[2 * N + o0 + k]; uw[k] = U[which * N + o0 + k]; }
  for (int c = threadIdx.x; c < L; c += blockDim.x) {
    float wv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wv[k] = W[(int64_t)(o0 + k) * ldw + col + c];
    float ab = 0.f, aw = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = round16 ? (float)(bf16_t)wv[k] : wv[k];      // the operand the product form of this gradient multiplies by
      ab = fmaf(w, us[k], ab);
      aw = fmaf(w, uw[k], aw);
    }
    atomicAdd(&gb[c], ab);
    atomicAdd(&gw[c], aw * mul);
  }
}

}  // namespace

#define EW_STREAM ((hipStream_t)stream)

extern "C" int tad_add_noise(int64_t B, int64_t T, int64_t K, const float* x0, const float* noise, const int32_t* i_idx,
                             const float* alphas, float* x_noisy, float* i_f, tad_stream_t stream) {
  TAD_CHECK(B >= 0 && T > 0 && K > 0 && x0 && noise && i_idx && alphas && x_noisy, "tad_add_noise: bad arguments");
  if (B == 0) return TAD_OK;
  hipLaunchKernelGGL(add_noise_kernel, dim3(ew_grid(B * T * K)), dim3(EW_BLOCK), 0, EW_STREAM, B, T, K, x0, noise, i_idx,
                     alphas, x_noisy, i_f);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_sampler_update(int64_t n, float* x, const float* pred, const float* z, float c1, float c2, float c3,
                                  int last, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && x && pred && z, "tad_sampler_update: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(sampler_update_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, x, pred, z, c1, c2, c3, last);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

static int sampler_tail(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                        const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                        float c1, float c2, float c3, int last, const float* w_in, const float* b_in, float* xh,
                        int64_t ldxh, bf16_t* xh16, tad_stream_t stream, const float* vrow = nullptr, const float* pinv = nullptr,
                        int64_t ldp = 0, TailNoise nz = TailNoise{}) {
  TAD_CHECK(rows >= 0 && D > 0 && D <= 1024 && K > 0 && K <= 8 && H > 0 && H <= 1024, "tad_sampler_tail: supported D <= 1024, K <= 8, H <= 1024 (got D=%d K=%d H=%d)", D, K, H);
  TAD_CHECK(h && gamma && beta && fc_w && fc_b && x && (z || nz.on) && w_in && b_in && (xh || xh16) && ldh >= D && ldxh >= H, "tad_sampler_tail: bad arguments");
  if (rows == 0) return TAD_OK;
  int64_t g = cdiv64(rows, EW_BLOCK / 64);
  if (g > 512) g = 512;          // two workgroups per CU: the prologue (weights into LDS) is a third of a 25 k-row launch
  int nc = (int)cdiv64(D, 64);                      // 64-column blocks of the recurrent output a lane holds
  nc = nc <= 2 ? 2 : nc <= 8 ? 8 : 16;
  const int nh = (int)cdiv64(H, 64);                // ... of the projection: half of nc in the reference's models (D = 2 H)
  const size_t lds = sizeof(float) * ((size_t)H * 10 + 3 + (size_t)10 * 64 * nc);
  TAD_CHECK(lds <= 65536, "tad_sampler_tail: 10 H + 640 n exceeds the launch's LDS (D=%d K=%d H=%d)", D, K, H);
  const int r16 = prec == TAD_BF16 ? 1 : 0;
#define TAIL(NC, KT, NP) hipLaunchKernelGGL((sampler_tail_kernel<NC, KT, NP>), dim3((unsigned)g), dim3(EW_BLOCK), lds, EW_STREAM, rows, D, K, H, r16, h, ldh, gamma, beta, eps, fc_w, fc_b, x, z, nz, c1, c2, c3, last, w_in, b_in, xh, ldxh, xh16, vrow, pinv, ldp)
#define TAIL_K(NC, NP) do { if (K == 6) TAIL(NC, 6, NP); else TAIL(NC, 0, NP); } while (0)      /* (6: the latent width of every reference notebook) */
  if (nc == 2) { if (nh <= 1) TAIL_K(2, 1); else TAIL_K(2, 16); }
  else if (nc == 8) { if (nh <= 4) TAIL_K(8, 4); else TAIL_K(8, 16); }
  else { if (nh <= 8) TAIL_K(16, 8); else TAIL_K(16, 16); }
#undef TAIL_K
#undef TAIL
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_sampler_tail(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                                const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                                float c1, float c2, float c3, int last, const float* w_in, const float* b_in, float* xh,
                                int64_t ldxh, tad_stream_t stream) {
  return sampler_tail(prec, rows, D, K, H, h, ldh, gamma, beta, eps, fc_w, fc_b, x, z, c1, c2, c3, last, w_in, b_in, xh, ldxh, nullptr, stream);
}

extern "C" int tad_sampler_tail_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                                    const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                                    float c1, float c2, float c3, int last, const float* w_in, const float* b_in, void* xh16,
                                    int64_t ldxh, tad_stream_t stream) {
  return sampler_tail(prec, rows, D, K, H, h, ldh, gamma, beta, eps, fc_w, fc_b, x, z, c1, c2, c3, last, w_in, b_in, nullptr, ldxh,
                      reinterpret_cast<bf16_t*>(xh16), stream);
}

extern "C" int tad_sampler_tail_proj_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                                         const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                                         float c1, float c2, float c3, int last, const float* w_fold, const float* b_fold,
                                         const float* vrow, const float* pinv, int64_t ld_pinv, void* pa16, int64_t ldpa,
                                         tad_stream_t stream) {
  TAD_CHECK(vrow && pinv && ld_pinv >= H && pa16, "tad_sampler_tail_proj_b16: bad arguments");
  return sampler_tail(prec, rows, D, K, H, h, ldh, gamma, beta, eps, fc_w, fc_b, x, z, c1, c2, c3, last, w_fold, b_fold, nullptr, ldpa,
                      reinterpret_cast<bf16_t*>(pa16), stream, vrow, pinv, ld_pinv);
}

extern "C" int tad_sampler_tail_proj_cn_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                                            const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, uint64_t seed,
                                            uint64_t counter, int64_t row0, float c1, float c2, float c3, int last, const float* w_fold,
                                            const float* b_fold, const float* vrow, const float* pinv, int64_t ld_pinv, void* pa16,
                                            int64_t ldpa, tad_stream_t stream) {
  TAD_CHECK(vrow && pinv && ld_pinv >= H && pa16 && row0 >= 0 && counter < (1ull << 32), "tad_sampler_tail_proj_cn_b16: bad arguments");
  TailNoise nz{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)counter, 1, row0};
  return sampler_tail(prec, rows, D, K, H, h, ldh, gamma, beta, eps, fc_w, fc_b, x, nullptr, c1, c2, c3, last, w_fold, b_fold, nullptr, ldpa,
                      reinterpret_cast<bf16_t*>(pa16), stream, vrow, pinv, ld_pinv, nz);
}

extern "C" int tad_mse_loss(int64_t n, const float* pred, const float* target, float grad_scale, float* dpred,
                            float* loss_sum, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && pred && target, "tad_mse_loss: bad arguments");
  if (n == 0) return TAD_OK;
  unsigned g = ew_grid(n, 4);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(mse_loss_kernel, dim3(g), dim3(EW_BLOCK), 0, EW_STREAM, n, pred, target, grad_scale, dpred, loss_sum);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

// loss_sum[0] += sum |pred - target|; dpred = grad_scale * sign(pred - target) (0 where they are equal, as torch's L1Loss)
__global__ void l1_loss_kernel(int64_t n, const float* __restrict__ pred, const float* __restrict__ target, float gs,
                               float* __restrict__ dpred, float* __restrict__ loss_sum) {
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[e] - target[e];
    acc += fabsf(d);
    if (dpred) dpred[e] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
  }
  if (loss_sum) block_atomic_sum(acc, loss_sum);
}

extern "C" int tad_l1_loss(int64_t n, const float* pred, const float* target, float grad_scale, float* dpred, float* loss_sum,
                           tad_stream_t stream) {
  TAD_CHECK(n >= 0 && (n == 0 || (pred && target)), "tad_l1_loss: bad arguments");
  if (n == 0) return TAD_OK;
  unsigned g = ew_grid(n, 4);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(l1_loss_kernel, dim3(g), dim3(EW_BLOCK), 0, EW_STREAM, n, pred, target, grad_scale, dpred, loss_sum);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_grad_norm_sq(int64_t n, const float* g, float* out_sum, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && out_sum && (n == 0 || g), "tad_grad_norm_sq: bad arguments");
  TAD_CHECK((reinterpret_cast<uintptr_t>(g) & 15) == 0, "tad_grad_norm_sq: the buffer must be 16-byte aligned");
  if (n == 0) return TAD_OK;
  unsigned gr = ew_grid(n, 16);
  if (gr > 1024) gr = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(gr), dim3(EW_BLOCK), 0, EW_STREAM, n, g, out_sum);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

static int sincos_features(int64_t rows, int n_nums, const float* x, int64_t ldx, void* out, int out_dtype, int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && n_nums > 0 && x && out && ldx >= n_nums && ldo >= 16 * n_nums, "tad_sincos_features: bad arguments");
  if (rows == 0) return TAD_OK;
  const int64_t n8 = rows * n_nums;
  const bool wide = wide_rows_ok(out, ldo, out_dtype == TAD_BF16 ? 2 : 4, 16) && cdiv64(n8, EW_BLOCK) < ((int64_t)1 << 31);
  if (wide && out_dtype == TAD_BF16)
    hipLaunchKernelGGL(sincos8_kernel<bf16_t>, dim3((unsigned)cdiv64(n8, EW_BLOCK)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_nums, x, ldx, (bf16_t*)out, ldo);
  else if (wide)
    hipLaunchKernelGGL(sincos8_kernel<float>, dim3((unsigned)cdiv64(n8, EW_BLOCK)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_nums, x, ldx, (float*)out, ldo);
  else if (out_dtype == TAD_BF16)
    hipLaunchKernelGGL(sincos_kernel<bf16_t>, dim3(ew_grid(rows * n_nums * 8)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_nums, x, ldx, (bf16_t*)out, ldo);
  else
    hipLaunchKernelGGL(sincos_kernel<float>, dim3(ew_grid(rows * n_nums * 8)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_nums, x, ldx, (float*)out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
extern "C" int tad_sincos_features(int64_t rows, int n_nums, const float* x, int64_t ldx, float* out, int64_t ldo,
                                   tad_stream_t stream) {
  return sincos_features(rows, n_nums, x, ldx, out, TAD_F32, ldo, stream);
}
extern "C" int tad_sincos_features_b16(int64_t rows, int n_nums, const float* x, int64_t ldx, void* out16, int64_t ldo,
                                       tad_stream_t stream) {
  return sincos_features(rows, n_nums, x, ldx, out16, TAD_BF16, ldo, stream);
}

static int fill_meta(DiscMeta& m, int n_disc, const int32_t* tab_off, const int32_t* cards, const char* who) {
  if (n_disc <= 0 || n_disc > MAX_DISC || !tab_off || !cards) {
    tad_set_error("%s: n_disc=%d out of range (1..%d) or null metadata", who, n_disc, MAX_DISC);
    return TAD_ERR_INVALID;
  }
  for (int d = 0; d < n_disc; ++d) {
    if (cards[d] <= 0 || tab_off[d] < 0) {
      tad_set_error("%s: bad table metadata for column %d", who, d);
      return TAD_ERR_INVALID;
    }
    m.off[d] = tab_off[d];
    m.card[d] = cards[d];
  }
  return TAD_OK;
}

static int embed_fwd(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* tables,
                     const int32_t* tab_off, const int32_t* cards, void* out, int out_dtype, int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && E > 0 && x && tables && out && ldx >= n_disc && ldo >= (int64_t)n_disc * E, "tad_embed_fwd: bad arguments");
  DiscMeta m{};
  if (int rc = fill_meta(m, n_disc, tab_off, cards, "tad_embed_fwd")) return rc;
  if (rows == 0) return TAD_OK;
  const int64_t n8 = rows * n_disc * (E / 8);
  const bool wide = E % 8 == 0 && (reinterpret_cast<uintptr_t>(tables) & 15) == 0 && wide_rows_ok(out, ldo, out_dtype == TAD_BF16 ? 2 : 4, 16) &&
                    cdiv64(n8, EW_BLOCK) < ((int64_t)1 << 31);
  if (wide && out_dtype == TAD_BF16)
    hipLaunchKernelGGL(embed_fwd8_kernel<bf16_t>, dim3((unsigned)cdiv64(n8, EW_BLOCK)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_disc, E, x, ldx,
                       tables, m, (bf16_t*)out, ldo);
  else if (wide)
    hipLaunchKernelGGL(embed_fwd8_kernel<float>, dim3((unsigned)cdiv64(n8, EW_BLOCK)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_disc, E, x, ldx,
                       tables, m, (float*)out, ldo);
  else if (out_dtype == TAD_BF16)
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(ew_grid(rows * n_disc * E)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_disc, E, x, ldx,
                       tables, m, (bf16_t*)out, ldo);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(ew_grid(rows * n_disc * E)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_disc, E, x, ldx,
                       tables, m, (float*)out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
extern "C" int tad_embed_fwd(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* tables,
                             const int32_t* tab_off, const int32_t* cards, float* out, int64_t ldo, tad_stream_t stream) {
  return embed_fwd(rows, n_disc, E, x, ldx, tables, tab_off, cards, out, TAD_F32, ldo, stream);
}
extern "C" int tad_embed_fwd_b16(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* tables,
                                 const int32_t* tab_off, const int32_t* cards, void* out16, int64_t ldo, tad_stream_t stream) {
  return embed_fwd(rows, n_disc, E, x, ldx, tables, tab_off, cards, out16, TAD_BF16, ldo, stream);
}

extern "C" int tad_embed_bwd(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* dout, int64_t ldo,
                             float* dtables, const int32_t* tab_off, const int32_t* cards, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && E > 0 && x && dout && dtables && ldx >= n_disc && ldo >= (int64_t)n_disc * E, "tad_embed_bwd: bad arguments");
  DiscMeta m{};
  if (int rc = fill_meta(m, n_disc, tab_off, cards, "tad_embed_bwd")) return rc;
  if (rows == 0) return TAD_OK;
  int maxcard = 0;
  for (int d = 0; d < n_disc; ++d) maxcard = cards[d] > maxcard ? cards[d] : maxcard;
  const size_t lds = (size_t)maxcard * E * sizeof(float);
  static const char* reg_env = getenv("TAD_EMBED_BWD_REG");      // experiments: 0 = always the LDS-atomic kernel
  if (maxcard <= EB_CARD && E <= 128 && rows >= 1024 && !(reg_env && atoi(reg_env) == 0)) {
    int64_t nb = cdiv64(rows, 4 * 16);                   // at least 16 rows per wave
    static const char* cap_env = getenv("TAD_EMBED_BWD_WGS");
    const int64_t cap = cap_env && atoi(cap_env) > 0 ? atoi(cap_env) : 512;
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(embed_bwd_reg_kernel, dim3((unsigned)nb, (unsigned)n_disc), dim3(256), 0, EW_STREAM, rows, n_disc, E, x, ldx, dout, ldo,
                       dtables, m);
  } else if (lds <= 48 * 1024) {
    int64_t nb = cdiv64(rows, 256);
    if (nb > 512) nb = 512;
    const int64_t rpb = cdiv64(rows, nb);
    nb = cdiv64(rows, rpb);
    hipLaunchKernelGGL(embed_bwd_lds_kernel, dim3((unsigned)nb, (unsigned)n_disc), dim3(EW_BLOCK), lds, EW_STREAM, rows, n_disc, E,
                       x, ldx, dout, ldo, dtables, m, rpb);
  } else {
    hipLaunchKernelGGL(embed_bwd_direct_kernel, dim3(ew_grid(rows * n_disc * E)), dim3(EW_BLOCK), 0, EW_STREAM, rows, n_disc, E, x,
                       ldx, dout, ldo, dtables, m);
  }
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

template <typename OT>
static int posenc_fwd(int64_t rows, int H, const float* t, int64_t t_stride, float inv_max, const float* w,
                      const float* b, const float* scale, const float* shift, OT* out, int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && H >= 2 && t && w && b && scale && shift && out && ldo >= H, "tad_posenc_fwd: bad arguments");
  if (rows == 0) return TAD_OK;
  const int64_t n4 = rows * (H / 8);
  const bool tables16 = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(scale) |
                          reinterpret_cast<uintptr_t>(shift)) & 15) == 0;
  if (H % 8 == 0 && tables16 && wide_rows_ok(out, ldo, (int)sizeof(OT), 4 * (int)sizeof(OT)) && cdiv64(n4, EW_BLOCK) < ((int64_t)1 << 31))
    hipLaunchKernelGGL((posenc4_kernel<OT>), dim3((unsigned)cdiv64(n4, EW_BLOCK), 2), dim3(EW_BLOCK), 0, EW_STREAM, rows, H, t, t_stride, inv_max,
                       w, b, scale, shift, out, ldo);
  else if (rows * H + (int64_t)4096 * EW_BLOCK < ((int64_t)1 << 32))       // the grid-stride counter must not wrap either
    hipLaunchKernelGGL((posenc_kernel<uint32_t, OT>), dim3(ew_grid(rows * H)), dim3(EW_BLOCK), 0, EW_STREAM, rows, H, t, t_stride, inv_max,
                       w, b, scale, shift, out, ldo);
  else
    hipLaunchKernelGGL((posenc_kernel<int64_t, OT>), dim3(ew_grid(rows * H)), dim3(EW_BLOCK), 0, EW_STREAM, rows, H, t, t_stride, inv_max,
                       w, b, scale, shift, out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
extern "C" int tad_posenc_fwd(int64_t rows, int H, const float* t, int64_t t_stride, float inv_max, const float* w,
                              const float* b, const float* scale, const float* shift, float* out, int64_t ldo,
                              tad_stream_t stream) {
  return posenc_fwd<float>(rows, H, t, t_stride, inv_max, w, b, scale, shift, out, ldo, stream);
}
extern "C" int tad_posenc_fwd_b16(int64_t rows, int H, const float* t, int64_t t_stride, float inv_max, const float* w,
                                  const float* b, const float* scale, const float* shift, void* out16, int64_t ldo,
                                  tad_stream_t stream) {
  return posenc_fwd<bf16_t>(rows, H, t, t_stride, inv_max, w, b, scale, shift, (bf16_t*)out16, ldo, stream);
}

extern "C" int tad_silu_fwd(int64_t n, const float* z, float* a, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && z && a, "tad_silu_fwd: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(silu_fwd_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, z, a);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
extern "C" int tad_silu_fwd_b16(int64_t n, const float* z, void* a16, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && z && a16, "tad_silu_fwd_b16: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(silu_fwd_kernel<bf16_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, z, (bf16_t*)a16);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_silu_bwd(int64_t n, const float* z, const float* da, float* dz, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && z && da && dz, "tad_silu_bwd: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(silu_bwd_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, z, da, dz);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
extern "C" int tad_silu_bwd_b16(int64_t n, const float* z, const float* da, void* dz16, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && z && da && dz16, "tad_silu_bwd_b16: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(silu_bwd_kernel<bf16_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, z, da, (bf16_t*)dz16);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_layernorm_fwd(int64_t rows, int D, const float* x, int64_t ldx, const float* gamma, const float* beta,
                                 float eps, float* y, int64_t ldy, float* mean_rstd, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && D > 0 && x && gamma && beta && y && ldx >= D && ldy >= D, "tad_layernorm_fwd: bad arguments");
  if (rows == 0) return TAD_OK;
  int64_t g = cdiv64(rows, EW_BLOCK / 64);
  if (g > 4096) g = 4096;
  const int nc = (int)cdiv64(D, 64);
  TAD_CHECK(nc <= 32, "tad_layernorm_fwd: D = %d too large (<= 2048)", D);
#define LN_FWD(NC) hipLaunchKernelGGL(layernorm_fwd_kernel<NC>, dim3((unsigned)g), dim3(EW_BLOCK), 0, EW_STREAM, rows, D, x, ldx, gamma, beta, eps, y, ldy, mean_rstd)
  if (nc <= 2) LN_FWD(2);
  else if (nc <= 8) LN_FWD(8);
  else if (nc <= 16) LN_FWD(16);
  else LN_FWD(32);
#undef LN_FWD
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_layernorm_bwd(int64_t rows, int D, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                                 const float* gamma, const float* mean_rstd, float* dx, int64_t lddx, float* dgamma,
                                 float* dbeta, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && D > 0 && D <= 2048 && dy && x && gamma && mean_rstd && dx && dgamma && dbeta, "tad_layernorm_bwd: bad arguments (D <= 2048)");
  TAD_CHECK(lddy >= D && ldx >= D && lddx >= D, "tad_layernorm_bwd: bad leading dimension");
  if (rows == 0) return TAD_OK;
  int64_t g = cdiv64(rows, EW_BLOCK / 64);
  // six workgroups per CU: measured 139 us at 1536 workgroups against 157 (1024), 174 (2048), 178 (4096) for 98 304 x 400
  static const char* cap_env = getenv("TAD_LN_BWD_GRID");     // experiments: workgroups of the launch
  const int64_t cap = cap_env ? atoi(cap_env) : 1536;
  if (g > cap) g = cap;
  const int nc = (int)cdiv64(D, 64);
#define LN_BWD(NC) hipLaunchKernelGGL(layernorm_bwd_kernel<NC>, dim3((unsigned)g), dim3(EW_BLOCK), 0, EW_STREAM, rows, D, dy, lddy, x, ldx, gamma, mean_rstd, dx, lddx, dgamma, dbeta)
  if (nc <= 2) LN_BWD(2);
  else if (nc <= 8) LN_BWD(8);
  else if (nc <= 16) LN_BWD(16);
  else LN_BWD(32);
#undef LN_BWD
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_ln_fc_fwd(int prec, int64_t rows, int D, int K, const float* x, int64_t ldx, const float* gamma, const float* beta,
                             float eps, const float* fc_w, const float* fc_b, float* y, int64_t ldy, float* mean_rstd, float* pred,
                             tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && D > 0 && D <= 1024 && K > 0 && K <= 8, "tad_ln_fc_fwd: supported D <= 1024, K <= 8 (got D=%d K=%d)", D, K);
  TAD_CHECK(x && gamma && beta && fc_w && fc_b && mean_rstd && pred && ldx >= D && (!y || ldy >= D), "tad_ln_fc_fwd: bad arguments");
  if (rows == 0) return TAD_OK;
  int64_t g = cdiv64(rows, EW_BLOCK / 64);
  if (g > 2048) g = 2048;
  const int r16 = prec == TAD_BF16 ? 1 : 0;
#define LNFC(NC) hipLaunchKernelGGL((ln_fc_fwd_kernel<NC, 8>), dim3((unsigned)g), dim3(EW_BLOCK), 0, EW_STREAM, rows, D, K, r16, x, ldx, gamma, beta, eps, fc_w, fc_b, y, ldy, mean_rstd, pred)
  if (D <= 128) LNFC(2);
  else if (D > 384 && D <= 448) LNFC(7);      // 2H = 400 of the bench: 7 columns per lane, not 8
  else if (D <= 512) LNFC(8);
  else LNFC(16);                              // 2H = 1024 (hidden size 512): one wave per SIMD, the next row in flight
#undef LNFC
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_ln_fc_bwd(int prec, int64_t rows, int D, int K, const float* dpred, const float* x, int64_t ldx, const float* gamma,
                             const float* beta, const float* mean_rstd, const float* fc_w, float* dx, int64_t lddx, float* dgamma,
                             float* dbeta, float* dfc_w, float* dfc_b, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && D > 0 && D <= 1024 && K > 0 && K <= 8, "tad_ln_fc_bwd: supported D <= 1024, K <= 8 (got D=%d K=%d)", D, K);
  TAD_CHECK(dpred && x && gamma && beta && mean_rstd && fc_w && dx && dgamma && dbeta && dfc_w && dfc_b && ldx >= D && lddx >= D,
            "tad_ln_fc_bwd: bad arguments");
  if (rows == 0) return TAD_OK;
  static const char* cap_env = getenv("TAD_LNFC_BWD_GRID");     // experiments: workgroups of the two launches
  static const char* cap2_env = getenv("TAD_LNFC_DW_GRID");
  // every workgroup ends with 2 D (first launch) or K D (second) atomics onto the same addresses: at least 4 rows per
  // wave in both (2400 rows x 400, the pair: 48 us at 1 / 1 rows per wave, 36 at 1 / 4, 29.5 at 2 / 4 and 4 / 8,
  // 27 at 4 / 4, 32 at 8 / 8); the caps below bind from 24 576 / 12 288 rows on
  static const char* rpw_env = getenv("TAD_LNFC_RPW");          // experiments: "a,b" = rows per wave of the two launches
  int rpw1 = 4, rpw2 = 4;
  if (rpw_env) sscanf(rpw_env, "%d,%d", &rpw1, &rpw2);
  int64_t g = cdiv64(rows, (int64_t)rpw1 * (EW_BLOCK / 64)), g2 = cdiv64(rows, (int64_t)rpw2 * (EW_BLOCK / 64));
  const int64_t cap = cap_env ? atoi(cap_env) : 1536, cap2 = cap2_env ? atoi(cap2_env) : 768;     // measured at 98 304 x 400: 246 us; (1536, 256) 390, (1024, 1024) 316
  if (g > cap) g = cap;
  if (g2 > cap2) g2 = cap2;
  const int r16 = prec == TAD_BF16 ? 1 : 0;
#define LNFC(NC, DWV, G, NAV) hipLaunchKernelGGL((ln_fc_bwd_kernel<NC, 8, DWV>), dim3((unsigned)(G)), dim3(EW_BLOCK), sizeof(float) * ((NAV) + ((NC > 8 && !DWV) ? 10 : 0)) * NC * 64, EW_STREAM, rows, D, K, r16, dpred, x, ldx, gamma, beta, mean_rstd, fc_w, dx, lddx, dgamma, dbeta, dfc_w, dfc_b)
  if (D <= 128) { LNFC(2, false, g, 2); LNFC(2, true, g2, 8); }
  else if (D > 384 && D <= 448) { LNFC(7, false, g, 2); LNFC(7, true, g2, 8); }
  else if (D <= 512) { LNFC(8, false, g, 2); LNFC(8, true, g2, 8); }
  else { LNFC(16, false, g, 2); LNFC(16, true, g2, 8); }
#undef LNFC
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_gather_rows(int64_t rows, int64_t row_len, const float* src, const int64_t* idx, float* out,
                               tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && row_len > 0 && src && idx && out, "tad_gather_rows: bad arguments");
  if (rows == 0) return TAD_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(rows * row_len)), dim3(EW_BLOCK), 0, EW_STREAM, rows, row_len, src, idx, out);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_gather_windows(int64_t B, int64_t T, int64_t F, int64_t stride, const float* table, int64_t ld,
                                  const int64_t* idx, float* out, tad_stream_t stream) {
  TAD_CHECK(B >= 0 && T > 0 && F > 0 && stride > 0 && ld >= F && T < (1 << 30) && F < (1 << 30) && table && idx && out,
            "tad_gather_windows: bad arguments B=%ld T=%ld F=%ld stride=%ld ld=%ld", (long)B, (long)T, (long)F, (long)stride, (long)ld);
  if (B == 0) return TAD_OK;
  const bool vec = F % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(gather_windows_kernel<true>, dim3(ew_grid(B * T * (F / 4))), dim3(EW_BLOCK), 0, EW_STREAM, B, (int)T, (int)F, stride,
                       table, ld, idx, out);
  else
    hipLaunchKernelGGL(gather_windows_kernel<false>, dim3(ew_grid(B * T * F)), dim3(EW_BLOCK), 0, EW_STREAM, B, (int)T, (int)F, stride, table,
                       ld, idx, out);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && p && g && m && v && step >= 1, "tad_adam_step: bad arguments");
  if (n == 0) return TAD_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, p, g, m, v, beta1, beta2, eps, weight_decay,
                     step_size, inv_sqrt_bc2, (const float*)nullptr);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" void tad_adam_coefficients(float lr, float beta1, float beta2, int step, float* out2) {
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  out2[0] = (float)((double)lr / bc1);
  out2[1] = (float)(1.0 / sqrt(bc2));
}

extern "C" int tad_adam_step_dev(int64_t n, float* p, const float* g, float* m, float* v, float beta1, float beta2, float eps,
                                 float weight_decay, const float* coef2, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && p && g && m && v && coef2, "tad_adam_step_dev: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, EW_STREAM, n, p, g, m, v, beta1, beta2, eps, weight_decay,
                     0.f, 0.f, coef2);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_adam_tick(float lr, float beta1, float beta2, float* coef4, tad_stream_t stream) {
  TAD_CHECK(coef4 && (reinterpret_cast<uintptr_t>(coef4) & 3) == 0, "tad_adam_tick: bad coef4");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, EW_STREAM, lr, beta1, beta2, coef4);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_enc_grad_b16(int64_t M, int N, const void* g16, int64_t ldg, const float* s1, int64_t s1_stride, const float* s2,
                                int64_t s2_stride, const float* W, int64_t ldw, int col1, int col2, int L, float mul1, float mul2,
                                int round_w, float* sums, float* gb1, float* gw1, float* gb2, float* gw2, tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && N % 8 == 0 && N <= 512 && g16 && ldg >= N && ldg % 8 == 0 && s1 && s2 && W && sums && gb1 && gw1 && gb2 && gw2,
            "tad_enc_grad_b16: bad arguments (N a multiple of 8, at most 512; ldg a multiple of 8)");
  TAD_CHECK((reinterpret_cast<uintptr_t>(g16) & 15) == 0, "tad_enc_grad_b16: g16 must be 16-byte aligned");
  TAD_CHECK(L > 0 && col1 >= 0 && col2 >= 0 && col1 + L <= ldw && col2 + L <= ldw, "tad_enc_grad_b16: column ranges outside W");
  if (M == 0) return TAD_OK;
  if (tad_zero_async(sums, (size_t)3 * N * sizeof(float), EW_STREAM) != hipSuccess) { tad_set_error("tad_enc_grad_b16: zeroing the sums failed"); return TAD_ERR_LAUNCH; }
  const int lpr = N <= 64 ? 8 : N <= 128 ? 16 : N <= 256 ? 32 : 64;
  const int rpw = 64 / lpr;
  int64_t wgs = cdiv64(M, (int64_t)4 * rpw * 4);          // at least four row groups per wave
  if (wgs > 512) wgs = 512;
  const size_t lds = (size_t)4 * 3 * N * sizeof(float);
  const bf16_t* g = (const bf16_t*)g16;
  switch (lpr) {
    case 8: hipLaunchKernelGGL(wcolsum16_kernel<8>, dim3((unsigned)wgs), dim3(256), lds, EW_STREAM, M, N, g, ldg, s1, s1_stride, s2, s2_stride, sums); break;
    case 16: hipLaunchKernelGGL(wcolsum16_kernel<16>, dim3((unsigned)wgs), dim3(256), lds, EW_STREAM, M, N, g, ldg, s1, s1_stride, s2, s2_stride, sums); break;
    case 32: hipLaunchKernelGGL(wcolsum16_kernel<32>, dim3((unsigned)wgs), dim3(256), lds, EW_STREAM, M, N, g, ldg, s1, s1_stride, s2, s2_stride, sums); break;
    default: hipLaunchKernelGGL(wcolsum16_kernel<64>, dim3((unsigned)wgs), dim3(256), lds, EW_STREAM, M, N, g, ldg, s1, s1_stride, s2, s2_stride, sums); break;
  }
  TAD_LAUNCH_CHECK();
  hipLaunchKernelGGL(enc_grad_finish_kernel, dim3(2, (unsigned)(N / 8)), dim3(128), 0, EW_STREAM, N, L, W, ldw, col1, col2, sums, mul1, mul2,
                     round_w, gb1, gw1, gb2, gw2);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_colsum(int64_t M, int64_t N, const float* dY, int64_t ldy, const float* row_scale,
                          int64_t row_scale_stride, float scale_mul, float* out_sum, float* out_sum2, float* out_wsum,
                          tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && dY && ldy >= N, "tad_colsum: bad arguments");
  TAD_CHECK(!out_wsum || row_scale, "tad_colsum: out_wsum needs row_scale");
  if (M == 0) return TAD_OK;
  const int64_t ctiles = cdiv64(N, 64);
  int64_t slabs = cdiv64(1024, ctiles);
  if (slabs > cdiv64(M, 64)) slabs = cdiv64(M, 64);
  if (slabs < 1) slabs = 1;
  const int64_t rpb = cdiv64(M, slabs);
  slabs = cdiv64(M, rpb);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ctiles, (unsigned)slabs), dim3(EW_BLOCK), 0, EW_STREAM, M, N, dY, ldy, row_scale,
                     row_scale_stride, scale_mul, out_sum, out_sum2, out_wsum, rpb);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
