// Persistent LSTM / GRU sequence kernels for gfx950 (forward and BPTT).
//
// One workgroup owns 32 batch rows of one direction (or head) and walks all T steps.
// Per step the recurrent product is an MFMA GEMM whose A operand (h_{t-1}, or the gate
// gradients in the backward) lives in LDS in MFMA operand type and whose B operand
// (W_hh) is streamed from L2 in a pre-packed, fragment-ordered image: every wave
// instruction reads one contiguous 1 KiB block (64 lanes x 16 B).  Wave w owns the
// hidden-unit chunks j = w, w+NW, ... (32 units each): for a chunk it holds the G gate
// accumulators of the same (row, unit) elements in the same lanes/registers, so the
// gate nonlinearities, the cell update and the state (c or h, kept in registers for the
// whole sequence) need no cross-lane traffic.
//
// Accumulator layout (v_mfma_f32_32x32x*): column = lane & 31 = unit within the chunk,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = batch row within the tile.
#include "common.h"
#include "rnn_layout.h"
#include <stdlib.h>
#include <type_traits>

// Diagnostic build only (-DEXP_STAMPS): cycle counts of the phases of a forward step, summed over the steps
// of wave 0 of every workgroup, in a buffer of their own that no kernel reads.
#ifdef EXP_STAMPS
__device__ unsigned long long g_rnn_stamps[8];
#define RSTAMP(var) __builtin_amdgcn_sched_barrier(0); unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0)
extern "C" int tad_dbg_rnn_stamps(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rnn_stamps), 64);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_stamps), z, 64); }
  return 0;
}
#else
#define RSTAMP(var)
#endif

namespace {

constexpr int MAXDIR = 2;

__host__ __device__ inline int ks_of(bool bf16) { return bf16 ? 16 : 8; }

// ---------------------------------------------------------------------------
// weight packing
struct PackArgs {
  const float* w;   // [G*H, H]
  uint4* out;
  int H, G, NJ, nk, width;
  int bf16, backward;
};

__device__ __forceinline__ uint32_t pk2(float a, float b) {
  bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

// forward image : blocks [j][kk][g], B(k = u_in, n = unit 32j+c of gate g)  = W[(g*H + unit)*H + u_in]
// backward image: blocks [j][kk],    B(k = g*width + u, n = u_in = 32j+c)    = W[(g*H + u)*H + u_in]
__device__ __forceinline__ void pack_elem(const PackArgs& a, int64_t gid) {
  const int lane = gid & 63;
  int64_t blk = gid >> 6;
  const int ks = a.bf16 ? 16 : 8;
  const int per = a.bf16 ? 8 : 4;
  const int64_t nblocks = a.backward ? (int64_t)a.NJ * a.nk : (int64_t)a.NJ * a.nk * a.G;
  if (blk >= nblocks) return;
  int j, kk, g = 0;
  if (a.backward) { j = blk / a.nk; kk = blk % a.nk; }
  else { g = blk % a.G; int64_t t = blk / a.G; kk = t % a.nk; j = t / a.nk; }
  const int c = lane & 31, half = lane >> 5;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = 0.f;
    if (e < per) {
      const int k = kk * ks + half * per + e;
      if (!a.backward) {
        const int unit = 32 * j + c;
        if (unit < a.H && k < a.H) v[e] = a.w[((int64_t)g * a.H + unit) * a.H + k];
      } else {
        const int gg = k / a.width, u = k % a.width, uin = 32 * j + c;
        if (gg < a.G && u < a.H && uin < a.H) v[e] = a.w[((int64_t)gg * a.H + u) * a.H + uin];
      }
    }
  }
  uint4 o;
  if (a.bf16) {
    o.x = pk2(v[0], v[1]); o.y = pk2(v[2], v[3]); o.z = pk2(v[4], v[5]); o.w = pk2(v[6], v[7]);
  } else {
    o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
  }
  a.out[gid] = o;
}
__global__ void pack_kernel(PackArgs a) { pack_elem(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }

// Forward image for the 16-row form (v_mfma_f32_16x16x32_bf16): blocks [j][kk][g][nb] of 64 uint4, lane l holds
// B(k = 32 kk + 8 (l / 16) + e, n = unit 32 j + 16 nb + l % 16 of gate g), e = 0..7; zeros beyond H either way.
struct Pack16Args {
  const float* w;   // [G*H, H]
  uint4* out;
  int H, G, NJ, nk32;
};
__device__ __forceinline__ void pack16_elem(const Pack16Args& a, int64_t gid) {
  const int lane = gid & 63;
  int64_t blk = gid >> 6;
  if (blk >= (int64_t)a.NJ * a.nk32 * a.G * 2) return;
  const int nb = (int)(blk & 1); blk >>= 1;
  const int g = (int)(blk % a.G); blk /= a.G;
  const int kk = (int)(blk % a.nk32), j = (int)(blk / a.nk32);
  const int unit = 32 * j + 16 * nb + (lane & 15);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * kk + 8 * (lane >> 4) + e;
    v[e] = (unit < a.H && k < a.H) ? a.w[((int64_t)g * a.H + unit) * a.H + k] : 0.f;
  }
  uint4 o;
  o.x = pk2(v[0], v[1]); o.y = pk2(v[2], v[3]); o.z = pk2(v[4], v[5]); o.w = pk2(v[6], v[7]);
  a.out[gid] = o;
}
__global__ void pack16_kernel(Pack16Args a) { pack16_elem(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }

// Forward image for the cluster form (H a multiple of 64): blocks [mw][kk] of 64 uint4, mw = 0 .. H/8 - 1 the wave
// (8 per workgroup) that owns units 8 mw .. 8 mw + 7.  W is the A operand of v_mfma_f32_32x32x16_bf16 there: lane l
// holds row m = l % 32 of the wave's 32 gate rows, columns k = 16 kk + 8 (l / 32) + e, e = 0..7; row m is slot m % 4 of
// unit 8 mw + 4 ((m / 4) % 2) + m / 8, so that a lane of the result (row = (reg % 4) + 8 (reg / 4) + 4 (lane / 32)) holds
// the four slots of four neighbouring units.  Slot 3 of a GRU is zeros.
__device__ __forceinline__ void pack_cluster_elem(const float* w, uint4* out, int H, int G, int64_t gid) {
  const int lane = gid & 63;
  const int64_t blk = gid >> 6;
  const int ks = H / 16;
  if (blk >= (int64_t)(H / 8) * ks) return;
  const int kk = (int)(blk % ks), mw = (int)(blk / ks);
  const int m = lane & 31, slot = m & 3, unit = 8 * mw + 4 * ((m >> 2) & 1) + (m >> 3);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * kk + 8 * (lane >> 5) + e;
    v[e] = slot < G ? w[((int64_t)slot * H + unit) * H + k] : 0.f;
  }
  uint4 o;
  o.x = pk2(v[0], v[1]); o.y = pk2(v[2], v[3]); o.z = pk2(v[4], v[5]); o.w = pk2(v[6], v[7]);
  out[gid] = o;
}
struct PackClArgs { const float* w; uint4* out; int H, G; };
__global__ void pack_cluster_kernel(PackClArgs a) { pack_cluster_elem(a.w, a.out, a.H, a.G, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }

// Every image of up to 16 recurrent weight matrices of one shape in ONE launch (blockIdx.y = matrix; the threads of a
// matrix cover its forward image, then its backward image, then its 16-row-form image): a model's images are rebuilt
// after every optimizer step, and two dozen launches of a few microseconds each were 2 % of a training step.
constexpr int PACK_MAXJOBS = 16;
struct PackMultiArgs {
  const float* w[PACK_MAXJOBS];
  uint4* fwd[PACK_MAXJOBS];
  uint4* bwd[PACK_MAXJOBS];
  int H, G, NJ, nkf, nkb, nk32, width, bf16;
  int64_t nf, nb, n16, ncl;      // threads of the images: 32-row forward, backward, 16-row forward, cluster forward
};
__global__ void pack_multi_kernel(PackMultiArgs a) {
  const int job = blockIdx.y;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < a.nf + a.nb + a.n16 + a.ncl; gid += (int64_t)gridDim.x * blockDim.x) {
    if (gid < a.nf) {
      if (a.fwd[job]) pack_elem(PackArgs{a.w[job], a.fwd[job], a.H, a.G, a.NJ, a.nkf, a.width, a.bf16, 0}, gid);
    } else if (gid < a.nf + a.nb) {
      if (a.bwd[job]) pack_elem(PackArgs{a.w[job], a.bwd[job], a.H, a.G, a.NJ, a.nkb, a.width, a.bf16, 1}, gid - a.nf);
    } else if (gid < a.nf + a.nb + a.n16) {
      if (a.fwd[job]) pack16_elem(Pack16Args{a.w[job], a.fwd[job] + a.nf, a.H, a.G, a.NJ, a.nk32}, gid - a.nf - a.nb);
    } else if (a.fwd[job]) {
      pack_cluster_elem(a.w[job], a.fwd[job] + a.nf + a.n16, a.H, a.G, gid - a.nf - a.nb - a.n16);
    }
  }
}

// ---------------------------------------------------------------------------
// acc[n] += A(32 x K from LDS) * B(block stream): blocks ordered [kk][n], 64 uint4 each.  The first kl k-blocks
// are read from a copy of theirs in LDS (wl, same order), the rest from global memory through a ring of PF
// k-blocks in registers; the ring's first loads are issued before the LDS part, which hides their latency.
// (Routing the LDS blocks through the ring as well was measured slower: the forward LSTM then spills.)
// The same product with nothing streamed: the first kl k-blocks of W come from their copy in LDS (wl), the last NKR
// from registers (wres, loaded once before the time loop).  The A tile holds only `arows` rows (the workgroup's real
// batch rows); the lanes of the other rows of the 32-row MFMA operand supply zeros.  bf16 only.
template <int NT, int NKR>
__device__ __forceinline__ void mma_resident(f32x16* acc, const unsigned char* lds_a, int lds_ld, int arows, const uint4* wl, int kl,
                                             const uint4 (&wres)[NKR][NT], int lane) {
  const int row = lane & 31, half = lane >> 5;
  const bool live = row < arows;
  const unsigned char* abase = lds_a + ((size_t)(live ? row : 0) * lds_ld + 8 * half) * 2;
  auto afrag = [&](int kk) -> bf16x8 {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(abase + 32 * kk);
    if (!live) v = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    return v;
  };
  for (int kk = 0; kk < kl; ++kk) {
    const bf16x8 av = afrag(kk);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const bf16x8 bv = __builtin_bit_cast(bf16x8, wl[(kk * NT + n) * 64 + lane]);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
    }
  }
  // A fragments four k-blocks at a time: their LDS latency is paid once per group, at 16 registers
#pragma unroll
  for (int p0 = 0; p0 < NKR; p0 += 4) {
    bf16x8 av[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p0 + p < NKR) av[p] = afrag(kl + p0 + p);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p0 + p < NKR) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p], __builtin_bit_cast(bf16x8, wres[p0 + p][n]), acc[n], 0, 0, 0);
      }
  }
}

template <bool BF16, int NT, int PF>
__device__ __forceinline__ void mma_stream(f32x16* acc, const unsigned char* lds_a, int lds_ld,
                                           const uint4* __restrict__ wp, int nk, int lane, const uint4* wl = nullptr, int kl = 0) {
  const int row = lane & 31, half = lane >> 5;
  typedef typename std::conditional<BF16, bf16x8, f32x4>::type afrag_t;
  auto afrag = [&](int kk) -> afrag_t {
    if constexpr (BF16) return *reinterpret_cast<const bf16x8*>(lds_a + ((size_t)row * lds_ld + 16 * kk + 8 * half) * 2);
    else return *reinterpret_cast<const f32x4*>(lds_a + ((size_t)row * lds_ld + 8 * kk + 4 * half) * 4);
  };
  auto mma = [&](afrag_t av, const uint4* bsrc) {
    if constexpr (BF16) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 bv = __builtin_bit_cast(bf16x8, bsrc[n]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f32x4 bv = __builtin_bit_cast(f32x4, bsrc[n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[n], 0, 0, 0);
      }
    }
  };
  uint4 ring[PF][NT];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (kl + p < nk) {
#pragma unroll
      for (int n = 0; n < NT; ++n) ring[p][n] = wp[((int64_t)(kl + p) * NT + n) * 64 + lane];
    }
  for (int kk = 0; kk < kl; ++kk) {
    const afrag_t av = afrag(kk);
    uint4 b[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = wl[(kk * NT + n) * 64 + lane];
    mma(av, b);
  }
  // the A fragments of a whole ring round are fetched together, so their LDS latency is paid once per
  // round instead of once per k-block (indices beyond nk re-read block nk - 1, unused)
  for (int kk0 = kl; kk0 < nk; kk0 += PF) {
    afrag_t av[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) av[p] = afrag(kk0 + p < nk ? kk0 + p : nk - 1);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kk = kk0 + p;
      if (kk < nk) {
        mma(av[p], ring[p]);
        if (kk + PF < nk) {
#pragma unroll
          for (int n = 0; n < NT; ++n) ring[p][n] = wp[((int64_t)(kk + PF) * NT + n) * 64 + lane];
        }
      }
    }
  }
}

// One chunk's product of the backward pass (bf16, one accumulator) with most of the transposed W_hh image resident:
// k-blocks [0, kl) come from the wave's copy in LDS (wl), the next NKR from registers (wres, loaded once before the
// time loop), only the rest is streamed from L2 through the ring, whose first loads are issued before everything else.
template <int PF, int NKR>
__device__ __forceinline__ void mma_mixed(f32x16& acc, const unsigned char* lds_a, int lds_ld, const uint4* __restrict__ wp, int nk, int lane,
                                          const uint4* wl, int kl, const uint4 (&wres)[NKR > 0 ? NKR : 1]) {
  const int row = lane & 31, half = lane >> 5;
  const unsigned char* ab = lds_a + ((size_t)row * lds_ld + 8 * half) * 2;
  auto afrag = [&](int kk) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(ab + 32 * kk); };
  const int ks0 = kl + NKR;
  uint4 ring[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (ks0 + p < nk) ring[p] = wp[(int64_t)(ks0 + p) * 64 + lane];
#pragma unroll 4
  for (int kk = 0; kk < kl; ++kk) {
    const bf16x8 av = afrag(kk);
    const bf16x8 bv = __builtin_bit_cast(bf16x8, wl[kk * 64 + lane]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
  if constexpr (NKR > 0) {
#pragma unroll
    for (int p0 = 0; p0 < NKR; p0 += 4) {
      bf16x8 av[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (p0 + p < NKR) av[p] = afrag(kl + p0 + p);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (p0 + p < NKR) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p], __builtin_bit_cast(bf16x8, wres[p0 + p]), acc, 0, 0, 0);
    }
  }
  for (int kk0 = ks0; kk0 < nk; kk0 += PF) {
    bf16x8 av[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) av[p] = afrag(kk0 + p < nk ? kk0 + p : nk - 1);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kk = kk0 + p;
      if (kk < nk) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p], __builtin_bit_cast(bf16x8, ring[p]), acc, 0, 0, 0);
        if (kk + PF < nk) ring[p] = wp[(int64_t)(kk + PF) * 64 + lane];
      }
    }
  }
}

template <bool BF16>
__device__ __forceinline__ void lds_put(unsigned char* base, int lds_ld, int row, int col, float v) {
  if constexpr (BF16) reinterpret_cast<bf16_t*>(base)[(size_t)row * lds_ld + col] = (bf16_t)v;
  else reinterpret_cast<float*>(base)[(size_t)row * lds_ld + col] = v;
}

// ---------------------------------------------------------------------------
// Gate-interleaved tensors (xproj, gate stash, dgates): the 4 slots of hidden unit u of a direction sit
// at columns 4u .. 4u+3, as fp32 (16 bytes) or bf16 (8 bytes): one access per lane and batch row.
template <bool ST16> struct Slots;
template <> struct Slots<false> {
  typedef float4 raw;
  static constexpr int ESZ = 4;
  static __device__ __forceinline__ raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ f32x4 unpack(raw v) { return f32x4{v.x, v.y, v.z, v.w}; }
  static __device__ __forceinline__ raw pack(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
  static __device__ __forceinline__ float one(const void* base, int64_t idx) { return reinterpret_cast<const float*>(base)[idx]; }
};
template <> struct Slots<true> {
  typedef uint2 raw;
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ raw zero() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ f32x4 unpack(raw v) {
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ raw pack(float a, float b, float c, float d) { return make_uint2(pk2(a, b), pk2(c, d)); }
  static __device__ __forceinline__ float one(const void* base, int64_t idx) {
    return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16);
  }
};
// value a bf16 store / LDS operand will hold (what the backward's recurrent product multiplies)
__device__ __forceinline__ float bf16_rt(float v) {
  bf16_t b = (bf16_t)v;
  return __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

// ---------------------------------------------------------------------------
// (batch tile, direction) of a workgroup of a 1-D grid.  With two directions and xcd_dirs set, the workgroups that share
// an XCD (blocks b and b + 8 do under the round-robin dispatch that is observed: a placement for speed, nothing depends
// on it) all work on ONE direction, so that an XCD's L2 holds one direction's W_hh image instead of two: at H = 512 the
// two images (2 x 2 MB for the LSTM) are the whole L2 and the per-step stream of W_hh came from beyond it.
__device__ __forceinline__ bool rnn_block(int ntiles, int ndir, int xcd_dirs, int& tile, int& dir) {
  const int bid = blockIdx.x;
  if (ndir == 2 && xcd_dirs) { const int x = bid & 7; dir = x & 1; tile = (bid >> 3) * 4 + (x >> 1); }
  else { dir = bid / ntiles; tile = bid - dir * ntiles; }
  return tile < ntiles;
}
inline unsigned rnn_grid(int64_t ntiles, int ndir, int xcd_dirs) {
  return (unsigned)((ndir == 2 && xcd_dirs) ? cdiv64(ntiles, 4) * 8 : ntiles * ndir);
}
// Measured: H = 512 (two images of 1.5-2 MB) 1-4 % faster with one direction per XCD, H = 200 (2 x 320 KB) 0-4 % slower: the
// stream of W_hh is bound by what a CU's vector memory path takes in (44 GB/s per CU at H = 512), not by where in the
// cache hierarchy it comes from.  On when the two images together exceed 3 MB.
inline int rnn_xcd_dirs(int64_t H, int G) {
  static const char* e = getenv("TAD_RNN_XCD");      // experiments: 0 = tiles of direction 0, then of direction 1; 1 = always
  return e ? atoi(e) : (2 * G * H * H * 2 > 3 * 1024 * 1024);
}

struct RnnFwdArgs {
  tad_rnn_dir_t d[MAXDIR];
  int ntiles, ndir, xcd_dirs;
  int B, T, H, NJ, nk, lds_ld, rpw;
  int kl;      // leading k-blocks of every chunk's W_hh image kept in LDS for the whole sequence
  int out16_rows;   // 1: the bf16 copy of h_t is written from the LDS tile in 16-byte pieces (bf16 precision, aligned rows)
  int tile_rows;    // rows of an h tile in LDS: 32, or rpw in the resident form
  int nk32;         // 16-row form: 32-deep k-steps of its W_hh image
};

// NKR > 0, the resident form (small batches: rpw <= 16, one chunk per wave, bf16): nothing of W_hh is streamed inside
// the time loop -- k-blocks [0, kl) sit in LDS, the last NKR = nk - kl in registers -- and the h tiles hold only the
// rpw real rows, which is what makes room for kl blocks in LDS.  RL = rpw / 2 (the live accumulator registers) is then
// a compile-time constant, so that the dead rows cost no registers.  The step of such a launch is a dependent chain
// (h tile -> product -> gate math -> barrier) in which waiting for W_hh blocks from L2 was the longest link.
template <int CELL, bool BF16, bool ST16, int JPW, int NKR = 0, int RL = 0>
__global__ __launch_bounds__(512) void rnn_fwd_kernel(RnnFwdArgs a) {
  static_assert(NKR == 0 || (BF16 && JPW == 1 && RL > 0), "resident form: bf16, one chunk per wave, fixed row count");
  constexpr int G = CELL == TAD_CELL_LSTM ? 4 : 3;
  constexpr int ESZ = BF16 ? 2 : 4;
  constexpr int PF = BF16 ? 4 : 2;
  typedef Slots<ST16> SL;
  typedef typename SL::raw raw_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int tile_, dir_;
  if (!rnn_block(a.ntiles, a.ndir, a.xcd_dirs, tile_, dir_)) return;
  const tad_rnn_dir_t d = a.d[dir_];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = lane & 31;
  const int64_t b0 = (int64_t)tile_ * a.rpw;   // rows >= rpw of the 32-row MFMA tile stay empty
  const size_t buf_bytes = (size_t)a.tile_rows * a.lds_ld * ESZ;
  const int H = a.H, T = a.T;
  const int rlive = RL > 0 ? RL : (a.rpw >> 1);     // accumulator registers that hold real rows (rpw is 8, 16 or 32)
  const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed);

  for (size_t i = threadIdx.x; i < 2 * buf_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  // W_hh is the same every step: the part of the wave's chunk image that fits the LDS beside the two h
  // buffers is copied there once; only the rest is streamed from L2 each step (JPW == 1 only)
  uint4* wl = reinterpret_cast<uint4*>(smem + 2 * buf_bytes) + (size_t)wave * a.kl * G * 64;
  if (JPW == 1 && wave < a.NJ) {
    for (int i = 0; i < a.kl * G; ++i) wl[i * 64 + lane] = wp[((int64_t)wave * a.nk * G + i) * 64 + lane];
  }
  uint4 wres[NKR > 0 ? NKR : 1][G];
  if constexpr (NKR > 0) {
    if (wave < a.NJ) {
#pragma unroll
      for (int p = 0; p < NKR; ++p)
#pragma unroll
        for (int n = 0; n < G; ++n) wres[p][n] = wp[(((int64_t)wave * a.nk + a.kl + p) * G + n) * 64 + lane];
    }
  }

  float st[JPW][16];   // LSTM: c; GRU: h
  float bh[JPW][3];
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[jj][r] = 0.f;
    bh[jj][0] = bh[jj][1] = bh[jj][2] = 0.f;
    if constexpr (CELL == TAD_CELL_GRU) {
      const int u = 32 * (wave + jj * nw) + col;
      if (u < H && d.b_hh) { bh[jj][0] = d.b_hh[u]; bh[jj][1] = d.b_hh[H + u]; bh[jj][2] = d.b_hh[2 * H + u]; }
    }
  }
  __syncthreads();

  // Addressing: row r of the tile is batch b0 + cr + 4*half with cr = (r&3) + 8*(r>>2) a
  // compile-time constant, so every access is (wave-uniform row base) + (one 32-bit per-lane
  // byte offset shared by all 16 rows); lanes 0-31 / 32-63 each cover one contiguous run of a row.
  const int half = lane >> 5;
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(d.xproj);
  // input-side pre-activations (all slots of the lane's unit) of one step for chunk slot jj
  auto load_xp = [&](int step_, int jj, raw_t* raw) {
    const int tt_ = d.reverse ? T - 1 - step_ : step_;
    const int u = 32 * (wave + jj * nw) + col;
    const bool uok = u < H;
    const uint32_t voff = (uint32_t)(((4 * half * T + tt_) * (int)d.ld_xproj + 4 * u) * SL::ESZ);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      if (r >= rlive) { raw[r] = SL::zero(); continue; }      // no batch row in this register: nothing to fetch
      const bool ok = uok && (cr + 4 * half) < a.rpw && (b0 + cr + 4 * half) < a.B;
      const unsigned char* xp = xbase + (b0 + cr) * (int64_t)T * d.ld_xproj * SL::ESZ;
      raw[r] = ok ? *reinterpret_cast<const raw_t*>(xp + voff) : SL::zero();
    }
  };
  // accumulator seeds: LSTM all four slots; GRU r, z (b_hh is added in the epilogue) and the n slot kept aside
  auto seed = [&](const raw_t* raw, f32x16* acc, float* xn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x4 v = SL::unpack(raw[r]);
      if constexpr (CELL == TAD_CELL_LSTM) {
        acc[0][r] = v[0]; acc[1][r] = v[1]; acc[2][r] = v[2]; acc[3][r] = v[3];
      } else {
        acc[0][r] = v[0]; acc[1][r] = v[1]; acc[2][r] = 0.f;
        xn[r] = v[2];
      }
    }
  };

  // With one chunk per wave the next step's xproj is fetched right after this step's MFMAs,
  // so its HBM latency hides under the gate math, the stores and the barrier.
  constexpr bool PREFETCH = (JPW == 1);
  raw_t nraw[16];
  if constexpr (PREFETCH) {
    if (wave < a.NJ) load_xp(0, 0, nraw);
  }

#ifdef EXP_STAMPS
  unsigned long long sSeed = 0, sMma = 0, sPre = 0, sGate = 0, sBar = 0;
#endif
  for (int step = 0; step < T; ++step) {
    const int tt = d.reverse ? T - 1 - step : step;
    const unsigned char* cur = smem + (size_t)(step & 1) * buf_bytes;
    unsigned char* nxt = smem + (size_t)((step + 1) & 1) * buf_bytes;
    RSTAMP(q0);
#ifdef EXP_STAMPS
    unsigned long long q1 = q0, q2 = q0, q3 = q0;
#endif
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int j = wave + jj * nw;
      if (j < a.NJ) {
        const int u = 32 * j + col;
        const bool uok = u < H;
        f32x16 acc[G];
        float xn[16];
        if constexpr (PREFETCH) {
          seed(nraw, acc, xn);
        } else {
          raw_t raw[16];
          load_xp(step, jj, raw);
          seed(raw, acc, xn);
        }
#ifdef EXP_STAMPS
        { RSTAMP(qq); q1 = qq; }
#endif
        if constexpr (NKR > 0) mma_resident<G, NKR>(acc, cur, a.lds_ld, a.tile_rows, wl, a.kl, wres, lane);
        else mma_stream<BF16, G, PF>(acc, cur, a.lds_ld, wp + (int64_t)j * a.nk * G * 64, a.nk, lane, wl, JPW == 1 ? a.kl : 0);
#ifdef EXP_STAMPS
        { RSTAMP(qq); q2 = qq; }
#endif
        if constexpr (PREFETCH) {
          if (step + 1 < T) load_xp(step + 1, 0, nraw);
        }
#ifdef EXP_STAMPS
        { RSTAMP(qq); q3 = qq; }
#endif
        const uint32_t ooff = (uint32_t)((4 * half * T + tt) * (int)d.ld_out + u);
        const uint32_t o16off = (uint32_t)((4 * half * T + tt) * (int)d.ld_out16 + u);
        const uint32_t goff = (uint32_t)(((4 * half * T + tt) * (int)d.ld_gates + 4 * u) * SL::ESZ);
        const uint32_t coff = (uint32_t)((4 * half * T + tt) * (int)d.ld_c + u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // accumulator register r holds tile rows cr and cr + 4: with fewer than 32 real rows per workgroup
          // (small batches) registers r >= rpw / 2 hold no row of the batch and skip the gate math -- the
          // step is bound by it (10 quarter-rate exp / rcp per element), not by the matrix product.
          // Their rows of the LDS operand stay the zeros both buffers start with.
          if (r >= rlive) continue;
          const int cr = (r & 3) + 8 * (r >> 2);
          const int row = cr + 4 * half;
          const bool ok = uok && row < a.rpw && (b0 + row) < a.B;
          float h, cst = 0.f;
          raw_t sv;
          if constexpr (CELL == TAD_CELL_LSTM) {
            const float ig = fast_sigmoid(acc[0][r]), fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]), og = fast_sigmoid(acc[3][r]);
            const float c = fg * st[jj][r] + ig * gg;
            st[jj][r] = c;
            h = og * fast_tanh(c);
            cst = c;
            sv = SL::pack(ig, fg, gg, og);
          } else {
            const float rg = fast_sigmoid(acc[0][r] + bh[jj][0]), zg = fast_sigmoid(acc[1][r] + bh[jj][1]);
            const float hpn = acc[2][r] + bh[jj][2];
            const float ng = fast_tanh(xn[r] + rg * hpn);
            h = (1.f - zg) * ng + zg * st[jj][r];
            st[jj][r] = h;
            sv = SL::pack(rg, zg, ng, hpn);
          }
          if (!ok) h = 0.f;
          lds_put<BF16>(nxt, a.lds_ld, row, u, h);
          if (ok) {
            if (d.out) (d.out + (b0 + cr) * (int64_t)T * d.ld_out)[ooff] = h;
            if (d.out16 && !(BF16 && a.out16_rows)) (reinterpret_cast<bf16_t*>(d.out16) + (b0 + cr) * (int64_t)T * d.ld_out16)[o16off] = (bf16_t)h;
            if (d.gates) {
              unsigned char* gp = reinterpret_cast<unsigned char*>(d.gates) + (b0 + cr) * (int64_t)T * d.ld_gates * SL::ESZ;
              *reinterpret_cast<raw_t*>(gp + goff) = sv;
              if constexpr (CELL == TAD_CELL_LSTM) (d.cstate + (b0 + cr) * (int64_t)T * d.ld_c)[coff] = cst;
            }
          }
        }
      }
    }
    RSTAMP(q4);
    __syncthreads();
    RSTAMP(q5);
    if constexpr (BF16) {
      // h_t sits in the LDS tile as bf16 (the next step's matrix operand): its copy for the products that read h
      // leaves from there, 16 bytes per lane along a row, instead of 2 bytes per lane from the gate math
      if (a.out16_rows && d.out16) {
        const int cpr = H >> 3;                                    // 16-byte pieces per row
        const int total = a.rpw * cpr;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
          const int row = i / cpr, ch = i - row * cpr;
          if (b0 + row < a.B) {
            const uint4 v = *reinterpret_cast<const uint4*>(nxt + ((size_t)row * a.lds_ld + 8 * ch) * 2);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.out16) + ((b0 + row) * (int64_t)T + tt) * d.ld_out16 + 8 * ch) = v;
          }
        }
      }
    }
#ifdef EXP_STAMPS
    sSeed += q1 - q0; sMma += q2 - q1; sPre += q3 - q2; sGate += q4 - q3; sBar += q5 - q4;
#endif
  }
#ifdef EXP_STAMPS
  if (threadIdx.x == 0) {
    atomicAdd(&g_rnn_stamps[0], sSeed); atomicAdd(&g_rnn_stamps[1], sMma); atomicAdd(&g_rnn_stamps[2], sPre);
    atomicAdd(&g_rnn_stamps[3], sGate); atomicAdd(&g_rnn_stamps[4], sBar); atomicAdd(&g_rnn_stamps[7], (unsigned long long)T);
  }
#endif
}

// ---------------------------------------------------------------------------
// The 16-row form of the forward kernel, for launches of 8 batch rows per workgroup without a gate stash (every launch
// of the sampler at up to 2047 sequences per call; bf16 precision and storage, one chunk of 32 units per wave).  In the
// 32-row form three quarters of every MFMA there multiply padding and the matrix pipe, shared by the two waves of a
// SIMD, is the longest phase of the step; v_mfma_f32_16x16x32_bf16 halves that.  The 8 real rows sit at rows
// {0,1, 4,5, 8,9, 12,13} of the 16-row operand, so that in the result (row = 4 * (lane / 16) + register) every lane
// keeps two live registers per 16-unit block: the gate math runs on all 64 lanes, four elements each, as before.
// W_hh is resident: k-steps (32 deep) [0, kl) in LDS, the last NKR in registers (4 VGPRs per k-step, gate and block).
// RPW = 16: all 16 rows of the operand are batch rows (four live registers per block); STASH: the launch also writes the
// gate stash (and the LSTM's cell state) a backward pass reads.
template <int CELL, int NKR, int RPW = 8, bool STASH = false>
__global__ __launch_bounds__(512) void rnn_fwd16_kernel(RnnFwdArgs a) {
  static_assert(RPW == 8 || RPW == 16, "8 or 16 batch rows per workgroup");
  constexpr int LIVE = RPW == 16 ? 4 : 2;      // live result registers per 16-unit block
  constexpr int G = CELL == TAD_CELL_LSTM ? 4 : 3;
  typedef Slots<true> SL;
  typedef typename SL::raw raw_t;
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int tile_, dir_;
  if (!rnn_block(a.ntiles, a.ndir, a.xcd_dirs, tile_, dir_)) return;
  const tad_rnn_dir_t d = a.d[dir_];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int64_t b0 = (int64_t)tile_ * RPW;
  const size_t buf_bytes = (size_t)RPW * a.lds_ld * 2;
  const int H = a.H, T = a.T;
  // the 16-row-form image follows the 32-row-form image in the packed buffer
  const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed) + (int64_t)a.NJ * a.nk * G * 64 + (int64_t)wave * a.nk32 * G * 2 * 64;

  for (size_t i = threadIdx.x; i < 2 * buf_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  uint4* wl = reinterpret_cast<uint4*>(smem + 2 * buf_bytes) + (size_t)wave * a.kl * G * 2 * 64;
  uint4 wres[NKR][G][2];
  if (wave < a.NJ) {
    for (int i = 0; i < a.kl * G * 2; ++i) wl[i * 64 + lane] = wp[i * 64 + lane];
#pragma unroll
    for (int p = 0; p < NKR; ++p)
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wres[p][g][nb] = wp[(((a.kl + p) * G + g) * 2 + nb) * 64 + lane];
  }
  float st[2][LIVE], bh[2][3];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int e = 0; e < LIVE; ++e) st[nb][e] = 0.f;
    bh[nb][0] = bh[nb][1] = bh[nb][2] = 0.f;
    if constexpr (CELL == TAD_CELL_GRU) {
      const int u = 32 * wave + 16 * nb + c16;
      if (u < H && d.b_hh) { bh[nb][0] = d.b_hh[u]; bh[nb][1] = d.b_hh[H + u]; bh[nb][2] = d.b_hh[2 * H + u]; }
    }
  }
  __syncthreads();

  // A operand: lane -> operand row lane % 16, real when (row & 3) < 2: batch row 2 * (row / 4) + (row & 1) of the tile
  const int arow = lane & 15;
  const bool a_live = RPW == 16 || (arow & 3) < 2;
  const uint32_t a_off = (uint32_t)(((RPW == 16 ? arow : (a_live ? 2 * (arow >> 2) + (arow & 1) : 0)) * a.lds_ld + 8 * q) * 2);
  // result / gate math: element (nb, e) of the lane = batch row 2 q + e, unit 32 wave + 16 nb + lane % 16
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(d.xproj);
  auto load_xp = [&](int step_, raw_t (&raw)[2][LIVE]) {
    const int tt_ = d.reverse ? T - 1 - step_ : step_;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int u = 32 * wave + 16 * nb + c16;
#pragma unroll
      for (int e = 0; e < LIVE; ++e) {
        const int64_t b = b0 + LIVE * q + e;
        const bool ok = u < H && b < a.B;
        raw[nb][e] = ok ? *reinterpret_cast<const raw_t*>(xbase + ((b * T + tt_) * d.ld_xproj + 4 * u) * 2) : SL::zero();
      }
    }
  };
  raw_t nraw[2][LIVE];
  if (wave < a.NJ) load_xp(0, nraw);

  for (int step = 0; step < T; ++step) {
    const int tt = d.reverse ? T - 1 - step : step;
    const unsigned char* cur = smem + (size_t)(step & 1) * buf_bytes;
    unsigned char* nxt = smem + (size_t)((step + 1) & 1) * buf_bytes;
    if (wave < a.NJ) {
      f32x4v acc[G][2];
      float xn[2][LIVE];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (e < LIVE) {
            const f32x4 v = SL::unpack(nraw[nb][e]);
            if constexpr (CELL == TAD_CELL_LSTM) {
              acc[0][nb][e] = v[0]; acc[1][nb][e] = v[1]; acc[2][nb][e] = v[2]; acc[3][nb][e] = v[3];
            } else {
              acc[0][nb][e] = v[0]; acc[1][nb][e] = v[1]; acc[2][nb][e] = 0.f; xn[nb][e] = v[2];
            }
          } else {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][nb][e] = 0.f;
          }
        }
      auto afrag = [&](int kk) -> bf16x8 {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(cur + a_off + 64 * kk);
        if (!a_live) v = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
        return v;
      };
      for (int kk = 0; kk < a.kl; ++kk) {
        const bf16x8 av = afrag(kk);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wl[((kk * G + g) * 2 + nb) * 64 + lane]), acc[g][nb], 0, 0, 0);
      }
      {
        bf16x8 av[NKR];
#pragma unroll
        for (int p = 0; p < NKR; ++p) av[p] = afrag(a.kl + p);
#pragma unroll
        for (int p = 0; p < NKR; ++p)
#pragma unroll
          for (int g = 0; g < G; ++g)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[p], __builtin_bit_cast(bf16x8, wres[p][g][nb]), acc[g][nb], 0, 0, 0);
      }
      if (step + 1 < T) load_xp(step + 1, nraw);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int u = 32 * wave + 16 * nb + c16;
#pragma unroll
        for (int e = 0; e < LIVE; ++e) {
          const int row = LIVE * q + e;
          const int64_t b = b0 + row;
          const bool ok = u < H && b < a.B;
          float h;
          raw_t sv;
          float cst = 0.f;
          if constexpr (CELL == TAD_CELL_LSTM) {
            const float ig = fast_sigmoid(acc[0][nb][e]), fg = fast_sigmoid(acc[1][nb][e]);
            const float gg = fast_tanh(acc[2][nb][e]), og = fast_sigmoid(acc[3][nb][e]);
            const float c = fg * st[nb][e] + ig * gg;
            st[nb][e] = c;
            h = og * fast_tanh(c);
            cst = c;
            sv = SL::pack(ig, fg, gg, og);
          } else {
            const float rg = fast_sigmoid(acc[0][nb][e] + bh[nb][0]), zg = fast_sigmoid(acc[1][nb][e] + bh[nb][1]);
            const float hpn = acc[2][nb][e] + bh[nb][2];
            const float ng = fast_tanh(xn[nb][e] + rg * hpn);
            h = (1.f - zg) * ng + zg * st[nb][e];
            st[nb][e] = h;
            sv = SL::pack(rg, zg, ng, hpn);
          }
          if (!ok) h = 0.f;
          if (u < a.lds_ld) lds_put<true>(nxt, a.lds_ld, row, u, h);
          if (ok) {
            if (d.out) d.out[(b * T + tt) * d.ld_out + u] = h;
            if (d.out16 && !a.out16_rows) reinterpret_cast<bf16_t*>(d.out16)[(b * T + tt) * d.ld_out16 + u] = (bf16_t)h;
            if constexpr (STASH) {
              *reinterpret_cast<raw_t*>(reinterpret_cast<unsigned char*>(d.gates) + ((b * T + tt) * d.ld_gates + 4 * u) * 2) = sv;
              if constexpr (CELL == TAD_CELL_LSTM) d.cstate[(b * T + tt) * d.ld_c + u] = cst;
            }
          }
        }
      }
    }
    __syncthreads();
    if (a.out16_rows && d.out16) {
      const int cpr = H >> 3;
      const int total = RPW * cpr;
      for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int row = i / cpr, ch = i - row * cpr;
        if (b0 + row < a.B) {
          const uint4 v = *reinterpret_cast<const uint4*>(nxt + ((size_t)row * a.lds_ld + 8 * ch) * 2);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.out16) + ((b0 + row) * (int64_t)T + tt) * d.ld_out16 + 8 * ch) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
struct RnnBwdArgs {
  tad_rnn_bwd_dir_t d[MAXDIR];
  int ntiles, ndir, xcd_dirs;
  int B, T, H, NJ, width, lds_ld, nk_total, nk_gate, rpw;
  int kl;      // leading k-blocks of every chunk's transposed W_hh image kept in LDS for the whole sequence (one-pass bf16 form)
};

// NKR: k-blocks [kl, kl + NKR) of the wave's image stay in registers for the whole sequence (one-pass bf16 form with one
// chunk per wave).  Streaming all 52 (LSTM, H = 200) from L2 on every step was what bounded the step: the vector L1 of a
// CU fills at 10-20 bytes per clock, and 364 KB per workgroup and step went through it.
template <int CELL, bool BF16, bool ST16, int JPW, bool PHASED, int NKR = 0>
__global__ __launch_bounds__(512) void rnn_bwd_kernel(RnnBwdArgs a) {
  static_assert(NKR == 0 || (BF16 && JPW == 1 && !PHASED), "register-resident k-blocks: one-pass bf16 form, one chunk per wave");
  constexpr int G = CELL == TAD_CELL_LSTM ? 4 : 3;
  // k-blocks of the transposed W_hh image a wave keeps in flight from L2 (4 VGPRs each).  With two chunks per wave (H = 512:
  // everything is streamed, 1.5-2 MB per step) six measured 4-5 % faster than four (two: 28 % slower; eight: as six, with
  // spills); with one chunk per wave (H = 200: most of the image is resident) six cost the GRU 10 %.
  constexpr int PF = BF16 ? (JPW > 1 ? 6 : 4) : 2;
  typedef Slots<ST16> SL;
  typedef typename SL::raw raw_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int tile_, dir_;
  if (!rnn_block(a.ntiles, a.ndir, a.xcd_dirs, tile_, dir_)) return;
  const tad_rnn_bwd_dir_t d = a.d[dir_];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = lane & 31;
  const int64_t b0 = (int64_t)tile_ * a.rpw;   // rows >= rpw of the 32-row MFMA tile stay empty
  const int H = a.H, T = a.T;
  const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed_t);
  const int half = lane >> 5;
  // accumulator registers that hold real rows (see the forward); the rest skip phase 1, their rows of the
  // LDS tile stay the zeros it is cleared to here, so their accumulators stay zero
  const int rlive = a.rpw >> 1;
  if (!PHASED && a.rpw < 32) {
    const size_t words = (size_t)32 * a.lds_ld * (BF16 ? 2 : 4) / 4;
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    __syncthreads();
  }

  constexpr bool MIXED = BF16 && JPW == 1 && !PHASED;
  const uint4* wl = nullptr;
  uint4 wres[NKR > 0 ? NKR : 1];
  if constexpr (MIXED) {
    // the wave's own part of the LDS behind the gate-gradient tile: written and read by this wave only
    uint4* w = reinterpret_cast<uint4*>(smem + (size_t)32 * a.lds_ld * 2) + (size_t)wave * a.kl * 64;
    if (wave < a.NJ) {
      const uint4* src = wp + (int64_t)wave * a.nk_total * 64;
      for (int i = 0; i < a.kl; ++i) w[i * 64 + lane] = src[i * 64 + lane];
      if constexpr (NKR > 0) {
#pragma unroll
        for (int q = 0; q < NKR; ++q) wres[q] = src[(a.kl + q) * 64 + lane];
      }
    }
    wl = w;
  }

  f32x16 acc[JPW];
  float dcs[JPW][16];      // LSTM: dc carried to the step before; GRU with SPLIT_Z: dh z, the direct part of dh_{t-1}
  // GRU: dh_{t-1} = dh z + (gate gradients) W_hh.  With one chunk per wave the accumulator of the product is seeded with
  // dh z; with two, writing computed values into the two 16-register accumulators element by element sent the register
  // allocator into 16 000 spilled registers, so there the accumulators start at zero and dh z is added when dh is formed.
  constexpr bool SPLIT_Z = CELL == TAD_CELL_GRU && JPW > 1 && !PHASED;
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[jj][r] = 0.f; dcs[jj][r] = 0.f; }

  // Operands of one step for the owned (row, unit) elements: dout, the gate stash, and the states the cell
  // needs (LSTM: c_t and c_{t-1}; GRU: h_{t-1}).  With one chunk per wave they are fetched one step ahead,
  // right before the recurrent product of the step in progress, so their HBM latency hides under it;
  // c_t is then carried over from the previous step's c_{t-1}.  The GRU has the registers for that; the
  // LSTM (one more state and gate per element) would spill, so it fetches at the start of its step.
  // (wave-uniform row bases + one 32-bit per-lane offset per buffer, as in the forward)
  constexpr bool PREFETCH = (JPW == 1) && CELL == TAD_CELL_GRU;
  float q_dout[JPW][16], q_cur[JPW][16], q_prev[JPW][16];
  raw_t q_g[JPW][16];
  auto fetch = [&](int stp, int jj, bool carry) {
    const int n_ = T - 1 - stp;
    const int tt_ = d.reverse ? T - 1 - n_ : n_;
    const int tp_ = (n_ > 0) ? (d.reverse ? tt_ + 1 : tt_ - 1) : -1;
    const int u = 32 * (wave + jj * nw) + col;
    const bool uok = u < H;
    const int rowT = 4 * half * T;
    const uint32_t doff = (uint32_t)((rowT + tt_) * (int)d.ld_dout + u);
    const uint32_t soff = (uint32_t)(((rowT + tt_) * (int)d.ld_gates + 4 * u) * SL::ESZ);
    const uint32_t coff = (uint32_t)((rowT + tt_) * (int)d.ld_c + u);
    const uint32_t cpoff = (uint32_t)((rowT + (tp_ >= 0 ? tp_ : 0)) * (int)d.ld_c + u);
    const uint32_t ooff = (uint32_t)((rowT + (tp_ >= 0 ? tp_ : 0)) * (int)d.ld_out + u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      const bool ok = uok && (cr + 4 * half) < a.rpw && (b0 + cr + 4 * half) < a.B;
      const int64_t rb = (b0 + cr) * (int64_t)T;
      q_dout[jj][r] = ok ? (d.dout + rb * d.ld_dout)[doff] : 0.f;
      q_g[jj][r] = ok ? *reinterpret_cast<const raw_t*>(reinterpret_cast<const unsigned char*>(d.gates) + rb * d.ld_gates * SL::ESZ + soff)
                      : SL::zero();
      if constexpr (CELL == TAD_CELL_LSTM) {
        const float* cp = d.cstate + rb * d.ld_c;
        q_cur[jj][r] = carry ? q_prev[jj][r] : (ok ? cp[coff] : 0.f);
        q_prev[jj][r] = (ok && tp_ >= 0) ? cp[cpoff] : 0.f;
      } else {
        q_prev[jj][r] = (ok && tp_ >= 0) ? (d.out + rb * d.ld_out)[ooff] : 0.f;
      }
    }
  };
  if constexpr (PREFETCH) {
    if (wave < a.NJ) fetch(0, 0, false);
  }

#ifdef EXP_STAMPS
  unsigned long long sFetch = 0, sGate = 0, sBar1 = 0, sMma = 0, sBar2 = 0;
#endif
  for (int step = 0; step < T; ++step) {
    const int n = T - 1 - step;                       // forward step being undone
    const int tt = d.reverse ? T - 1 - n : n;
    RSTAMP(s0);
#ifdef EXP_STAMPS
    unsigned long long s1 = s0;
#endif
    // ---- phase 1: gate gradients for the owned (row, unit) elements
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int j = wave + jj * nw;
      if (j < a.NJ) {
        const int u = 32 * j + col;
        const bool uok = u < H;
        if constexpr (!PREFETCH) fetch(step, jj, false);
#ifdef EXP_STAMPS
        { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RSTAMP(qq); s1 = qq; }
#endif
        const uint32_t goff = (uint32_t)(((4 * half * T + tt) * (int)d.ld_dgates + 4 * u) * SL::ESZ);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r >= rlive) continue;
          const int cr = (r & 3) + 8 * (r >> 2);
          const int row = cr + 4 * half;
          const bool ok = uok && row < a.rpw && (b0 + row) < a.B;
          float da[G];
          float nacc;
          // rows / units outside the batch carry zeros through the same arithmetic
          const float dh = SPLIT_Z ? (q_dout[jj][r] + acc[jj][r]) + dcs[jj][r] : q_dout[jj][r] + acc[jj][r];
          const f32x4 gv = SL::unpack(q_g[jj][r]);
          raw_t dgv;
          if constexpr (CELL == TAD_CELL_LSTM) {
            const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
            const float c = q_cur[jj][r], cprev = q_prev[jj][r];
            const float tc = fast_tanh(c);
            const float dog = dh * tc;
            const float dc = dh * og * (1.f - tc * tc) + dcs[jj][r];
            dcs[jj][r] = dc * fg;
            da[0] = dc * gg * ig * (1.f - ig);
            da[1] = dc * cprev * fg * (1.f - fg);
            da[2] = dc * ig * (1.f - gg * gg);
            da[3] = dog * og * (1.f - og);
            dgv = SL::pack(da[0], da[1], da[2], da[3]);
            nacc = 0.f;
          } else {
            const float rg = gv[0], zg = gv[1], ng = gv[2], hpn = gv[3];
            const float hprev = q_prev[jj][r];
            const float dn = dh * (1.f - zg);
            const float dz = dh * (hprev - ng);
            const float dan = dn * (1.f - ng * ng);
            const float dar = dan * hpn * rg * (1.f - rg);
            const float daz = dz * zg * (1.f - zg);
            da[0] = dar; da[1] = daz; da[2] = dan * rg;
            dgv = SL::pack(dar, daz, dan, da[2]);
            nacc = dh * zg;
          }
          if (ok) {
            unsigned char* dgp = reinterpret_cast<unsigned char*>(d.dgates) + (b0 + cr) * (int64_t)T * d.ld_dgates * SL::ESZ;
            *reinterpret_cast<raw_t*>(dgp + goff) = dgv;
          } else {
            nacc = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) da[g] = 0.f;
          }
          if constexpr (SPLIT_Z) { dcs[jj][r] = nacc; acc[jj][r] = 0.f; }
          else acc[jj][r] = nacc;
          if constexpr (!PHASED) {
#pragma unroll
            for (int g = 0; g < G; ++g) lds_put<BF16>(smem, a.lds_ld, row, g * a.width + u, da[g]);
          }
        }
      }
    }
    RSTAMP(s2);
    if constexpr (PREFETCH) {
      if (wave < a.NJ && step + 1 < T) fetch(step + 1, 0, true);
    }
    // ---- phase 2: dh_rec = dgates * W_hh (K = G * width), A operand from LDS
    if constexpr (!PHASED) {
      __syncthreads();
      RSTAMP(s3);
#pragma unroll
      for (int jj = 0; jj < JPW; ++jj) {
        const int j = wave + jj * nw;
        if (j < a.NJ) {
          if constexpr (MIXED) mma_mixed<PF, NKR>(acc[jj], smem, a.lds_ld, wp + (int64_t)j * a.nk_total * 64, a.nk_total, lane, wl, a.kl, wres);
          else mma_stream<BF16, 1, PF>(&acc[jj], smem, a.lds_ld, wp + (int64_t)j * a.nk_total * 64, a.nk_total, lane);
        }
      }
      RSTAMP(s4);
      __syncthreads();
      RSTAMP(s5);
#ifdef EXP_STAMPS
      sFetch += s1 - s0; sGate += s2 - s1; sBar1 += s3 - s2; sMma += s4 - s3; sBar2 += s5 - s4;
#endif
    } else {
      // the tile does not fit LDS: one gate at a time, re-read from the dgates rows this thread wrote itself
      for (int g = 0; g < G; ++g) {
        const int slot = (CELL == TAD_CELL_GRU && g == 2) ? 3 : g;     // GRU: the hidden-side n gradient
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
          const int j = wave + jj * nw;
          if (j < a.NJ) {
            const int u = 32 * j + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = acc_row32(r, lane);
              const int64_t b = b0 + row;
              float v = 0.f;
              if (u < H && row < a.rpw && b < a.B) v = SL::one(d.dgates, (b * T + tt) * d.ld_dgates + 4 * u + slot);
              lds_put<BF16>(smem, a.lds_ld, row, u, v);
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
          const int j = wave + jj * nw;
          if (j < a.NJ)
            mma_stream<BF16, 1, PF>(&acc[jj], smem, a.lds_ld, wp + ((int64_t)j * a.nk_total + (int64_t)g * a.nk_gate) * 64,
                                    a.nk_gate, lane);
        }
        __syncthreads();
      }
    }
  }
#ifdef EXP_STAMPS
  if (threadIdx.x == 0) {
    atomicAdd(&g_rnn_stamps[0], sFetch); atomicAdd(&g_rnn_stamps[1], sGate); atomicAdd(&g_rnn_stamps[2], sBar1);
    atomicAdd(&g_rnn_stamps[3], sMma); atomicAdd(&g_rnn_stamps[4], sBar2); atomicAdd(&g_rnn_stamps[7], (unsigned long long)T);
  }
#endif
}

// ---------------------------------------------------------------------------
// gate-interleaved image of the input-side parameters (see tad_rnn_pack_input)
struct PackInArgs {
  const float* w; const float* b_ih; const float* b_hh;
  void* w_gl; float* bias_gl;
  int H, G, in_dim, nblk, ld_gl, w16, lstm;
};

__global__ __launch_bounds__(256) void pack_input_kernel(PackInArgs a) {
  const int64_t rows = (int64_t)a.nblk * 4 * a.H;
  const int64_t total = rows * a.ld_gl;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / a.ld_gl;
    const int k = (int)(i % a.ld_gl);
    const int s = (int)(n & 3);
    const int64_t q = n >> 2;
    const int64_t blk = q / a.H, u = q % a.H;
    const int64_t src = (blk * a.G + s) * a.H + u;
    const bool real = s < a.G;
    const float v = (real && k < a.in_dim) ? a.w[src * a.in_dim + k] : 0.f;
    if (a.w16) reinterpret_cast<bf16_t*>(a.w_gl)[i] = (bf16_t)v;
    else reinterpret_cast<float*>(a.w_gl)[i] = v;
    if (k == 0) a.bias_gl[n] = real ? (a.lstm ? a.b_ih[src] + a.b_hh[src] : a.b_ih[src]) : 0.f;
  }
}

// ---------------------------------------------------------------------------
struct Geom {
  int NJ, width, nkf, nkb, jpw, nw;
  int nk32;      // 32-deep k-steps of the 16-row-form forward image (bf16 precision only)
};

// Rows of the batch one workgroup owns.  The MFMA tile is 32 rows; when the batch is too
// small to give every CU a workgroup, fewer real rows per workgroup spread the per-step
// HBM traffic (xproj in, h / stash out) over more CUs -- the step time is bound by the
// per-CU memory path, not by the matrix cores, in that regime.
int rows_per_wg(int64_t B, int ndir) {
  static const char* force = getenv("TAD_RNN_RPW");      // experiments: 8, 16 or 32
  if (force) return atoi(force);
  int rpw = 32;
  while (rpw > 8 && cdiv64(B, rpw) * ndir < 256) rpw >>= 1;
  return rpw;
}

Geom geom(int prec, int64_t H, int G) {
  Geom g;
  const int ks = ks_of(prec == TAD_BF16);
  g.NJ = (int)cdiv64(H, 32);
  g.width = 32 * g.NJ;
  g.nkf = (int)cdiv64(H, ks);
  g.nkb = G * g.width / ks;
  g.nk32 = prec == TAD_BF16 ? (int)cdiv64(H, 32) : 0;
  g.jpw = g.NJ <= 8 ? 1 : (g.NJ <= 16 ? 2 : 4);
  g.nw = (int)cdiv64(g.NJ, g.jpw);
  return g;
}

constexpr size_t LDS_LIMIT = 160 * 1024;

// raises the dynamic-LDS limit of a kernel; `configured` caches the largest size already set
template <typename K>
int set_lds(K kernel, size_t bytes, size_t (&per_device)[64]) {
  // the attribute belongs to the (kernel, device) pair: remember what was set for the device that is current
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { tad_set_error("set_lds: hipGetDevice failed or device index beyond 63"); return TAD_ERR_LAUNCH; }
  size_t& configured = per_device[dev];
  if (bytes <= configured) return TAD_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    tad_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", bytes, hipGetErrorString(e));
    return TAD_ERR_LAUNCH;
  }
  configured = bytes;
  return TAD_OK;
}

template <int CELL, bool BF16, bool ST16, int JPW, int NKR = 0, int RL = 0>
int launch_fwd(const RnnFwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto k = rnn_fwd_kernel<CELL, BF16, ST16, JPW, NKR, RL>;
  static size_t configured[64] = {};
  if (int rc = set_lds(k, lds, configured)) return rc;
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_fwd_kernel<%d, %s, %s, %d, %d, %d>", CELL, BF16 ? "true" : "false", ST16 ? "true" : "false", JPW, NKR, RL);
  return TAD_OK;
}

constexpr int RES_NKR = 8;     // k-blocks of W_hh the resident form keeps in registers (4 VGPRs per k-block and gate)
constexpr int F16_NKR_LSTM = 5, F16_NKR_GRU = 4;   // 32-deep k-steps the 16-row form keeps in registers (8 / 6 blocks of 4 VGPRs each)

template <int CELL, int NKR, int RPW = 8, bool STASH = false>
int launch_fwd16(const RnnFwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto k = rnn_fwd16_kernel<CELL, NKR, RPW, STASH>;
  static size_t configured[64] = {};
  if (int rc = set_lds(k, lds, configured)) return rc;
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_fwd16_kernel<%d, %d, %d, %s>", CELL, NKR, RPW, STASH ? "true" : "false");
  return TAD_OK;
}

constexpr int BWD_NKR = 16;    // k-blocks of the transposed W_hh image the backward keeps in registers (4 VGPRs each)

template <int CELL, bool BF16, bool ST16, int JPW, bool PHASED, int NKR = 0>
int launch_bwd(const RnnBwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto k = rnn_bwd_kernel<CELL, BF16, ST16, JPW, PHASED, NKR>;
  static size_t configured[64] = {};
  if (int rc = set_lds(k, lds, configured)) return rc;
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_bwd_kernel<%d, %s, %s, %d, %s, %d>", CELL, BF16 ? "true" : "false", ST16 ? "true" : "false", JPW,
                  PHASED ? "true" : "false", NKR);
  return TAD_OK;
}

}  // namespace

extern "C" int64_t tad_rnn_packed_bytes(int prec, int64_t H, int G) {
  if (H <= 0 || (G != 3 && G != 4) || (prec != TAD_F32 && prec != TAD_BF16)) return -1;
  const Geom g = geom(prec, H, G);
  // (forward buffer: the 32-row-form image, then in bf16 precision the 16-row-form image, then the cluster-form image)
  const int64_t f = rnn_fwd_image_blocks(prec, H, G) + rnn_cluster_image_blocks(prec, H), b = (int64_t)g.NJ * g.nkb;
  return (f > b ? f : b) * 1024;
}

extern "C" int tad_rnn_pack_weights(int prec, int64_t H, int G, const float* w_hh, void* packed_fwd, void* packed_bwd,
                                    tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_pack_weights: bad prec %d", prec);
  TAD_CHECK(H > 0 && H <= 1024 && (G == 3 || G == 4) && w_hh, "tad_rnn_pack_weights: bad arguments H=%ld G=%d", (long)H, G);
  const Geom g = geom(prec, H, G);
  for (int bw = 0; bw < 2; ++bw) {
    void* out = bw ? packed_bwd : packed_fwd;
    if (!out) continue;
    PackArgs a{};
    a.w = w_hh; a.out = reinterpret_cast<uint4*>(out);
    a.H = (int)H; a.G = G; a.NJ = g.NJ; a.nk = bw ? g.nkb : g.nkf; a.width = g.width;
    a.bf16 = prec == TAD_BF16; a.backward = bw;
    const int64_t nblocks = bw ? (int64_t)g.NJ * g.nkb : (int64_t)g.NJ * g.nkf * G;
    const int64_t threads = nblocks * 64;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
    TAD_LAUNCH_CHECK();
  }
  if (packed_fwd && g.nk32 > 0) {
    Pack16Args a{};
    a.w = w_hh; a.out = reinterpret_cast<uint4*>(packed_fwd) + (int64_t)g.NJ * g.nkf * G * 64;
    a.H = (int)H; a.G = G; a.NJ = g.NJ; a.nk32 = g.nk32;
    const int64_t threads = (int64_t)g.NJ * g.nk32 * G * 2 * 64;
    hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
    TAD_LAUNCH_CHECK();
  }
  if (packed_fwd && rnn_cluster_image_blocks(prec, H) > 0) {
    PackClArgs a{w_hh, reinterpret_cast<uint4*>(packed_fwd) + rnn_fwd_image_blocks(prec, H, G) * 64, (int)H, G};
    const int64_t threads = rnn_cluster_image_blocks(prec, H) * 64;
    hipLaunchKernelGGL(pack_cluster_kernel, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
    TAD_LAUNCH_CHECK();
  }
  return TAD_OK;
}

extern "C" int tad_rnn_pack_weights_multi(int prec, int64_t H, int G, int njobs, const tad_rnn_pack_job_t* jobs, tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_pack_weights_multi: bad prec %d", prec);
  TAD_CHECK(H > 0 && H <= 1024 && (G == 3 || G == 4) && jobs && njobs >= 1 && njobs <= PACK_MAXJOBS, "tad_rnn_pack_weights_multi: bad arguments H=%ld G=%d njobs=%d (1..%d)",
            (long)H, G, njobs, PACK_MAXJOBS);
  const Geom g = geom(prec, H, G);
  PackMultiArgs a{};
  for (int k = 0; k < njobs; ++k) {
    TAD_CHECK(jobs[k].w_hh, "tad_rnn_pack_weights_multi: job %d: null w_hh", k);
    a.w[k] = jobs[k].w_hh;
    a.fwd[k] = reinterpret_cast<uint4*>(jobs[k].packed_fwd);
    a.bwd[k] = reinterpret_cast<uint4*>(jobs[k].packed_bwd);
  }
  a.H = (int)H; a.G = G; a.NJ = g.NJ; a.nkf = g.nkf; a.nkb = g.nkb; a.nk32 = g.nk32; a.width = g.width; a.bf16 = prec == TAD_BF16;
  a.nf = (int64_t)g.NJ * g.nkf * G * 64; a.nb = (int64_t)g.NJ * g.nkb * 64; a.n16 = (int64_t)g.NJ * g.nk32 * G * 2 * 64;
  a.ncl = rnn_cluster_image_blocks(prec, H) * 64;
  int64_t gx = cdiv64(a.nf + a.nb + a.n16 + a.ncl, 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("pack_multi_kernel");
  return TAD_OK;
}

namespace {
// typed gate-interleaved pointers: 16-byte aligned base, leading dimension a multiple of 4
bool gl_ok(const void* p, int64_t ld, int64_t H) { return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0 && ld >= 4 * H; }
}  // namespace

extern "C" int tad_rnn_fwd(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir,
                           const tad_rnn_dir_t* dirs, tad_stream_t stream) {
  TAD_CHECK(cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU, "tad_rnn_fwd: bad cell %d", cell);
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_fwd: bad prec %d", prec);
  TAD_CHECK(store == TAD_F32 || (store == TAD_BF16 && prec == TAD_BF16), "tad_rnn_fwd: bad store %d (bf16 storage needs prec = TAD_BF16)", store);
  TAD_CHECK(B >= 0 && T > 0 && H > 0 && H <= 1024 && (ndir == 1 || ndir == 2) && dirs, "tad_rnn_fwd: bad shape B=%ld T=%ld H=%ld ndir=%d",
            (long)B, (long)T, (long)H, ndir);
  if (B == 0) return TAD_OK;                   // an empty shard of a batch: nothing to do
  const int G = cell == TAD_CELL_LSTM ? 4 : 3;
  const bool bf = prec == TAD_BF16, st16 = store == TAD_BF16;
  const Geom g = geom(prec, H, G);
  RnnFwdArgs a{};
  for (int i = 0; i < ndir; ++i) {
    a.d[i] = dirs[i];
    const tad_rnn_dir_t& di = dirs[i];
    TAD_CHECK(di.xproj && di.w_packed && (di.out || di.out16), "tad_rnn_fwd: null pointer in dir %d", i);
    TAD_CHECK(!di.out16 || di.ld_out16 >= H, "tad_rnn_fwd: bad ld_out16 in dir %d", i);
    TAD_CHECK(gl_ok(di.xproj, di.ld_xproj, H) && (!di.out || di.ld_out >= H), "tad_rnn_fwd: bad xproj alignment / leading dimension in dir %d", i);
    TAD_CHECK(!di.gates || (gl_ok(di.gates, di.ld_gates, H) && (cell == TAD_CELL_GRU || (di.cstate && di.ld_c >= H))),
              "tad_rnn_fwd: bad gate stash (alignment, leading dimension, or missing cstate) in dir %d", i);
    TAD_CHECK((5 * T + 1) * (di.ld_xproj + di.ld_out + di.ld_out16 + di.ld_gates + di.ld_c) * 4 < (int64_t)1 << 31, "tad_rnn_fwd: T * ld too large for 32-bit lane offsets");
  }
  a.B = (int)B; a.T = (int)T; a.H = (int)H; a.NJ = g.NJ; a.nk = g.nkf;
  a.rpw = rows_per_wg(B, ndir);
  a.lds_ld = g.width + (bf ? 8 : 4);
  {
    // The training LSTM at full 32-row tiles streams 36 of its 52 W_hh blocks per wave and step; two rounds of
    // 16-row workgroups in the resident form (nothing streamed) are faster when that form fits (at H = 200: 330 -> 305 us
    // for 4096 windows).  Without the gate stash (inference) the streaming form at 32 rows wins.
    static const char* force = getenv("TAD_RNN_RPW");
    const int kl = g.nkf - RES_NKR;
    const size_t need16 = (size_t)2 * 16 * a.lds_ld * 2 + (size_t)(kl > 0 ? kl : 0) * g.NJ * G * 1024;
    if (!force && cell == TAD_CELL_LSTM && a.rpw == 32 && bf && st16 && g.jpw == 1 && kl >= 0 && need16 <= LDS_LIMIT && dirs[0].gates)
      a.rpw = 16;
  }
  a.out16_rows = bf && H % 8 == 0 && a.lds_ld % 8 == 0;
  for (int i = 0; i < ndir; ++i)
    if (dirs[i].out16 && (dirs[i].ld_out16 % 8 != 0 || (reinterpret_cast<uintptr_t>(dirs[i].out16) & 15) != 0)) a.out16_rows = 0;
  a.tile_rows = 32;
  size_t lds = (size_t)2 * 32 * a.lds_ld * (bf ? 2 : 4);
  TAD_CHECK(lds <= LDS_LIMIT, "tad_rnn_fwd: H=%ld needs %zu bytes of LDS in this precision (limit %zu)", (long)H, lds, LDS_LIMIT);
  a.kl = 0;
  a.ndir = ndir; a.xcd_dirs = rnn_xcd_dirs(H, G); a.ntiles = (int)cdiv64(B, a.rpw);
  dim3 grid(rnn_grid(a.ntiles, ndir, a.xcd_dirs)), block((unsigned)g.nw * 64);
  hipStream_t s = (hipStream_t)stream;
  {
    // The 16-row form, if W_hh fits its registers + LDS, at 8 rows per workgroup: the sampler, and training at the
    // reference's batch of 100 (LSTM 80 -> 64 us, GRU 66 -> 57 us per launch).  (The training GRU at full batches as two
    // rounds of 16-row workgroups in this form measured the same 217 us as its 32-row resident form: there the step is
    // bound by its stores and gate math, not by the matrix pipe.)
    static const char* f16_env = getenv("TAD_RNN_F16");      // experiments: 0 = never, 1 = only without a stash
    const int f16 = f16_env ? atoi(f16_env) : 2;
    const int nkr = cell == TAD_CELL_LSTM ? F16_NKR_LSTM : F16_NKR_GRU;
    const int kl = g.nk32 - nkr;
    const int ld16 = 32 * g.nk32 + 8;
    bool stash = false, all_stash = true;
    for (int i = 0; i < ndir; ++i) { stash = stash || dirs[i].gates; all_stash = all_stash && dirs[i].gates; }
    const int rpw = a.rpw;
    const size_t need = (size_t)2 * rpw * ld16 * 2 + (size_t)(kl > 0 ? kl : 0) * g.NJ * G * 2 * 1024;
    const bool shape_ok = rpw == 8;
    if (bf && st16 && g.jpw == 1 && shape_ok && (!stash || (all_stash && f16 >= 2)) && kl >= 0 && need <= LDS_LIMIT && f16 >= 1) {
      RnnFwdArgs b = a;
      b.rpw = rpw; b.kl = kl; b.nk32 = g.nk32; b.lds_ld = ld16; b.tile_rows = rpw;
      b.out16_rows = H % 8 == 0;
      for (int i = 0; i < ndir; ++i)
        if (dirs[i].out16 && (dirs[i].ld_out16 % 8 != 0 || (reinterpret_cast<uintptr_t>(dirs[i].out16) & 15) != 0)) b.out16_rows = 0;
      b.ntiles = (int)cdiv64(B, rpw);
      dim3 grid16(rnn_grid(b.ntiles, ndir, b.xcd_dirs));
      if (cell == TAD_CELL_LSTM)
        return stash ? launch_fwd16<TAD_CELL_LSTM, F16_NKR_LSTM, 8, true>(b, grid16, block, need, s)
                     : launch_fwd16<TAD_CELL_LSTM, F16_NKR_LSTM, 8, false>(b, grid16, block, need, s);
      return stash ? launch_fwd16<TAD_CELL_GRU, F16_NKR_GRU, 8, true>(b, grid16, block, need, s)
                   : launch_fwd16<TAD_CELL_GRU, F16_NKR_GRU, 8, false>(b, grid16, block, need, s);
    }
  }
  {
    // small batches: the resident form, if the k-blocks that do not fit its registers fit the LDS beside the h tiles
    static const char* res_env = getenv("TAD_RNN_RES");      // experiments: 0 = always the streaming form
    // (the full-tile GRU keeps one k-block fewer in registers: its 16 live accumulator rows need them)
    const int nkr = (cell == TAD_CELL_GRU && a.rpw == 32) ? RES_NKR - 1 : RES_NKR;
    const int kl = g.nkf - nkr;
    const size_t need = (size_t)2 * a.rpw * a.lds_ld * 2 + (size_t)(kl > 0 ? kl : 0) * g.NJ * G * 1024;
    // (full 32-row tiles: the GRU's three gates leave room in LDS and registers at H = 200, the LSTM's four do not)
    const bool rows_ok = a.rpw <= 16 || cell == TAD_CELL_GRU;
    if (bf && st16 && g.jpw == 1 && rows_ok && kl >= 0 && need <= LDS_LIMIT && !(res_env && atoi(res_env) == 0)) {
      a.kl = kl; a.tile_rows = a.rpw;
      if (cell == TAD_CELL_LSTM)
        return a.rpw == 8 ? launch_fwd<TAD_CELL_LSTM, true, true, 1, RES_NKR, 4>(a, grid, block, need, s)
                          : launch_fwd<TAD_CELL_LSTM, true, true, 1, RES_NKR, 8>(a, grid, block, need, s);
      if (a.rpw == 32) return launch_fwd<TAD_CELL_GRU, true, true, 1, RES_NKR - 1, 16>(a, grid, block, need, s);
      return a.rpw == 8 ? launch_fwd<TAD_CELL_GRU, true, true, 1, RES_NKR, 4>(a, grid, block, need, s)
                        : launch_fwd<TAD_CELL_GRU, true, true, 1, RES_NKR, 8>(a, grid, block, need, s);
    }
  }
  if (g.jpw == 1) {
    static const char* force = getenv("TAD_RNN_KL");      // experiments: k-blocks of W_hh kept in LDS
    const size_t per_kl = (size_t)g.NJ * G * 1024;
    int64_t kl = (int64_t)((LDS_LIMIT - lds) / per_kl);
    if (kl > g.nkf) kl = g.nkf;
    if (force && atoi(force) < kl) kl = atoi(force);
    a.kl = (int)kl;
    lds += (size_t)a.kl * per_kl;
  }
#define TAD_FWD(C, BFV, STV)                                                              \
  (g.jpw == 1 ? launch_fwd<C, BFV, STV, 1>(a, grid, block, lds, s)                         \
              : (g.jpw == 2 ? launch_fwd<C, BFV, STV, 2>(a, grid, block, lds, s) : launch_fwd<C, BFV, STV, 4>(a, grid, block, lds, s)))
  if (cell == TAD_CELL_LSTM) return bf ? (st16 ? TAD_FWD(TAD_CELL_LSTM, true, true) : TAD_FWD(TAD_CELL_LSTM, true, false)) : TAD_FWD(TAD_CELL_LSTM, false, false);
  return bf ? (st16 ? TAD_FWD(TAD_CELL_GRU, true, true) : TAD_FWD(TAD_CELL_GRU, true, false)) : TAD_FWD(TAD_CELL_GRU, false, false);
#undef TAD_FWD
}

extern "C" int tad_rnn_bwd(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir,
                           const tad_rnn_bwd_dir_t* dirs, tad_stream_t stream) {
  TAD_CHECK(cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU, "tad_rnn_bwd: bad cell %d", cell);
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_bwd: bad prec %d", prec);
  TAD_CHECK(store == TAD_F32 || (store == TAD_BF16 && prec == TAD_BF16), "tad_rnn_bwd: bad store %d (bf16 storage needs prec = TAD_BF16)", store);
  TAD_CHECK(B >= 0 && T > 0 && H > 0 && H <= 1024 && (ndir == 1 || ndir == 2) && dirs, "tad_rnn_bwd: bad shape B=%ld T=%ld H=%ld ndir=%d",
            (long)B, (long)T, (long)H, ndir);
  if (B == 0) return TAD_OK;
  const int G = cell == TAD_CELL_LSTM ? 4 : 3;
  const bool bf = prec == TAD_BF16, st16 = store == TAD_BF16;
  const Geom g = geom(prec, H, G);
  RnnBwdArgs a{};
  for (int i = 0; i < ndir; ++i) {
    a.d[i] = dirs[i];
    const tad_rnn_bwd_dir_t& di = dirs[i];
    TAD_CHECK(di.dout && di.gates && di.w_packed_t && di.dgates, "tad_rnn_bwd: null pointer in dir %d", i);
    TAD_CHECK(cell == TAD_CELL_GRU ? (di.out && di.ld_out >= H) : (di.cstate && di.ld_c >= H),
              "tad_rnn_bwd: LSTM needs cstate, GRU needs out (dir %d)", i);
    TAD_CHECK(di.ld_dout >= H && gl_ok(di.gates, di.ld_gates, H) && gl_ok(di.dgates, di.ld_dgates, H),
              "tad_rnn_bwd: bad alignment / leading dimension in dir %d", i);
    TAD_CHECK((32 * T + 1) * (di.ld_dout + di.ld_out + di.ld_gates + di.ld_c + di.ld_dgates) * 4 < (int64_t)1 << 31,
              "tad_rnn_bwd: T * ld too large for 32-bit lane offsets");
  }
  const int ks = ks_of(bf);
  a.B = (int)B; a.T = (int)T; a.H = (int)H; a.NJ = g.NJ; a.width = g.width;
  a.rpw = rows_per_wg(B, ndir);
  a.nk_total = g.nkb; a.nk_gate = g.width / ks;
  const int pad = bf ? 8 : 4;
  const size_t esz = bf ? 2 : 4;
  size_t lds = (size_t)32 * (G * g.width + pad) * esz;
  static const char* phased_env = getenv("TAD_RNN_BWD_PHASED");      // experiments: 1 = the gate-by-gate form wherever it exists
  const bool phased = lds > LDS_LIMIT || (phased_env && atoi(phased_env) == 1 && g.jpw >= 2);
  a.lds_ld = (phased ? g.width : G * g.width) + pad;
  lds = (size_t)32 * a.lds_ld * esz;
  TAD_CHECK(lds <= LDS_LIMIT, "tad_rnn_bwd: H=%ld needs %zu bytes of LDS (limit %zu)", (long)H, lds, LDS_LIMIT);
  a.ndir = ndir; a.xcd_dirs = rnn_xcd_dirs(H, G); a.ntiles = (int)cdiv64(B, a.rpw);
  dim3 grid(rnn_grid(a.ntiles, ndir, a.xcd_dirs)), block((unsigned)g.nw * 64);
  hipStream_t s = (hipStream_t)stream;
  a.kl = 0;
  if (bf && g.jpw == 1 && !phased) {
    // one-pass bf16 form, one chunk per wave: the transposed W_hh image stays on the chip as far as it fits -- the
    // leading k-blocks of every wave in the LDS behind the tile, the next BWD_NKR in registers when that many are left
    static const char* kl_env = getenv("TAD_RNN_BWD_KL");      // experiments: cap on the k-blocks kept in LDS (0 with
    static const char* nkr_env = getenv("TAD_RNN_BWD_NKR");    // TAD_RNN_BWD_NKR=0: the streaming form)
    int64_t kl = (int64_t)((LDS_LIMIT - lds) / ((size_t)g.NJ * 1024));
    if (kl > g.nkb) kl = g.nkb;
    if (kl_env && atoi(kl_env) < kl) kl = atoi(kl_env);
    // (with an fp32 gate stash the one-pass form has no registers to spare)
    const bool regs = st16 && g.nkb - kl >= BWD_NKR && !(nkr_env && atoi(nkr_env) == 0);
    a.kl = (int)kl;
    lds += (size_t)kl * g.NJ * 1024;
    if (regs) {
      if (cell == TAD_CELL_LSTM) return launch_bwd<TAD_CELL_LSTM, true, true, 1, false, BWD_NKR>(a, grid, block, lds, s);
      return launch_bwd<TAD_CELL_GRU, true, true, 1, false, BWD_NKR>(a, grid, block, lds, s);
    }
  }
#define TAD_BWD2(C, BFV, STV, J) (phased ? launch_bwd<C, BFV, STV, J, true>(a, grid, block, lds, s) : launch_bwd<C, BFV, STV, J, false>(a, grid, block, lds, s))
#define TAD_BWD(C, BFV, STV) (g.jpw == 1 ? TAD_BWD2(C, BFV, STV, 1) : (g.jpw == 2 ? TAD_BWD2(C, BFV, STV, 2) : TAD_BWD2(C, BFV, STV, 4)))
  if (cell == TAD_CELL_LSTM) return bf ? (st16 ? TAD_BWD(TAD_CELL_LSTM, true, true) : TAD_BWD(TAD_CELL_LSTM, true, false)) : TAD_BWD(TAD_CELL_LSTM, false, false);
  return bf ? (st16 ? TAD_BWD(TAD_CELL_GRU, true, true) : TAD_BWD(TAD_CELL_GRU, true, false)) : TAD_BWD(TAD_CELL_GRU, false, false);
#undef TAD_BWD
#undef TAD_BWD2
}

extern "C" int tad_rnn_pack_input(int cell, int64_t H, int64_t in_dim, int nblk, const float* w_ih, const float* b_ih,
                                  const float* b_hh, void* w_gl, int w_dtype, int64_t ld_gl, float* bias_gl,
                                  tad_stream_t stream) {
  TAD_CHECK(cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU, "tad_rnn_pack_input: bad cell %d", cell);
  TAD_CHECK(H > 0 && H <= 1024 && in_dim > 0 && nblk > 0 && ld_gl >= in_dim && ld_gl < (1 << 22), "tad_rnn_pack_input: bad shape H=%ld in=%ld nblk=%d ld=%ld",
            (long)H, (long)in_dim, nblk, (long)ld_gl);
  TAD_CHECK(w_ih && b_ih && w_gl && bias_gl && (cell == TAD_CELL_GRU || b_hh), "tad_rnn_pack_input: null pointer");
  TAD_CHECK(w_dtype == TAD_F32 || w_dtype == TAD_BF16, "tad_rnn_pack_input: bad w_dtype %d", w_dtype);
  PackInArgs a{};
  a.w = w_ih; a.b_ih = b_ih; a.b_hh = b_hh; a.w_gl = w_gl; a.bias_gl = bias_gl;
  a.H = (int)H; a.G = cell == TAD_CELL_LSTM ? 4 : 3; a.in_dim = (int)in_dim; a.nblk = nblk; a.ld_gl = (int)ld_gl;
  a.w16 = w_dtype == TAD_BF16; a.lstm = cell == TAD_CELL_LSTM;
  const int64_t total = (int64_t)nblk * 4 * H * ld_gl;
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
