// MFMA GEMM for the nn.Linear family on gfx950: C[i,j] = sum_k A(i,k) * B(j,k).
//
// One kernel template serves the three products of a linear layer by describing each operand
// as "k-contiguous" (activation rows, weight rows) or "row-contiguous" (the same memory read as
// its transpose).  Tiles are staged global -> registers -> LDS as [row][k] images, so the MFMA
// fragment reads are identical in all cases.
//
// These products are bound by the bytes and, even more, by the number of vector-memory
// instructions a CU pushes through its memory path (a wave-wide load costs about 16 cycles of that
// path whether it moves 4 or 16 bytes per lane), not by the matrix cores.  Hence:
//   * every global load is 16 bytes per lane: k-contiguous operands along k, row-contiguous
//     operands along their rows (4 rows of one k per lane, transposed in registers);
//   * loads are unconditional (a load inside a divergent branch is waited for where the branch
//     ends, which serialises a k-step's loads): indices are clamped into the operand and whatever
//     must read as zero is zeroed in registers afterwards;
//   * tiles are as wide as the accumulator file allows, so operands are re-read less often:
//
//   W : 128 x 512, 8 waves (2 x 4), wave tile 64 x 128 (2 x 4 accumulators of 32 x 32)   bf16
//   S : 256 x 256, 8 waves (4 x 2), wave tile 64 x 128                                    bf16
//   F : 128 x 256, 8 waves (2 x 4), wave tile 64 x 64                                     fp32
//   T : 128 x 128, 4 waves (2 x 2), wave tile 64 x 64       small problems (fill the chip)
//
//   fp32 : v_mfma_f32_32x32x2_f32;  bf16 : v_mfma_f32_32x32x16_bf16 (operands rounded on staging)
// LDS image of a tile: unpadded rows of 32 k (64 B in bf16, 128 B in fp32); the 16-byte chunk c of
// row r sits at chunk position c ^ sw(r), which makes the 16-byte fragment reads conflict-free and
// the transposing stores at most 2-way.  Two LDS buffers: the global loads of k-step t+1 are in
// flight under the MFMAs of step t and are written to the other buffer afterwards -- one barrier
// per step.  Tiles are handed to workgroups so that the workgroups of one XCD (blockIdx % 8 equal)
// walk consecutive tiles with the column tile fastest: the row panel they share stays in that L2.
#pragma once
#include "common.h"
#include <stdlib.h>

// Diagnostic build only (-DEXP_STAMPS): per-phase cycle counts of wave 0 of every workgroup, summed
// into a buffer of their own that no kernel reads.
#ifdef EXP_STAMPS
__device__ unsigned long long g_stamps[8];
#define STAMP(var) __builtin_amdgcn_sched_barrier(0); unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0)
#else
#define STAMP(var)
#endif

namespace {

constexpr int BK = 32;

// Operand kinds: how the [row][k] operand of the product is stored in memory.
//   KC32 / KC16: k-contiguous rows (fp32 / bf16);  RC32 / RC16: row-contiguous, element (row,k) at P[k*s_k + row].
// 16-bit operands are staged without touching the values: KC16 chunks go into the swizzled [row][k] image,
// RC16 chunks go into a plain [k][row] image that the fragments are read from with the transposing LDS
// read of gfx950 (ds_read_b64_tr_b16).
enum OpKind { KC32 = 0, RC32 = 1, KC16 = 2, RC16 = 3 };

struct GemmParams {
  const void* A; int64_t sa_row, sa_k;
  const void* B; int64_t sb_row, sb_k;
  void* C; int64_t ldc;
  int c16;                   // C is bf16 (wide epilogue only)
  int64_t M, N, K;
  const float* bias; const float* bias2;
  const float* relu_of; int64_t ld_relu;
  int relu16;                // relu_of points to bf16 values (wide epilogue only)
  const float* residual; int64_t ld_res;
  float* rowsum; float* rowsum2;   // += sum_k A(i,k) (bias gradients; added by the first column tile)
  int act, atomic;
  int wide_epi;              // 16-byte epilogue possible (alignment of C, bias, residual, relu operand)
  int64_t k_per_split;
  int shiftT, shift;         // time-shifted rows of the B operand along k (see header)
  int64_t tiles_n, ntiles, nsplits;
  // Gate map of the atomic epilogue and the row sums (map_H == 0: none).  Row n of the product is the
  // gate-interleaved index blk*4*map_H + 4*u + s; it is accumulated into row blk*map_G*map_H +
  // map_slot[s]*map_H + u of C (rowsum likewise), or dropped when map_slot[s] < 0.
  int map_H, map_G;
  int map_slot[4];
};

__device__ __forceinline__ int64_t map_row(const GemmParams& p, int64_t n) {
  if (p.map_H == 0) return n;
  const uint32_t s = (uint32_t)n & 3u, q = (uint32_t)n >> 2;
  const uint32_t blk = q / (uint32_t)p.map_H, u = q % (uint32_t)p.map_H;
  const int g = s == 0 ? p.map_slot[0] : s == 1 ? p.map_slot[1] : s == 2 ? p.map_slot[2] : p.map_slot[3];
  return g < 0 ? (int64_t)-1 : (int64_t)blk * p.map_G * p.map_H + (int64_t)g * p.map_H + u;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16_t x = (bf16_t)a, y = (bf16_t)b;
  uint16_t ux = __builtin_bit_cast(uint16_t, x), uy = __builtin_bit_cast(uint16_t, y);
  return (uint32_t)ux | ((uint32_t)uy << 16);
}

// ---- LDS tile image
template <bool BF16>
__device__ __forceinline__ int lds_sw(int row) { return BF16 ? (row >> 2) & 3 : (row >> 1) & 7; }
// byte offset of element k (k * element size a multiple of the access size) of row `row`
template <bool BF16>
__device__ __forceinline__ uint32_t lds_off(int row, int k) {
  const int byte = k * (BF16 ? 2 : 4);
  return (uint32_t)(row * (BF16 ? 64 : 128) + ((((byte >> 4) ^ lds_sw<BF16>(row))) << 4) + (byte & 15));
}

// store n consecutive k (n * element size = 4, 8 or a multiple of 16 bytes) of one row, from fp32 registers
template <bool BF16, int N>
__device__ __forceinline__ void lds_store_run(unsigned char* lds, int row, int k, const float* v) {
  if constexpr (BF16) {
    if constexpr (N == 2) {
      *reinterpret_cast<uint32_t*>(lds + lds_off<true>(row, k)) = pack_bf16x2(v[0], v[1]);
    } else if constexpr (N == 4) {
      uint2 u;
      u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(lds + lds_off<true>(row, k)) = u;
    } else {
      static_assert(N % 8 == 0, "run length");
#pragma unroll
      for (int j = 0; j < N / 8; ++j) {
        uint4 u;
        u.x = pack_bf16x2(v[8 * j + 0], v[8 * j + 1]); u.y = pack_bf16x2(v[8 * j + 2], v[8 * j + 3]);
        u.z = pack_bf16x2(v[8 * j + 4], v[8 * j + 5]); u.w = pack_bf16x2(v[8 * j + 6], v[8 * j + 7]);
        *reinterpret_cast<uint4*>(lds + lds_off<true>(row, k + 8 * j)) = u;
      }
    }
  } else {
    if constexpr (N == 2) {
      *reinterpret_cast<float2*>(lds + lds_off<false>(row, k)) = make_float2(v[0], v[1]);
    } else {
      static_assert(N % 4 == 0, "run length");
#pragma unroll
      for (int j = 0; j < N / 4; ++j)
        *reinterpret_cast<float4*>(lds + lds_off<false>(row, k + 4 * j)) = make_float4(v[4 * j + 0], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
  }
}

// ---- k-contiguous operand: tile R rows x 32 k; thread -> rows tid/8 + (NT/8)*it, 4 floats at k = 4*(tid%8).
// Addresses are a wave-uniform base (scalar registers) plus a 32-bit per-lane offset.
// VEC: rows 16-byte aligned and the k range a multiple of 4 -> one 16-byte load per row
template <int R, int NT, bool VEC>
__device__ __forceinline__ void load_kc(float* reg, const float* __restrict__ slab, int64_t s_row, int64_t row0,
                                        int64_t rows, int64_t k0, int64_t kend, int tid) {
  constexpr int ITS = R * 8 / NT;
  static_assert(ITS >= 1 && R * 8 % NT == 0, "tile rows vs threads");
  const int c4 = tid & 7, r = tid >> 3;
  const int64_t krem64 = kend - k0, nrow64 = rows - row0;
  const int krem = (int)(krem64 < BK ? krem64 : BK);
  const int nrow1 = (int)(nrow64 < R ? nrow64 : R) - 1;
  const float* base = slab;            // element (row0, k0): the kernel advances it from slab to slab
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    int row = r + (NT / 8) * it;
    row = row < nrow1 ? row : nrow1;
    if constexpr (VEC) {
      const int kq = (4 * c4 + 4 <= krem) ? 4 * c4 : 0;
      const float4 v = *reinterpret_cast<const float4*>(base + ((uint32_t)row * (uint32_t)s_row + (uint32_t)kq));
      reg[4 * it + 0] = v.x; reg[4 * it + 1] = v.y; reg[4 * it + 2] = v.z; reg[4 * it + 3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = (4 * c4 + i < krem) ? 4 * c4 + i : 0;
        reg[4 * it + i] = base[(uint32_t)row * (uint32_t)s_row + (uint32_t)kk];
      }
    }
  }
}

template <bool BF16, int R, int NT>
__device__ __forceinline__ void store_kc(float* reg, unsigned char* lds, int64_t k0, int64_t kend, int tid) {
  constexpr int ITS = R * 8 / NT;
  const int c4 = tid & 7, r = tid >> 3;
  const int64_t krem = kend - k0;
  if (krem < BK) {      // zero the k tail (only the last k-step of a range has one)
#pragma unroll
    for (int it = 0; it < ITS; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * c4 + i >= (int)krem) reg[4 * it + i] = 0.f;
  }
#pragma unroll
  for (int it = 0; it < ITS; ++it) lds_store_run<BF16, 4>(lds, r + (NT / 8) * it, 4 * c4, reg + 4 * it);
}

// ---- k-contiguous operand already stored as bf16 (pre-converted weights): tile R rows x 32 k = 4
// chunks of 16 bytes per row; thread -> rows tid/4 + (NT/4)*it, chunk tid%4.  No conversion: the
// chunk goes to LDS as it is.  Needs 16-byte aligned rows and a reduction length that is a multiple of 8.
template <int R, int NT>
__device__ __forceinline__ void load_kc16(uint4* reg, const bf16_t* __restrict__ slab, int64_t s_row, int64_t row0,
                                          int64_t rows, int64_t k0, int64_t kend, int tid) {
  constexpr int ITS = R * 4 / NT;
  static_assert(ITS >= 1 && R * 4 % NT == 0, "tile rows vs threads");
  const int c = tid & 3, r = tid >> 2;
  const int64_t krem64 = kend - k0, nrow64 = rows - row0;
  const int krem = (int)(krem64 < BK ? krem64 : BK);
  const int nrow1 = (int)(nrow64 < R ? nrow64 : R) - 1;
  const int kq = (8 * c + 8 <= krem) ? 8 * c : 0;
  const bf16_t* base = slab;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    int row = r + (NT / 4) * it;
    row = row < nrow1 ? row : nrow1;
    reg[it] = *reinterpret_cast<const uint4*>(base + ((uint32_t)row * (uint32_t)s_row + (uint32_t)kq));
  }
}

template <int R, int NT>
__device__ __forceinline__ void store_kc16(uint4* reg, unsigned char* lds, int64_t k0, int64_t kend, int tid) {
  constexpr int ITS = R * 4 / NT;
  const int c = tid & 3, r = tid >> 2;
  const int64_t krem = kend - k0;
  if (krem < BK && 8 * c >= (int)krem) {      // chunks beyond the reduction range (a multiple of 8) read as zero
#pragma unroll
    for (int it = 0; it < ITS; ++it) reg[it] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int it = 0; it < ITS; ++it) *reinterpret_cast<uint4*>(lds + lds_off<true>(r + (NT / 4) * it, 8 * c)) = reg[it];
}

// ---- row-contiguous bf16 operand (element (row,k) at P[k*s_k + row]): tile 32 k x R rows = R/8 chunks of
// 16 bytes (8 consecutive rows) per k; thread -> chunk tid % (R/8), k = tid / (R/8) + KPP*it.  The chunks go
// to LDS as they are, into a [k][row] image whose k-rows are R*2 + 64 bytes apart: four consecutive k-rows
// then start 64 bytes (16 banks) apart, which makes both the 16-byte stores and the transposing fragment
// reads conflict-free.  Needs 16-byte aligned k-rows and a row count that is a multiple of 8.
// Returns the bit mask of the thread's k positions that hold real data.
template <int R> constexpr int tr_stride() { return R * 2 + 64; }

template <int R, int NT>
__device__ __forceinline__ uint32_t load_rc16(uint4* reg, const bf16_t* __restrict__ slab, int64_t s_k, int64_t row0,
                                              int64_t rows, int64_t k0, int64_t kend, int tid) {
  constexpr int CH = R / 8, KPP = NT / CH, ITS = BK / KPP;
  static_assert(NT % CH == 0 && KPP >= 1 && BK % KPP == 0 && ITS >= 1, "tile rows vs threads");
  const int ch = tid % CH, kr = tid / CH;
  const int64_t nrow64 = rows - row0, krem64 = kend - k0;
  const int nch1 = (int)(nrow64 < R ? nrow64 : R) / 8 - 1;
  const int chc = ch < nch1 ? ch : nch1;
  const int krem = (int)(krem64 < BK ? krem64 : BK);
  const bf16_t* base = slab;
  uint32_t okbits = 0;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int k = kr + KPP * it;
    const bool ok = k < krem;
    const int kk = ok ? k : 0;                      // row k0 of P always exists
    reg[it] = *reinterpret_cast<const uint4*>(base + ((uint32_t)kk * (uint32_t)s_k + 8u * (uint32_t)chc));
    okbits |= ok ? (1u << it) : 0u;
  }
  return okbits;
}

template <int R, int NT>
__device__ __forceinline__ void store_rc16(uint4* reg, uint32_t okbits, unsigned char* lds, int tid) {
  constexpr int CH = R / 8, KPP = NT / CH, ITS = BK / KPP;
  const int ch = tid % CH, kr = tid / CH;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    if (!((okbits >> it) & 1u)) reg[it] = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(lds + (kr + KPP * it) * tr_stride<R>() + 16 * ch) = reg[it];
  }
}

// MFMA operand fragment (row = lane % 32 of a 32-row block, k = 8*(lane/32) .. +7 of a 16-wide k-slab) out of
// the [k][row] image: two transposing reads of 4 k-rows x 16 rows per 16-lane group.  Lane 4q+p of a group
// supplies the address of k-row q, rows 4p .. 4p+3, and receives its own row's element of the 4 k-rows.
// `lane_off` (see tr_lane_off) carries the per-lane part, `uni` the wave-uniform byte offset of the block.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int R>
__device__ __forceinline__ uint32_t tr_lane_off(int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  return (uint32_t)((8 * (g >> 1) + q) * tr_stride<R>() + (16 * (g & 1) + 4 * pp) * 2);
}

template <int R>
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* img, uint32_t lane_off, int uni) {
  const unsigned char* p0 = img + lane_off + uni;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * tr_stride<R>()));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// ---- row-contiguous operand (element (row,k) at P[k*s_k + row]).
// VEC form: thread -> 4 consecutive rows 4*(tid % (R/4)) and KPT = 8*R/NT consecutive k starting at
// KPT*(tid / (R/4)); one 16-byte load per k (a wave reads 1 KiB of one or two k-rows).  Needs s_k and
// the row count to be multiples of 4 and a 16-byte aligned base.  reg[4*e + i] = (row 4*rq + i, k e).
// Scalar form: thread -> row tid % R and 32*R/NT consecutive k, one dword load each.
// Both return the bit mask of the thread's k positions that hold real data.
// `tpos` = k0 mod shiftT and `kbm` = (the thread's first k inside the tile) mod shiftT come from the caller, which
// keeps them up incrementally: everything per k position is a handful of scalar operations (no division, no
// 64-bit products), and the loads are a wave-uniform k-row pointer plus one loop-invariant per-lane offset.
template <int R, int NT>
__device__ __forceinline__ uint32_t load_rc4(float* reg, const float* __restrict__ slab, int64_t s_k, int64_t row0,
                                             int64_t rows, int64_t k0, int64_t kend, int tid, int shiftT, int shift, int tpos,
                                             int kbm) {
  constexpr int RQ = R / 4, KG = NT / RQ, KPT = BK / KG;
  static_assert(NT % RQ == 0 && BK % KG == 0 && KPT >= 2, "tile rows vs threads");
  const int rq = tid % RQ;
  const int kg = (RQ % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / RQ) : tid / RQ;
  const int64_t nrow64 = rows - row0;
  const int nq1 = (int)(nrow64 < R ? nrow64 : R) / 4 - 1;
  const int rqc = rq < nq1 ? rq : nq1;
  const int kb = kg * KPT;
  const int64_t krem64 = kend - k0 - kb;
  const int kl = (int)(krem64 < KPT ? krem64 : KPT);         // <= 0: nothing of this thread's k range is real
  const uint32_t Tw = shiftT > 0 ? (uint32_t)shiftT : 0x7fffffffu;
  int t = tpos + kbm;                                        // position inside the window of shiftT rows
  if (shiftT > 0 && t >= shiftT) t -= shiftT;
  const float* first = slab;                                 // k-row k0 always exists: what invalid positions read
  const float* base = slab + (int64_t)(kb + shift) * s_k;
  const uint32_t voff = 4u * (uint32_t)rqc;
  uint32_t okbits = 0;
#pragma unroll
  for (int e = 0; e < KPT; ++e) {
    const bool ok = (e < kl) & ((uint32_t)(t + shift) < Tw);
    const float* src = ok ? base + e * s_k : first;
    const float4 v = *reinterpret_cast<const float4*>(src + voff);
    reg[4 * e + 0] = v.x; reg[4 * e + 1] = v.y; reg[4 * e + 2] = v.z; reg[4 * e + 3] = v.w;
    okbits |= ok ? (1u << e) : 0u;
    t = (t + 1 == shiftT) ? 0 : t + 1;
  }
  return okbits;
}

// (the thread's first k inside the tile) mod shiftT, for load_rc4
template <int R, int NT>
__device__ __forceinline__ int rc4_kbm(int tid, int shiftT) {
  constexpr int RQ = R / 4, KG = NT / RQ, KPT = BK / KG;
  const int kg = (RQ % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / RQ) : tid / RQ;
  return shiftT > 0 ? (kg * KPT) % shiftT : 0;
}

template <bool BF16, int R, int NT>
__device__ __forceinline__ void store_rc4(float* reg, uint32_t okbits, unsigned char* lds, int tid) {
  constexpr int RQ = R / 4, KG = NT / RQ, KPT = BK / KG;
  const int rq = tid % RQ, kg = tid / RQ;
  if (okbits != (1u << KPT) - 1u) {
#pragma unroll
    for (int e = 0; e < KPT; ++e)
      if (!((okbits >> e) & 1u)) { reg[4 * e + 0] = 0.f; reg[4 * e + 1] = 0.f; reg[4 * e + 2] = 0.f; reg[4 * e + 3] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[KPT];
#pragma unroll
    for (int e = 0; e < KPT; ++e) v[e] = reg[4 * e + i];
    lds_store_run<BF16, KPT>(lds, 4 * rq + i, kg * KPT, v);
  }
}

// bf16 precision: the float4 a lane holds per k is 4 consecutive rows of ONE k, which is exactly 8 bytes of the
// [k][row] image the 16-bit row-contiguous operands use: lanes write consecutive 8-byte pieces of a k-row
// (conflict-free; the [row][k] image would put every lane's rows 256 bytes apart, on the same banks) and the
// fragments come out through the transposing LDS read, with no transposition in registers.
template <int R, int NT>
__device__ __forceinline__ void store_rc4_tr(float* reg, uint32_t okbits, unsigned char* lds, int tid) {
  constexpr int RQ = R / 4, KG = NT / RQ, KPT = BK / KG;
  const int rq = tid % RQ, kg = tid / RQ;
  if (okbits != (1u << KPT) - 1u) {      // zeroed in the registers: the row sums of the caller read them afterwards
#pragma unroll
    for (int e = 0; e < KPT; ++e)
      if (!((okbits >> e) & 1u)) { reg[4 * e + 0] = 0.f; reg[4 * e + 1] = 0.f; reg[4 * e + 2] = 0.f; reg[4 * e + 3] = 0.f; }
  }
#pragma unroll
  for (int e = 0; e < KPT; ++e) {
    uint2 u;
    u.x = pack_bf16x2(reg[4 * e + 0], reg[4 * e + 1]);
    u.y = pack_bf16x2(reg[4 * e + 2], reg[4 * e + 3]);
    *reinterpret_cast<uint2*>(lds + (kg * KPT + e) * tr_stride<R>() + 8 * rq) = u;
  }
}

template <int R, int NT>
__device__ __forceinline__ uint32_t load_rc1(float* reg, const float* __restrict__ slab, int64_t s_k, int64_t row0,
                                             int64_t rows, int64_t k0, int64_t kend, int tid, int shiftT, int shift) {
  constexpr int KPT = BK * R / NT;
  static_assert(NT % R == 0 && R % 64 == 0 && KPT >= 4 && KPT % 4 == 0, "tile rows vs threads");
  const int kq = (NT == R) ? 0 : __builtin_amdgcn_readfirstlane(tid / R);
  const int64_t nrow64 = rows - row0;
  const int nrow1 = (int)(nrow64 < R ? nrow64 : R) - 1;
  int r = tid % R;
  r = r < nrow1 ? r : nrow1;
  const int64_t kb = k0 + kq * KPT;
  int t = shiftT > 0 ? (int)((uint32_t)kb % (uint32_t)shiftT) : 0;
  const int64_t krem64 = kend - kb;
  const int kl = (int)(krem64 < KPT ? krem64 : KPT);
  uint32_t okbits = 0;
#pragma unroll
  for (int e = 0; e < KPT; ++e) {
    bool ok = e < kl;
    if (shiftT > 0) {
      const int ts = t + shift;
      ok = ok && ts >= 0 && ts < shiftT;
      if (++t == shiftT) t = 0;
    }
    const int64_t kidx = ok ? kq * KPT + e + shift : 0;     // relative to k0, which is always a valid row
    reg[e] = (slab + kidx * s_k)[(uint32_t)r];
    okbits |= ok ? (1u << e) : 0u;
  }
  return okbits;
}

template <bool BF16, int R, int NT>
__device__ __forceinline__ void store_rc1(float* reg, uint32_t okbits, unsigned char* lds, int tid) {
  constexpr int KPT = BK * R / NT;
  const int r = tid % R, kq = tid / R;
  if (okbits != (KPT == 32 ? 0xffffffffu : ((1u << (KPT & 31)) - 1u))) {
#pragma unroll
    for (int e = 0; e < KPT; ++e)
      if (!((okbits >> e) & 1u)) reg[e] = 0.f;
  }
  lds_store_run<BF16, KPT>(lds, r, kq * KPT, reg);
}

// Epilogue of one wave tile whose first row / column is (ib, jb): accumulator (block it, jt; register r; lane)
// holds C[ib + 32*it + acc_row32(r, lane)][jb + 32*jt + (lane & 31)].  One store instruction writes two
// 128-byte row segments.  MODE 0: bias + activation; 1: + residual / relu mask; 2: atomic accumulation.
// FULL: the whole wave tile lies inside the matrix (no bounds predicates).
template <int MODE, bool FULL, int TM, int TN>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TM][TN], const GemmParams& p, int64_t ib, int64_t jb, int lane) {
  const int lr = lane & 31, half = lane >> 5;
  const int64_t mrem64 = p.M - ib, nrem64 = p.N - jb;
  const int mrem = (int)(mrem64 < 32 * TM ? mrem64 : 32 * TM), nrem = (int)(nrem64 < 32 * TN ? nrem64 : 32 * TN);
  float* cbase = reinterpret_cast<float*>(p.C) + ib * p.ldc + jb;
  const uint32_t ldc = (uint32_t)p.ldc;
  const uint32_t voff = (uint32_t)(4 * half) * ldc + (uint32_t)lr;
  // absent residual / relu operands alias C: the loads stay unconditional, their values unused
  const bool has_res = p.residual != nullptr, has_relu = p.relu_of != nullptr;
  const float* rbase = has_res ? p.residual + ib * p.ld_res + jb : cbase;
  const float* ubase = has_relu ? p.relu_of + ib * p.ld_relu + jb : cbase;
  const uint32_t ldr = has_res ? (uint32_t)p.ld_res : ldc, ldu = has_relu ? (uint32_t)p.ld_relu : ldc;
  const uint32_t voff_r = (uint32_t)(4 * half) * ldr + (uint32_t)lr, voff_u = (uint32_t)(4 * half) * ldu + (uint32_t)lr;
#pragma unroll
  for (int jt = 0; jt < TN; ++jt) {
    const int col = 32 * jt + lr;
    const bool cok = FULL || col < nrem;
    float bsum = 0.f;
    if constexpr (MODE != 2) {
      const int cc = cok ? col : 0;
      if (p.bias) bsum += p.bias[jb + cc];
      if (p.bias2) bsum += p.bias2[jb + cc];
    }
#pragma unroll
    for (int it = 0; it < TM; ++it) {
      float res[16], rel[16];
      if constexpr (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowl = 32 * it + (r & 3) + 8 * (r >> 2);
          const bool ok = FULL || (cok && rowl + 4 * half < mrem);
          // uniform part in the scalar base, one per-lane offset register for all elements; both parts
          // fall back to offset 0 outside the matrix, so every lane reads an element that exists
          const uint32_t urow = (FULL || rowl < mrem) ? (uint32_t)rowl : 0u, ucol = (FULL || 32 * jt < nrem) ? 32u * jt : 0u;
          res[r] = (rbase + (urow * ldr + ucol))[ok ? voff_r : 0u];
          rel[r] = (ubase + (urow * ldu + ucol))[ok ? voff_u : 0u];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowl = 32 * it + (r & 3) + 8 * (r >> 2);
        const bool ok = FULL || (cok && rowl + 4 * half < mrem);
        float* c = (cbase + ((uint32_t)rowl * ldc + 32u * jt)) + voff;
        float v = acc[it][jt][r];
#ifdef EXP_NOSTORE
        if (v != 12345.678f) continue;
#endif
        if constexpr (MODE == 2) {
          if (p.map_H == 0) {
            if (ok) atomicAdd(c, v);
          } else {
            const int64_t dst = map_row(p, ib + rowl + 4 * half);
            if (ok && dst >= 0) atomicAdd(reinterpret_cast<float*>(p.C) + dst * p.ldc + jb + 32 * jt + lr, v);
          }
        } else {
          v += bsum;
          if constexpr (MODE == 1) {
            if (has_res) v += res[r];
          }
          if (p.act == TAD_ACT_RELU) v = fmaxf(v, 0.f);
          if constexpr (MODE == 1) {
            if (has_relu && !(rel[r] > 0.f)) v = 0.f;
          }
          if (ok) *c = v;
        }
      }
    }
  }
}

// Same epilogue with 16-byte accesses: the wave passes its accumulators through a private 32 x 64
// fp32 staging block in LDS (the tile buffers are free once the k-loop is done) and reads them back
// row-major, 4 consecutive columns per lane, so C, the residual and the relu operand move as 16
// bytes per lane: a quarter of the vector-memory instructions of the dword form.  Needs N, ldc,
// ld_res, ld_relu multiples of 4 and 16-byte aligned bases.  MODE 0 / 1 only (no wide atomics).
template <int MODE, int TM, int TN>
__device__ __forceinline__ void epilogue_wide(const f32x16 (&acc)[TM][TN], const GemmParams& p, int64_t ib, int64_t jb,
                                              int lane, float* stage) {
  static_assert(TN % 2 == 0 && MODE != 2, "column pairs");
  const int lr = lane & 31;
  const int rrow = lane >> 4, c4 = lane & 15;          // read-back: row rrow + 4*pass, columns 4*c4 .. 4*c4+3 of a 64-wide pair
  const int64_t mrem64 = p.M - ib, nrem64 = p.N - jb;
  const int mrem = (int)(mrem64 < 32 * TM ? mrem64 : 32 * TM), nrem = (int)(nrem64 < 32 * TN ? nrem64 : 32 * TN);
  float* cbase = reinterpret_cast<float*>(p.C) + ib * p.ldc + jb;           // fp32 C
  bf16_t* cbase16 = reinterpret_cast<bf16_t*>(p.C) + ib * p.ldc + jb;       // bf16 C (p.c16)
  const uint32_t ldc = (uint32_t)p.ldc;
  const bool has_res = p.residual != nullptr, has_relu = p.relu_of != nullptr;
  const float* rbase = has_res ? p.residual + ib * p.ld_res + jb : cbase;
  const float* ubase = has_relu ? p.relu_of + ib * p.ld_relu + jb : cbase;
  const uint32_t ldr = has_res ? (uint32_t)p.ld_res : ldc, ldu = has_relu ? (uint32_t)p.ld_relu : ldc;
  const uint32_t voff = (uint32_t)rrow * ldc + 4u * c4, voff_r = (uint32_t)rrow * ldr + 4u * c4, voff_u = (uint32_t)rrow * ldu + 4u * c4;
#pragma unroll
  for (int jp = 0; jp < TN / 2; ++jp) {
    const int col = 64 * jp + 4 * c4;
    const bool cok = col < nrem;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const int cc = cok ? col : 0;
      if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + jb + cc); bsum.x += b.x; bsum.y += b.y; bsum.z += b.z; bsum.w += b.w; }
      if (p.bias2) { const float4 b = *reinterpret_cast<const float4*>(p.bias2 + jb + cc); bsum.x += b.x; bsum.y += b.y; bsum.z += b.z; bsum.w += b.w; }
    }
#pragma unroll
    for (int it = 0; it < TM; ++it) {
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[acc_row32(r, lane) * 64 + 32 * j2 + lr] = acc[it][2 * jp + j2][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float4 res[8], rel[8];
      if constexpr (MODE == 1) {
        // an absent operand is not read: its stand-in address (C) is a float32 one only when C is float32
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int rowl = 32 * it + 4 * ps;
          const bool ok = cok && rowl + rrow < mrem;
          const uint32_t urow = rowl < mrem ? (uint32_t)rowl : 0u, ucol = 64 * jp < nrem ? 64u * jp : 0u;
          res[ps] = rel[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (has_res) res[ps] = *reinterpret_cast<const float4*>((rbase + (urow * ldr + ucol)) + (ok ? voff_r : 0u));
          if (has_relu && p.relu16) {
            const uint2 t = *reinterpret_cast<const uint2*>((reinterpret_cast<const bf16_t*>(p.relu_of) + ib * p.ld_relu + jb + (urow * ldu + ucol)) + (ok ? voff_u : 0u));
            rel[ps] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
          } else if (has_relu) {
            rel[ps] = *reinterpret_cast<const float4*>((ubase + (urow * ldu + ucol)) + (ok ? voff_u : 0u));
          }
        }
      }
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int rowl = 32 * it + 4 * ps;
        const bool ok = cok && rowl + rrow < mrem;
        float4 v = *reinterpret_cast<const float4*>(stage + (4 * ps + rrow) * 64 + 4 * c4);
        v.x += bsum.x; v.y += bsum.y; v.z += bsum.z; v.w += bsum.w;
        if constexpr (MODE == 1) {
          if (has_res) { v.x += res[ps].x; v.y += res[ps].y; v.z += res[ps].z; v.w += res[ps].w; }
        }
        if (p.act == TAD_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if constexpr (MODE == 1) {
          if (has_relu) {
            if (!(rel[ps].x > 0.f)) v.x = 0.f;
            if (!(rel[ps].y > 0.f)) v.y = 0.f;
            if (!(rel[ps].z > 0.f)) v.z = 0.f;
            if (!(rel[ps].w > 0.f)) v.w = 0.f;
          }
        }
        if (p.c16) {
          uint2 u;
          u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
          if (ok) *reinterpret_cast<uint2*>((cbase16 + ((uint32_t)rowl * ldc + 64u * jp)) + voff) = u;
          continue;
        }
        if (ok) *reinterpret_cast<float4*>((cbase + ((uint32_t)rowl * ldc + 64u * jp)) + voff) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Staging registers of one operand tile (R rows x 32 k) and the two halves of its pipeline step.
template <bool BF16, int KIND, int R, int NT, bool VEC>
struct Stage {
  static constexpr bool IS16 = KIND == KC16 || KIND == RC16;
  static constexpr int NREG32 = IS16 ? 1 : R * BK / NT;
  static constexpr int NREG16 = IS16 ? R * 4 / NT : 1;
  // [k][row] image read through the transposing LDS read: 16-bit row-contiguous operands, and fp32 ones in bf16 precision
  static constexpr bool TR = KIND == RC16 || (BF16 && VEC && KIND == RC32);
  static constexpr int LDS_BYTES = TR ? BK * tr_stride<R>() : R * (BF16 ? 64 : 128);
  float r32[NREG32];
  uint4 r16[NREG16];
  uint32_t ok = 0;
  int64_t kcur = 0;
  int kbm = 0;

  // once per kernel, before the first load of an operand that is read with a time shift
  __device__ __forceinline__ void init(int tid, int shiftT) {
    if constexpr (KIND == RC32 && VEC) kbm = rc4_kbm<R, NT>(tid, shiftT);
  }

  // `slab` points at element (row0, k0) of the operand
  __device__ __forceinline__ void load(const void* P, int64_t s_row, int64_t s_k, int64_t row0, int64_t rows, int64_t k0,
                                       int64_t kend, int tid, int shiftT, int shift, int tpos) {
    kcur = k0;
    if constexpr (KIND == KC32) load_kc<R, NT, VEC>(r32, reinterpret_cast<const float*>(P), s_row, row0, rows, k0, kend, tid);
    else if constexpr (KIND == RC32 && VEC) ok = load_rc4<R, NT>(r32, reinterpret_cast<const float*>(P), s_k, row0, rows, k0, kend, tid, shiftT, shift, tpos, kbm);
    else if constexpr (KIND == RC32) ok = load_rc1<R, NT>(r32, reinterpret_cast<const float*>(P), s_k, row0, rows, k0, kend, tid, shiftT, shift);
    else if constexpr (KIND == KC16) load_kc16<R, NT>(r16, reinterpret_cast<const bf16_t*>(P), s_row, row0, rows, k0, kend, tid);
    else ok = load_rc16<R, NT>(r16, reinterpret_cast<const bf16_t*>(P), s_k, row0, rows, k0, kend, tid);
  }
  __device__ __forceinline__ void store(unsigned char* lds, int64_t kend, int tid) {
    if constexpr (KIND == KC32) store_kc<BF16, R, NT>(r32, lds, kcur, kend, tid);
    else if constexpr (KIND == RC32 && VEC && BF16) store_rc4_tr<R, NT>(r32, ok, lds, tid);
    else if constexpr (KIND == RC32 && VEC) store_rc4<BF16, R, NT>(r32, ok, lds, tid);
    else if constexpr (KIND == RC32) store_rc1<BF16, R, NT>(r32, ok, lds, tid);
    else if constexpr (KIND == KC16) store_kc16<R, NT>(r16, lds, kcur, kend, tid);
    else store_rc16<R, NT>(r16, ok, lds, tid);
  }
};

template <bool BF16, int AK, int BKD, bool VEC, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? (TN == 4 ? 2 : (BF16 ? 3 : 2)) : 1)) void gemm_kernel(GemmParams p) {
  constexpr int NT = 64 * WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int ROWB = BF16 ? 64 : 128;
  constexpr bool A_RC = AK == RC32 || AK == RC16;
  static_assert(BF16 || (AK < KC16 && BKD < KC16), "16-bit operands: bf16 kernels only");
  using StA = Stage<BF16, AK, BM, NT, VEC>;
  using StB = Stage<BF16, BKD, BN, NT, VEC>;
  constexpr bool A_TR = StA::TR, B_TR = StB::TR;
  constexpr int A_BYTES = StA::LDS_BYTES, B_BYTES = StB::LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: everything derived from it is wave-uniform
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware work assignment: workgroups with equal blockIdx % 8 share an XCD (L2).  The work items
  // (reduction split z, tile), z-major, are dealt out so that each XCD owns a contiguous run (bijective
  // for any count): neighbouring tiles share a row panel, and the tiles of one reduction split share
  // the slabs of both operands, which then come out of that XCD's L2 after the first read.
  int64_t tile, zsplit;
  {
    const int64_t bid = blockIdx.x, nw = p.ntiles * p.nsplits;
    const int64_t xcd = bid % 8, local = bid / 8, q = nw / 8, r = nw % 8;
    const int64_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    zsplit = w / p.ntiles;
    tile = w % p.ntiles;
  }
  const int64_t tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int64_t i0 = tm * BM, j0 = tn * BN;
  const int64_t kbeg = zsplit * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const int nt = (int)((kend - kbeg + BK - 1) / BK);
  if (nt <= 0) return;

  // first row / column of this wave's tile; a wave whose tile lies outside the matrix only helps staging
  const int64_t ib = i0 + 32 * TM * wm, jb = j0 + 32 * TN * wn;
  const bool active = ib < p.M && jb < p.N;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  StA sa;
  StB sb;
  // row-contiguous A only: sums over k of this thread's A rows (RC32: 4 rows in the VEC form, else 1; RC16: 8 rows)
  float rsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_rowsum = A_RC && p.rowsum != nullptr && tn == 0;
  // position of the next k-slab inside its window of shiftT rows: the slabs are loaded in order.  A load past
  // the end of the range re-reads the first slab WITH the first slab's position: the shifted rows it touches
  // must be the ones that exist.
  sb.init(tid, p.shiftT);
  const int tpos_first = p.shiftT > 0 ? (int)((uint64_t)kbeg % (uint32_t)p.shiftT) : 0;
  int tpos = tpos_first;
  const int bk_mod = p.shiftT > 0 ? BK % p.shiftT : 0;
  // The k-slabs are loaded in order; their addresses (element (tile row 0, k) of either operand) and k advance
  // by one slab per real load -- no products in the loop.  real == false: a load past the end of the range,
  // which re-reads the first slab.
  constexpr int ESA = (AK == KC16 || AK == RC16) ? 2 : 4, ESB = (BKD == KC16 || BKD == RC16) ? 2 : 4;
  const unsigned char* const a_first = reinterpret_cast<const unsigned char*>(p.A) + (i0 * p.sa_row + kbeg * p.sa_k) * ESA;
  const unsigned char* const b_first = reinterpret_cast<const unsigned char*>(p.B) + (j0 * p.sb_row + kbeg * p.sb_k) * ESB;
  const int64_t a_step = (int64_t)BK * p.sa_k * ESA, b_step = (int64_t)BK * p.sb_k * ESB;
  const unsigned char* a_next = a_first;
  const unsigned char* b_next = b_first;
  int64_t k_next = kbeg;
  auto gload = [&](bool real) {
    const int64_t k0 = real ? k_next : kbeg;
    sa.load(real ? a_next : a_first, p.sa_row, p.sa_k, i0, p.M, k0, kend, tid, 0, 0, 0);
    sb.load(real ? b_next : b_first, p.sb_row, p.sb_k, j0, p.N, k0, kend, tid, p.shiftT, p.shift, real ? tpos : tpos_first);
    if (real) {
      a_next += a_step; b_next += b_step; k_next += BK;
      tpos += bk_mod;
      if (tpos >= p.shiftT && p.shiftT > 0) tpos -= p.shiftT;
    }
  };
  auto lstore = [&](int buf, bool count) {
    unsigned char* As = smem + (size_t)buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
    sa.store(As, kend, tid);
    sb.store(Bs, kend, tid);
    if constexpr (A_RC) {
      if (want_rowsum && count) {       // the staged registers are already zeroed where k is out of range
        if constexpr (AK == RC16) {
#pragma unroll
          for (int e = 0; e < StA::NREG16; ++e) {
            const uint32_t w[4] = {sa.r16[e].x, sa.r16[e].y, sa.r16[e].z, sa.r16[e].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              rsum[2 * i] += __uint_as_float(w[i] << 16);
              rsum[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
            }
          }
        } else if constexpr (VEC) {
#pragma unroll
          for (int e = 0; e < StA::NREG32 / 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) rsum[i] += sa.r32[4 * e + i];
        } else {
#pragma unroll
          for (int e = 0; e < StA::NREG32; ++e) rsum[0] += sa.r32[e];
        }
      }
    }
  };

  // Pipeline: registers hold the tile of step t+1 while step t is multiplied.  Each step first moves
  // those registers into the idle LDS buffer, then issues the loads of step t+2 into the same
  // registers -- they have a whole step to arrive -- then runs the MFMAs of step t.  Loads are
  // unconditional: beyond the range they re-read its first tile, which nobody looks at.
  STAMP(ts0);
  gload(true);
  lstore(0, true);
  gload(nt > 1);
  __syncthreads();
  STAMP(ts1);
#ifdef EXP_STAMPS
  unsigned long long sA = 0, sB = 0, sC = 0, sD = 0, tprev = ts1;
#endif
  const int half = lane >> 5, lr = lane & 31;
  // per-lane byte offsets of the fragment chunks inside a 32-row block (the swizzle only depends on lr)
  uint32_t foff[BF16 ? 2 : 4];
#pragma unroll
  for (int kk = 0; kk < (BF16 ? 2 : 4); ++kk) foff[kk] = lds_off<BF16>(lr, (BF16 ? 16 : 8) * kk + (BF16 ? 8 : 4) * half);
  const uint32_t tra = tr_lane_off<BM>(lane), trb = tr_lane_off<BN>(lane);

  for (int t = 0; t < nt; ++t) {
    lstore((t + 1) & 1, t + 1 < nt);    // the other buffer: last read one barrier ago
    STAMP(tC);
    gload(t + 2 < nt);
    STAMP(tA);
    const unsigned char* Abuf = smem + (size_t)(t & 1) * (A_BYTES + B_BYTES);
    const unsigned char* Bbuf = Abuf + A_BYTES;
    const unsigned char* As = Abuf + (32 * TM * wm) * ROWB;      // [row][k] images
    const unsigned char* Bs = Bbuf + (32 * TN * wn) * ROWB;
#ifdef EXP_NOMFMA
    if (active && p.K == 12345) {
#else
    if (active) {
#endif
      if constexpr (BF16) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
          bf16x8 a[TM], b[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (A_TR) a[i] = tr_fragment<BM>(Abuf, tra, 16 * kk * tr_stride<BM>() + (32 * TM * wm + 32 * i) * 2);
            else a[i] = *reinterpret_cast<const bf16x8*>(As + 32 * i * ROWB + foff[kk]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (B_TR) b[j] = tr_fragment<BN>(Bbuf, trb, 16 * kk * tr_stride<BN>() + (32 * TN * wn + 32 * j) * 2);
            else b[j] = *reinterpret_cast<const bf16x8*>(Bs + 32 * j * ROWB + foff[kk]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
          f32x4 a[TM], b[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + 32 * i * ROWB + foff[kk]);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + 32 * j * ROWB + foff[kk]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
      }
    }
    STAMP(tB);
    __syncthreads();
    STAMP(tD);
#ifdef EXP_STAMPS
    sC += tC - tprev; sA += tA - tC; sB += tB - tA; sD += tD - tB; tprev = tD;
#endif
  }
  STAMP(ts2);

  if constexpr (A_RC) {
    if (want_rowsum) {     // uniform over the workgroup
      // several threads hold partial sums of one row: combine them through LDS (free after the last
      // barrier of the k-loop), then one atomic per row -- every split of the reduction adds into the
      // same few hundred addresses, so the atomics are kept as few as possible
      constexpr int PARTS = AK == RC16 ? NT / (BM / 8) : (VEC ? NT / (BM / 4) : NT / BM);
      float* red = reinterpret_cast<float*>(smem);           // [PARTS][BM]
      static_assert(PARTS * BM * 4 <= 2 * (A_BYTES + B_BYTES), "reduction scratch fits the tile buffers");
      if constexpr (AK == RC16) {
        const int ch = tid % (BM / 8), kr = tid / (BM / 8);
        *reinterpret_cast<float4*>(red + kr * BM + 8 * ch) = make_float4(rsum[0], rsum[1], rsum[2], rsum[3]);
        *reinterpret_cast<float4*>(red + kr * BM + 8 * ch + 4) = make_float4(rsum[4], rsum[5], rsum[6], rsum[7]);
      } else if constexpr (VEC) {
        const int rq = tid % (BM / 4), kg = tid / (BM / 4);
        *reinterpret_cast<float4*>(red + kg * BM + 4 * rq) = make_float4(rsum[0], rsum[1], rsum[2], rsum[3]);
      } else {
        red[(tid / BM) * BM + tid % BM] = rsum[0];
      }
      __syncthreads();
      if (tid < BM && i0 + tid < p.M) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) sum += red[q * BM + tid];
        const int64_t dst = map_row(p, i0 + tid);
        if (dst >= 0) {
          atomicAdd(&p.rowsum[dst], sum);
          if (p.rowsum2) atomicAdd(&p.rowsum2[dst], sum);
        }
      }
      __syncthreads();
    }
  }

  if (active && !p.atomic && p.wide_epi) {
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    static_assert((NT / 64) * 32 * 64 * 4 <= 2 * (A_BYTES + B_BYTES), "staging blocks fit the tile buffers");
    if (p.residual || p.relu_of) epilogue_wide<1>(acc, p, ib, jb, lane, stage);
    else epilogue_wide<0>(acc, p, ib, jb, lane, stage);
  } else if (active) {
    const bool full = ib + 32 * TM <= p.M && jb + 32 * TN <= p.N;
    if (p.atomic) {
      if (full) epilogue<2, true>(acc, p, ib, jb, lane); else epilogue<2, false>(acc, p, ib, jb, lane);
    } else if (p.residual || p.relu_of) {
      if (full) epilogue<1, true>(acc, p, ib, jb, lane); else epilogue<1, false>(acc, p, ib, jb, lane);
    } else {
      if (full) epilogue<0, true>(acc, p, ib, jb, lane); else epilogue<0, false>(acc, p, ib, jb, lane);
    }
  }
#ifdef EXP_STAMPS
  {
    STAMP(ts3);
    if (tid == 0) {
      atomicAdd(&g_stamps[0], ts1 - ts0); atomicAdd(&g_stamps[1], sA); atomicAdd(&g_stamps[2], sB); atomicAdd(&g_stamps[3], sC);
      atomicAdd(&g_stamps[4], sD); atomicAdd(&g_stamps[5], ts3 - ts2); atomicAdd(&g_stamps[6], ts3 - ts0); atomicAdd(&g_stamps[7], 1ull);
    }
  }
#endif
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// VEC: every operand can be read with 16-byte loads.  fp32 k-contiguous: 16-byte aligned rows and a
// reduction length that is a multiple of 4; fp32 row-contiguous: 16-byte aligned k-rows and a row count
// that is a multiple of 4.  bf16 operands (always 16-byte chunks): multiples of 8 instead of 4.
inline bool operand_vec_ok(int kind, const void* ptr, int64_t s_row, int64_t s_k, int64_t rows, int64_t K) {
  if (!aligned16(ptr)) return false;
  switch (kind) {
    case KC32: return s_row % 4 == 0 && K % 4 == 0;
    case RC32: return s_k % 4 == 0 && rows % 4 == 0;
    case KC16: return s_row % 8 == 0 && K % 8 == 0;
    default:   return s_k % 8 == 0 && rows % 8 == 0;
  }
}

// 16-byte epilogue (and the only one that can write a bf16 C): N, ldc, ld_res, ld_relu multiples of 4, aligned bases
inline bool wide_epilogue_ok(const GemmParams& p) {
  return !p.atomic && p.N % 4 == 0 && p.ldc % 4 == 0 && aligned16(p.C) && (!p.bias || aligned16(p.bias)) &&
         (!p.bias2 || aligned16(p.bias2)) && (!p.residual || (p.ld_res % 4 == 0 && aligned16(p.residual))) &&
         (!p.relu_of || (p.ld_relu % 4 == 0 && aligned16(p.relu_of)));
}

// Launches one tile configuration.  VEC selects the 16-byte loaders of the fp32 operands; 16-bit operands
// have no scalar-load form: the caller has checked operand_vec_ok for them (entry points reject misaligned
// bf16 operands).
template <bool BF16, int AK, int BKD, int WM, int WN, int TM, int TN>
void launch_cfg(GemmParams& p, int splits, hipStream_t s) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  p.tiles_n = cdiv64(p.N, BN);
  p.ntiles = cdiv64(p.M, BM) * p.tiles_n;
  p.nsplits = splits;
  dim3 grid((unsigned)(p.ntiles * splits));
  p.wide_epi = wide_epilogue_ok(p);
  const bool vec = (AK >= KC16 || operand_vec_ok(AK, p.A, p.sa_row, p.sa_k, p.M, p.K)) &&
                   (BKD >= KC16 || operand_vec_ok(BKD, p.B, p.sb_row, p.sb_k, p.N, p.K));
  if (vec) hipLaunchKernelGGL((gemm_kernel<BF16, AK, BKD, true, WM, WN, TM, TN>), grid, dim3(64 * WM * WN), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<BF16, AK, BKD, false, WM, WN, TM, TN>), grid, dim3(64 * WM * WN), 0, s, p);
  // the name as a profiler prints it (operand kinds as their enum values: 0 KC32, 1 RC32, 2 KC16, 3 RC16)
  tad_note_kernel("gemm_kernel<%s, %d, %d, %s, %d, %d, %d, %d>", BF16 ? "true" : "false", AK, BKD, vec ? "true" : "false", WM, WN, TM,
                  TN);
}

enum Cfg { CFG_W, CFG_S, CFG_F, CFG_T, CFG_H };
struct TileDim { int bm, bn; };
inline TileDim tile_dim(Cfg c) {
  switch (c) {
    case CFG_W: return {128, 512};
    case CFG_S: return {256, 256};
    case CFG_F: return {128, 256};
    case CFG_H: return {128, 256};
    default: return {128, 128};
  }
}

// Widest tile that still gives every CU a workgroup or two; split-k launches (min_wgs == 0) fill
// the chip through the split count instead.
inline Cfg pick_cfg(int prec, int64_t M, int64_t N, int64_t min_wgs) {
  // bf16: the square 256 x 256 tile measured fastest for all three products at the bench shapes
  // (98 304 x 1600 x 128..400); the 128 x 512 tile (W) stays selectable for experiments
  Cfg big = prec == TAD_BF16 ? CFG_S : CFG_F;
  static const char* force = getenv("TAD_GEMM_CFG");     // experiments: W, S, H or T for every bf16 launch
  if (force && prec == TAD_BF16) return force[0] == 'W' ? CFG_W : force[0] == 'S' ? CFG_S : force[0] == 'H' ? CFG_H : CFG_T;
  const TileDim d = tile_dim(big);
  // a result of at most 128 columns would leave half of the 256-wide tile's matrix-core work on padding; up to 256
  // columns (one tile column either way) the narrow tile's three workgroups per CU still measured 0-2 % faster,
  // from 432 columns on 2 % slower
  static const char* narrow = getenv("TAD_GEMM_NARROW");  // experiments: widest result that takes the narrow tile (0: none)
  static const int64_t narrow_max = narrow ? atoi(narrow) : 256;
  if (prec == TAD_BF16 && N <= narrow_max) return CFG_T;
  if (cdiv64(M, d.bm) * cdiv64(N, d.bn) >= min_wgs) return big;
  return CFG_T;
}

// bf16 tile configurations of one operand-kind pair
template <int AK, int BKD>
int launch_bf16(Cfg cfg, GemmParams& p, int splits, hipStream_t s) {
  if (cfg == CFG_W) launch_cfg<true, AK, BKD, 2, 4, 2, 4>(p, splits, s);
  else if (cfg == CFG_S) launch_cfg<true, AK, BKD, 4, 2, 2, 4>(p, splits, s);
  else if (cfg == CFG_H) launch_cfg<true, AK, BKD, 2, 2, 2, 4>(p, splits, s);
  else launch_cfg<true, AK, BKD, 2, 2, 2, 2>(p, splits, s);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

// fp32-stored operands in either precision
template <int AK, int BKD>
int launch(int prec, Cfg cfg, GemmParams& p, int splits, hipStream_t s) {
  static_assert(AK < KC16 && BKD < KC16, "fp32-stored operands");
  if (prec == TAD_BF16) return launch_bf16<AK, BKD>(cfg, p, splits, s);
  if (cfg == CFG_F) launch_cfg<false, AK, BKD, 2, 4, 2, 2>(p, splits, s);
  else launch_cfg<false, AK, BKD, 2, 2, 2, 2>(p, splits, s);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

constexpr int64_t LD_MAX = 1 << 22;   // tile-local offsets are 32-bit: 256 rows * ld * 4 bytes must fit
constexpr int64_t MIN_WGS = 384;      // 1.5 workgroups per CU before the narrow tile takes over

// Tile of a weight-gradient product (bf16): small results (the 4H x H recurrent-weight gradients, narrow input
// sides) take the 128 x 128 tile -- it wastes fewer padded columns and three of its workgroups share a CU,
// so their load / multiply / barrier phases overlap; measured 15-25 % faster there, slower on the larger ones.
inline Cfg pick_cfg_bwd_weight(int prec, int64_t rows, int64_t cols) {
  if (prec == TAD_BF16 && !getenv("TAD_GEMM_CFG") && rows * cols <= (int64_t)1 << 18) return CFG_T;
  return pick_cfg(prec, rows, cols, 0);
}

// Splits the long reduction of a weight-gradient product over about one workgroup per CU (three for the
// narrow tile, which fits three to a CU): every split adds its whole tile with float atomics (about
// 1.3 TB/s chip-wide), so more splits than that only add traffic.
inline int plan_splits(GemmParams& p, Cfg cfg) {
  const TileDim d = tile_dim(cfg);
  const int64_t tiles = cdiv64(p.M, d.bm) * cdiv64(p.N, d.bn);
  static const char* wgs = getenv("TAD_GEMM_WGS");       // experiments: workgroups a split-k launch aims for
  int64_t splits = (wgs ? atoi(wgs) : (cfg == CFG_T ? 768 : 256)) / tiles;
  const int64_t ksteps = cdiv64(p.K, BK);
  if (splits > ksteps) splits = ksteps;
  if (splits < 1) splits = 1;
  p.k_per_split = cdiv64(ksteps, splits) * BK;
  return (int)cdiv64(p.K, p.k_per_split);
}

// host copy of a gate map argument ({H, G, slot0..slot3} or NULL) into the launch parameters
inline bool set_gate_map(GemmParams& p, const int32_t* gate_map, int64_t rows) {
  p.map_H = 0;
  if (!gate_map) return true;
  const int H = gate_map[0], G = gate_map[1];
  if (H <= 0 || (G != 3 && G != 4) || rows % (4 * (int64_t)H) != 0) return false;
  for (int s = 0; s < 4; ++s) {
    if (gate_map[2 + s] >= G) return false;
    p.map_slot[s] = gate_map[2 + s];
  }
  p.map_H = H; p.map_G = G;
  return true;
}

}  // namespace
