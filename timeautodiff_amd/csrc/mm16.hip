// Matrix products of the nn.Linear family with BOTH operands stored in HBM as bf16, for gfx950.
//
// The activations of the hot path are tall and skinny ([B*T, 128..1024] with B*T around 1e5) and the weights are
// small (they stay in L2), so these products are bound by how many bytes a CU pulls through its memory path per
// matrix-core FLOP, not by HBM or by the matrix cores.  Two kernels, both without any vector-ALU work on the
// operands (no conversion, no transposition in registers):
//
//   mm16_nt   C[M,N] = act(A[M,K] W[N,K]^T + bias + residual) (. relu mask)          forward / input gradient
//     A workgroup owns a panel of 32*NW rows of A.  Each wave loads ITS 32 rows of A once, straight into
//     registers in MFMA operand layout (16 bytes per lane and 16-deep k-step), and keeps them for the whole
//     kernel.  The workgroup then walks the N columns in chunks of 64: the chunk's rows of W are streamed
//     from L2 into a 4-deep ring of LDS stages (64 rows x 128 k each) by direct-to-LDS loads
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers), three stages ahead of the MFMAs,
//     with counted vmcnt waits and one barrier per stage.  A is read exactly once; per MFMA the CU fills 4x less
//     than a square tiling of the same accumulator budget.  The product is evaluated transposed (W is the MFMA's
//     A operand), so a lane ends up with 4 consecutive columns of one row of C: 8-byte (bf16) or 16-byte (fp32)
//     stores, bias / residual / mask operands read the same way.
//
//   mm16_tn   C[Na,Nb] += A[M,Na]^T B[M,Nb]                                         weight gradient
//     The reduction runs over the rows; both operands are row-major, i.e. "k-major" for the MFMA.  Their
//     64-row slabs go to LDS as they are ([k][column] images, direct-to-LDS loads) and the MFMA fragments come
//     out through gfx950's transposing LDS read (ds_read_b64_tr_b16).  B rows can be read time-shifted inside
//     windows of T rows (the h_{t-1} operand of BPTT): rows that fall outside their window are fetched from a
//     page of zeros instead.  The long reduction is split over workgroups; every split adds its tile with fp32
//     atomics (two 128-byte row segments per wave instruction).  A launch takes a list of jobs that share A, so
//     the gate gradients of a recurrent layer are read once from HBM for dW_ih and dW_hh.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include <algorithm>
#include <vector>

#ifndef MM16_ST_AUX
#define MM16_ST_AUX 0
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
typedef __attribute__((address_space(1))) unsigned char gbyte_t;     // explicit global address space: no flat accesses
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(1))) u32x4_t g_u32x4_t;
typedef __attribute__((address_space(1))) u32x2_t g_u32x2_t;
typedef __attribute__((address_space(1))) f32x4 g_f32x4_t;


__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_uniform) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)lds_uniform, 16, 0, 0);
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
// two fp32 -> one dword of two bf16 (round to nearest even), one instruction
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a compile-time constant
template <typename F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifdef MM16_STAMPS
__device__ unsigned long long g_mm16_stamps[8];
__device__ unsigned long long g_tn_stamps[32];
#define TSTAMP(i) asm volatile("s_memtime %0" : "=s"(ts_[i]) :: "memory")
#define MSTAMP(var) __builtin_amdgcn_sched_barrier(0); unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#else
#define MSTAMP(var)
#define TSTAMP(i)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------- mm16_nt
// dynamic-LDS limit of a kernel; the attribute belongs to the (kernel, device) pair, so it is remembered per device
inline int mm16_set_lds(const void* kernel, size_t bytes, size_t (&per_device)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { tad_set_error("mm16: hipGetDevice failed or device index beyond 63"); return TAD_ERR_LAUNCH; }
  if (bytes <= per_device[dev]) return TAD_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { tad_set_error("mm16: hipFuncSetAttribute(%zu bytes LDS) failed: %s", bytes, hipGetErrorString(e)); return TAD_ERR_LAUNCH; }
  per_device[dev] = bytes;
  return TAD_OK;
}

struct NtParams {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  void* C; int64_t ldc; int c16;
  void* Z; int64_t ldz; int z32;           // optional: pre-activation values (before `act`), bf16 (fp32 with act = SILU_Z32)
  const float* bias;
  const void* res; int64_t ld_res; int res16;
  const void* relu; int64_t ld_relu; int relu16;
  int64_t M, N, K;
  int act;                                 // TAD_ACT_NONE / RELU / SILU
  int nchunks, nsplit, cps;                // 128-column chunks of N, splits of the chunk range, chunks per split
  int nks_img;                             // 128-deep slabs the weight image holds: ceil(K / 128)
  int64_t nitems;                          // row panels * nsplit
  int stagger;                             // start delay of every other workgroup, in units of 4096 cycles
  int dbg;                                 // experiments (TAD_MM16_DBG): 1 = no epilogue, 2 = no MFMAs, 4 = no ring loads
};

constexpr int NT_BN = 64;                  // columns of C per chunk
constexpr int NT_KD = 128;                 // k-depth of a ring stage (8 MFMA k-steps)
constexpr int NT_ST = 3;                   // ring stages
constexpr int NT_STAGE_BYTES = NT_BN * NT_KD * 2;
constexpr int NT_NW = 4;                   // waves per workgroup (128 rows of A)

// chunk epilogues among the two slabs before slab s of a chunk of NKS slabs (an epilogue follows the last slab)
constexpr int epilogues_in_window(int s, int NKS) {
  int n = 0;
  for (int j = 1; j <= 2; ++j)
    if ((((s + 1 - j) % NKS) + NKS) % NKS == 0) ++n;
  return n;
}

// Two workgroups of 4 waves share a CU (one when the rows of A need more than 256 registers together with the
// accumulators): their barriers are independent, so one's epilogue (HBM stores) runs under the other's MFMAs.
template <int KS16>
__global__ __launch_bounds__(64 * NT_NW, KS16 <= 36 ? 2 : 1) void mm16_nt_kernel(NtParams p) {
  constexpr int NW = NT_NW;
  constexpr int NKS = (KS16 + 7) / 8;      // 128-deep slabs of W per chunk
  constexpr int PW = 16 / NW;              // direct-to-LDS instructions per wave and stage
  constexpr int BM = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  float* bias_s = reinterpret_cast<float*>(smem + NT_ST * NT_STAGE_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // work item: workgroups with equal blockIdx % 8 share an XCD; each XCD takes a contiguous run of items, so the
  // splits of one row panel (which read the same rows of A) meet in one L2
  int64_t item;
  {
    const int64_t bid = blockIdx.x, nw = p.nitems;
    const int64_t xcd = bid % 8, local = bid / 8, q = nw / 8, rem = nw % 8;
    item = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int64_t panel = item / p.nsplit;
  const int split = (int)(item % p.nsplit);
  const int c_first = split * p.cps;
  int c_count = p.nchunks - c_first;
  if (c_count > p.cps) c_count = p.cps;
  if (c_count <= 0) return;

  // bias of this item's column range -> LDS (zeros without a bias); visible after the barrier below
  {
    const int n0 = c_first * NT_BN, nn = c_count * NT_BN;
    for (int i = tid; i < nn; i += 64 * NW) {
      const int n = n0 + i;
      bias_s[i] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
  }
  __syncthreads();

  // ---- W ring.  Stage image: [64 rows][16 chunks of 16 B] (256-byte rows), chunk c of row w at position
  // c ^ (w & 15) (conflict-free 16-byte fragment reads).  Instruction i of a stage fills rows 4i .. 4i+3; lane l
  // fills row 4i + l/16, position l%16 -- i.e. it fetches chunk (l%16) ^ (row & 15) of that row.  The image of W
  // is padded (rows to a multiple of 64, k to a multiple of 128), so a load needs no bounds: its address is a
  // per-lane constant plus a scalar offset that depends on the slab only.
  uint32_t woff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int i = wave * PW + j;
    const int wr = 4 * i + (lane >> 4);
    woff[j] = 2u * ((uint32_t)wr * (uint32_t)p.ldw + 8u * (uint32_t)((lane & 15) ^ (wr & 15)));
  }
  const int total = c_count * NKS;         // slabs this workgroup consumes
  const int Kk = (int)p.K;
  const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(p.W);
  const int64_t chunk_bytes = 2ll * NT_BN * p.ldw;
  auto issue = [&](int q, int stage) {     // slab q of the item (beyond the end: the last slab again, keeps the counts uniform)
    const int qq = q < total ? q : total - 1;
    const int c = qq / NKS;
    int s = qq - c * NKS;
    if (s >= p.nks_img) s = p.nks_img - 1;   // the image is ceil(K / 128) slabs wide; a slab beyond meets zero rows of A: any finite values do
    unsigned char* st = ring + stage * NT_STAGE_BYTES + wave * PW * 1024;
    const unsigned char* src = wbase + (int64_t)(c_first + c) * chunk_bytes + 2 * NT_KD * s;
#pragma unroll
    for (int j = 0; j < PW; ++j) glds16(src + woff[j], st + j * 1024);
  };
  issue(0, 0); issue(1, 1);
  // Workgroups are dispatched round-robin over the 8 XCDs and, inside an XCD, over its 32 CUs before a CU gets its
  // second workgroup.  Two workgroups that start together on one CU would run their MFMA and store phases in step;
  // the second one of a CU (by that dispatch pattern: a guess that costs nothing when wrong) starts half a chunk late.
  if (p.stagger && (((blockIdx.x >> 3) >> 5) & 1)) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }

  // ---- this wave's 32 rows of A, all of K, in registers (operand layout: row = lane % 32, 8 k at 16*kk + 8*(lane/32))
  const int64_t row = panel * BM + wave * 32 + r;
  const bool row_ok = row < p.M;
  bf16x8 areg[KS16];
  {
    const gbyte_t* arow = (const gbyte_t*)p.A + 2 * ((row_ok ? row : 0) * p.lda);
#pragma unroll
    for (int kk = 0; kk < KS16; ++kk) {
      const int k = 16 * kk + 8 * h;
      const bool kok = k < Kk;               // beyond K the operand must read as zero (rows beyond M feed rows of C nobody stores)
      const u32x4_t t = *reinterpret_cast<const g_u32x4_t*>(arow + 2 * (kok ? k : 0));
      areg[kk] = __builtin_bit_cast(bf16x8, kok ? t : u32x4_t{0u, 0u, 0u, 0u});
    }
    // the loads complete here (and not at their first use inside the pipelined loop, where waiting for them would
    // drain the ring's loads as well)
#pragma unroll
    for (int kk = 0; kk < KS16; ++kk) asm volatile("" : "+v"(areg[kk]));
  }

  // fragment read offsets inside a stage: row 32*blk + r, chunk (2*kq + h) ^ (r & 15)
  uint32_t foff[8];
#pragma unroll
  for (int kq = 0; kq < 8; ++kq) foff[kq] = (uint32_t)(r * 256 + (((2 * kq + h) ^ (r & 15)) << 4));

  // Epilogue in the accumulators' own layout: lane (r, h) holds row `row` of C and, per 32-column block, the four
  // column quads 8g + 4h .. + 3 (g = 0..3).  fp32 C: one 16-byte store per quad.  bf16 C: the two half-waves exchange
  // quads (v_permlane32_swap) so that a lane ends up with 8 consecutive columns: one 16-byte store per quad pair.
  // Either way a store instruction writes 32 contiguous bytes of each of the wave's 32 rows, and the four (bf16) or
  // eight (fp32) stores of a chunk complete 128 / 256 contiguous bytes per row.  All [M,N] operands are addressed
  // through buffer descriptors: an access outside the operand (it is given an offset beyond the descriptor's range)
  // is dropped by the hardware, so every wave issues every store and the
  // counted vmcnt waits of the pipeline stay exact.
  constexpr uint32_t OOB = 0xfffffff0u;
  const uint32_t esz = p.c16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(p.Z, 0, (int)(uint32_t)(p.Z ? p.M * p.ldz * (p.z32 ? 4 : 2) : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (int)(uint32_t)(p.res ? p.M * p.ld_res * (p.res16 ? 2 : 4) : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.relu), 0, (int)(uint32_t)(p.relu ? p.M * p.ld_relu * (p.relu16 ? 2 : 4) : 0), 0x00020000);
  // byte offsets of this lane's row
  const int64_t rowc = row_ok ? row : 0;
  const uint32_t c_row = (uint32_t)(rowc * p.ldc) * esz, z_row = (uint32_t)(rowc * p.ldz) * (p.z32 ? 4u : 2u);
  const uint32_t r_row = (uint32_t)(rowc * p.ld_res) * (p.res16 ? 2u : 4u), u_row = (uint32_t)(rowc * p.ld_relu) * (p.relu16 ? 2u : 4u);

  // 4 consecutive values of a bf16 (e16) or fp32 [M,N] operand at byte offset off (zeros outside the operand)
  auto load4 = [&](const __amdgpu_buffer_rsrc_t& rs, bool e16, uint32_t off, float* m) {
    if (e16) {
      const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      m[0] = bf_lo(t[0]); m[1] = bf_hi(t[0]); m[2] = bf_lo(t[1]); m[3] = bf_hi(t[1]);
    } else {
      const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      m[0] = __uint_as_float(t[0]); m[1] = __uint_as_float(t[1]); m[2] = __uint_as_float(t[2]); m[3] = __uint_as_float(t[3]);
    }
  };

  int q = 0, sidx = 0;                     // slab counter, ring stage of slab q
#ifdef MM16_STAMPS
  unsigned long long tW = 0, tB = 0, tM = 0, tE = 0;
  MSTAMP(t_begin);
#endif
  for (int c = 0; c < c_count; ++c) {
    // the accumulators start from the bias of their columns (read in assembly: the compiler would hold an LDS read
    // back until every direct-to-LDS load in flight -- the next slabs of the ring -- has landed)
    f32x16 acc[2];
    {
      const uint32_t ba = (uint32_t)(uintptr_t)(bias_s + c * NT_BN + 4 * h);     // LDS byte address (low 32 bits of the generic pointer)
      f32x4 bq[8];
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
                   "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[4]), "=&v"(bq[5]), "=&v"(bq[6]), "=&v"(bq[7]) : "v"(ba) : "memory");
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[b][4 * g + j] = bq[4 * b + g][j];
    }
    static_for<NKS>([&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
      // The loads of slab q have landed once at most the operations issued after them are outstanding: the loads of
      // slab q+1, and the stores of every chunk epilogue that ran since slab q's loads were issued (at the start of
      // slab q-2) -- every wave issues exactly 4 (bf16 C) or 8 (fp32 C) stores per epilogue.
      constexpr int EPI = epilogues_in_window(s, NKS);
      MSTAMP(m0_);
      if (EPI == 0 || q < 2) wait_vm<PW>();
      else if (p.c16) wait_vm<PW + 4 * EPI>();
      else wait_vm<PW + 8 * EPI>();
      MSTAMP(m1_);
      __builtin_amdgcn_s_barrier();        // everybody's part of slab q is in LDS; everybody is done with slab q-1
      MSTAMP(m2_);
      if (!(p.dbg & 4)) issue(q + 2, sidx == 0 ? 2 : sidx - 1);      // into the stage slab q-1 occupied
      const unsigned char* st = ring + sidx * NT_STAGE_BYTES;
      // Fragments are fetched two 16-deep steps (a group of 4) at a time into two register groups; the reads of group
      // g+2 are issued right after the MFMAs of group g, so they have the four MFMAs of group g+1 to arrive.
      bf16x8 wf[2][4];
      using STag = decltype(s_tag);
      auto fetch = [&](auto kp_tag, bf16x8 (&w)[4]) {        // steps 2*kp and 2*kp+1 of the slab
        constexpr int k0 = 8 * STag::value + 2 * decltype(kp_tag)::value;
        if constexpr (k0 < KS16) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int b = 0; b < 2; ++b) w[2 * e + b] = *reinterpret_cast<const bf16x8*>(st + b * 32 * 256 + foff[2 * decltype(kp_tag)::value + e]);
        }
      };
      auto mma = [&](auto kp_tag, const bf16x8 (&w)[4]) {
        constexpr int k0 = 8 * STag::value + 2 * decltype(kp_tag)::value;
        if constexpr (k0 < KS16) {
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[b], areg[k0], acc[b], 0, 0, 0);
        }
        if constexpr (k0 + 1 < KS16) {
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2 + b], areg[k0 + 1 < KS16 ? k0 + 1 : 0], acc[b], 0, 0, 0);
        }
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      if (!(p.dbg & 2)) {
        fetch(I0{}, wf[0]); fetch(I1{}, wf[1]);
        mma(I0{}, wf[0]); fetch(I2{}, wf[0]);
        mma(I1{}, wf[1]); fetch(I3{}, wf[1]);
        mma(I2{}, wf[0]);
        mma(I3{}, wf[1]);
        // the same order for the instruction scheduler
        __builtin_amdgcn_sched_group_barrier(0x020, PW, 0);     // VMEM read (direct-to-LDS)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
#ifdef MM16_STAMPS
      { MSTAMP(m3_); tW += m1_ - m0_; tB += m2_ - m1_; tM += m3_ - m2_; }
#endif
      ++q;
      sidx = sidx == 2 ? 0 : sidx + 1;
    });
    MSTAMP(e0_);
    // ---- epilogue of the chunk: 32 rows x 64 columns per wave, one 32-column block at a time through its staging block
    if (p.dbg & 1) { if (acc[0][0] == 123.456f) reinterpret_cast<float*>(p.C)[0] = acc[1][3]; continue; }
    const int nc = (c_first + c) * NT_BN;
    const bool simple = !p.res && !p.Z && !p.relu && p.act == TAD_ACT_NONE;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      uint32_t pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nc + 32 * bb + 8 * g + 4 * h;      // first column of the quad; N is a multiple of 8: a quad is inside or outside
        const bool nok = row_ok && n < (int)p.N && !(p.dbg & 8);
        float v[4] = {acc[bb][4 * g + 0], acc[bb][4 * g + 1], acc[bb][4 * g + 2], acc[bb][4 * g + 3]};
        if (!simple) {
          if (p.res) {
            float m[4];
            load4(r_rs, p.res16, nok ? r_row + (uint32_t)n * (p.res16 ? 2u : 4u) : OOB, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += m[j];
          }
          if (p.Z) {
            if (p.z32) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                              z_rs, nok ? z_row + (uint32_t)n * 4u : OOB, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])}, z_rs, nok ? z_row + (uint32_t)n * 2u : OOB, 0, 0);
          }
          if (p.act == TAD_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          } else if (p.act == TAD_ACT_SILU || p.act == TAD_ACT_SILU_Z32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] * sigmoid_acc(v[j]);
          }
          if (p.relu) {
            float m[4];
            load4(u_rs, p.relu16, nok ? u_row + (uint32_t)n * (p.relu16 ? 2u : 4u) : OOB, m);
            if (p.act == TAD_ACT_DSILU) {
              // backward of a SiLU whose pre-activations the operand holds (the arithmetic of silu_bwd_kernel, with the
              // sigmoid on the hardware exp / reciprocal units: 2e-7 absolute, far below the bf16 operands of this product)
#pragma unroll
              for (int j = 0; j < 4; ++j) { const float s = fast_sigmoid(m[j]); v[j] = v[j] * s * (1.f + m[j] * (1.f - s)); }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) if (!(m[j] > 0.f)) v[j] = 0.f;
            }
          }
        }
        if (p.c16) {
          pk[g][0] = cvt_pk_bf16(v[0], v[1]); pk[g][1] = cvt_pk_bf16(v[2], v[3]);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                 c_rs, nok ? c_row + (uint32_t)n * 4u : OOB, 0, MM16_ST_AUX);
        }
      }
      if (p.c16) {
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          // quads 2P (columns 16P + 4h ..) and 2P+1 (columns 16P + 8 + 4h ..): after the exchange the lower half-wave
          // holds columns 16P .. 16P+7 of its row and the upper one columns 16P+8 .. 16P+15
          uint32_t o[4];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto sw = __builtin_amdgcn_permlane32_swap(pk[2 * P][d], pk[2 * P + 1][d], false, false);
            o[d] = sw[0]; o[2 + d] = sw[1];
          }
          const int n = nc + 32 * bb + 16 * P + 8 * h;
          const bool nok = row_ok && n < (int)p.N && !(p.dbg & 8);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, c_rs, nok ? c_row + (uint32_t)n * 2u : OOB, 0, MM16_ST_AUX);
        }
      }
    }
#ifdef MM16_STAMPS
    { MSTAMP(e1_); tE += e1_ - e0_; }
#endif
  }
#ifdef MM16_STAMPS
  {
    MSTAMP(t_end);
    if (lane == 0 && wave == 0) {
      atomicAdd(&g_mm16_stamps[0], tW); atomicAdd(&g_mm16_stamps[1], tB); atomicAdd(&g_mm16_stamps[2], tM); atomicAdd(&g_mm16_stamps[3], tE);
      atomicAdd(&g_mm16_stamps[4], t_end - t_begin); atomicAdd(&g_mm16_stamps[5], 1ull);
    }
  }
#endif
  wait_vm<0>();
}

#ifdef MM16_STAMPS
}  // namespace
extern "C" int tad_dbg_tn_stamps(unsigned long long* out32, int reset) {
  if (out32) hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tn_stamps), 256);
  if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_tn_stamps), z, 256); }
  return 0;
}
extern "C" int tad_dbg_mm16_stamps(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mm16_stamps), 64);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_mm16_stamps), z, 64); }
  return 0;
}
namespace {
#endif

// Measured at 98 304 rows: one item per panel beats two for N = 128 (14.4 vs 17.4 us) and N = 400, K = 200 (35.9 vs 41.6 us),
// equals it at N = K = 432, loses from N = 1600 on: an item costs about two chunk-times before its first chunk.
inline double nt_item_overhead() {
  static const char* e = getenv("TAD_MM16_NT_OVERHEAD");
  static const double v = e ? atof(e) : 2.0;
  return v;
}

template <int KS16>
int launch_nt(NtParams& p, hipStream_t s) {
  constexpr int NW = NT_NW, BM = 32 * NW;
  const int64_t panels = cdiv64(p.M, BM);
  p.nchunks = (int)cdiv64(p.N, NT_BN);
  // splits of the column range: enough work items to fill the chip, a whole number of rounds where possible
  int best = 1;
  double best_cost = 1e30;
  const int slots = 256 * (KS16 <= 36 ? 2 : 1);
  for (int ns = 1; ns <= p.nchunks && ns <= 32; ++ns) {
    const int cps = (p.nchunks + ns - 1) / ns;
    const int64_t items = panels * ns;
    const double rounds = (double)cdiv64(items, slots);
    const double cost = rounds * (cps + nt_item_overhead());     // chunk-times of an item before its first chunk: loading the A panel, the pipeline fill
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
  }
  static const char* force = getenv("TAD_MM16_NSPLIT");
  if (force && atoi(force) > 0) best = atoi(force) < p.nchunks ? atoi(force) : p.nchunks;
  p.cps = (p.nchunks + best - 1) / best;
  p.nsplit = (p.nchunks + p.cps - 1) / p.cps;
  p.nitems = panels * p.nsplit;
  static const char* dbg = getenv("TAD_MM16_DBG");
  p.dbg = dbg ? atoi(dbg) : 0;
  static const char* stg_env = getenv("TAD_MM16_STAGGER");
  p.stagger = stg_env ? atoi(stg_env) : 0;
  const size_t lds = (size_t)NT_ST * NT_STAGE_BYTES + (size_t)p.cps * NT_BN * 4;
  auto k = mm16_nt_kernel<KS16>;
  static size_t configured[64] = {};
  if (int rc = mm16_set_lds(reinterpret_cast<const void*>(k), lds, configured)) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)p.nitems), dim3(64 * NW), lds, s, p);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("mm16_nt_kernel<%d>", KS16);
  return TAD_OK;
}

// ---------------------------------------------------------------------------------------------- mm16_tn
// Tile: 256 columns of A x 256 columns of B per workgroup of 8 waves, 4 along A x 2 along B: a wave owns 64 x 128 of
// the tile (2 x 4 MFMA blocks) and skips the 32-column blocks of B beyond the operand.  For the recurrent layers
// (B = x or h_{t-1}, at most 256 wide per tile) a column of the gate gradients is fetched into LDS by one workgroup
// only.  Stage = 32 reduction rows: A image [32][256] + B image [32][256] bf16 (32 KiB); 5 stages, three to four in flight.
constexpr int TN_MAXJOBS = 12;
constexpr int TN_BK = 32;                      // reduction rows per stage
constexpr int TN_TA = 256, TN_TB = 256;        // tile: columns of A, columns of B
constexpr int TN_IMG = TN_BK * 512;            // bytes of one image: [32 rows][256 columns] bf16
constexpr int TN_STAGE = 2 * TN_IMG;
constexpr int TN_ST = 5;

struct TnJob {
  const bf16_t* A; int64_t lda;                            // this job's A operand [M, lda]
  const bf16_t* B; int64_t ldb; int nb; int shift;       // B [M, nb]; rows read at m + shift inside windows of T rows
  float* C; int64_t ldc;                                   // C [na', nb] += ..., na' = rows after the gate map
  float* db; float* db2;                                   // += column sums of A (NULL: none)
  int a0, na, na_out;                                      // columns a0 .. a0 + na of A are multiplied (na a multiple of 8); the first na_out produce rows of C
  int mapH, mapG, slot[4];                                 // gate map of the C rows (mapH == 0: identity)
  int tiles_a, tiles_b, tile_base;
};
constexpr int TN_MAXITEMS = 320;               // work items of a launch whose row ranges are sized tile by tile
struct TnParams {
  int64_t M; int T;
  TnJob job[TN_MAXJOBS]; int njobs;
  int tiles_total, nsplits; int64_t rows_per_split, nitems; int dbg;
  // balanced = 1: item w is item[w] = tile | split << 10 | (reduction splits of that tile) << 21.  A tile at the edge of
  // a job multiplies fewer 32-column blocks than a full one and takes less time per row: it gets fewer, longer splits, so
  // that all items of the launch end together (with the same number of splits for every tile the launch took the time
  // of its full tiles while the workgroups of the edge tiles idled)
  int balanced;
  uint32_t item[TN_MAXITEMS];
};

static_assert(sizeof(TnParams) <= 4096, "kernel arguments");

__device__ __attribute__((aligned(256))) uint4 g_mm16_zero16[4];      // 16 zero bytes: source of loads that must read as zero

// row of C (and db) a product row lands in: gate-interleaved index blk*4*H + 4*u + s -> blk*G*H + slot[s]*H + u, -1 = dropped
__device__ __forceinline__ int tn_map_row(const TnJob& j, int n) {
  if (j.mapH == 0) return n;
  const uint32_t s = (uint32_t)n & 3u, q = (uint32_t)n >> 2;
  const uint32_t blk = q / (uint32_t)j.mapH, u = q % (uint32_t)j.mapH;
  const int g = s == 0 ? j.slot[0] : s == 1 ? j.slot[1] : s == 2 ? j.slot[2] : j.slot[3];
  return g < 0 ? -1 : (int)(blk * j.mapG * j.mapH + (uint32_t)g * j.mapH + u);
}

struct TnFrags { s16x4_t a[2][2]; s16x4_t b[4][2]; };
// the 12 transposing reads of one 16-deep step: A blocks at aa[0..1], B blocks at ab[0..3]; O = byte offset of the step
template <int O> __device__ __forceinline__ void tn_fetch(const uint32_t (&aa)[2], const uint32_t (&ab)[4], TnFrags& f) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %12 offset:%18\n\tds_read_b64_tr_b16 %1, %12 offset:%19\n\t"
      "ds_read_b64_tr_b16 %4, %14 offset:%18\n\tds_read_b64_tr_b16 %5, %14 offset:%19\n\t"
      "ds_read_b64_tr_b16 %6, %15 offset:%18\n\tds_read_b64_tr_b16 %7, %15 offset:%19\n\t"
      "ds_read_b64_tr_b16 %2, %13 offset:%18\n\tds_read_b64_tr_b16 %3, %13 offset:%19\n\t"
      "ds_read_b64_tr_b16 %8, %16 offset:%18\n\tds_read_b64_tr_b16 %9, %16 offset:%19\n\t"
      "ds_read_b64_tr_b16 %10, %17 offset:%18\n\tds_read_b64_tr_b16 %11, %17 offset:%19"
      : "=&v"(f.a[0][0]), "=&v"(f.a[0][1]), "=&v"(f.a[1][0]), "=&v"(f.a[1][1]), "=&v"(f.b[0][0]), "=&v"(f.b[0][1]), "=&v"(f.b[1][0]), "=&v"(f.b[1][1]),
        "=&v"(f.b[2][0]), "=&v"(f.b[2][1]), "=&v"(f.b[3][0]), "=&v"(f.b[3][1])
      : "v"(aa[0]), "v"(aa[1]), "v"(ab[0]), "v"(ab[1]), "v"(ab[2]), "v"(ab[3]), "n"(O), "n"(O + 4 * 512)
      : "memory");
}
// the reads of f have returned; later uses of f are ordered after the wait through the operand list
template <int LEFT> __device__ __forceinline__ void tn_landed(TnFrags& f) {      // LEFT = LDS reads issued after those of f
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]),
               "+v"(f.b[2][0]), "+v"(f.b[2][1]), "+v"(f.b[3][0]), "+v"(f.b[3][1]) : "n"(LEFT));
}

__global__ __launch_bounds__(512, 2) void mm16_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // work item -> (reduction split, tile): split-major, dealt to the XCDs in contiguous runs, so the workgroups of an
  // XCD work on the tiles of one (or two) row ranges and fetch those rows of A and B into their L2 once
  int64_t w;
  {
    const int64_t bid = blockIdx.x, nw = p.nitems;
    const int64_t xcd = bid % 8, local = bid / 8, q = nw / 8, rem = nw % 8;
    w = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  int split, t;
  int64_t rows_per_split = p.rows_per_split;
  if (p.balanced) {
    const uint32_t e = p.item[w];
    t = (int)(e & 1023u); split = (int)((e >> 10) & 2047u);
    const int64_t slabs = (p.M + TN_BK - 1) / TN_BK, n = (int64_t)(e >> 21);
    rows_per_split = (slabs + n - 1) / n * TN_BK;
  } else {
    split = (int)(w / p.tiles_total);
    t = (int)(w % p.tiles_total);
  }
  int ji = 0;
#pragma unroll
  for (int k = 1; k < TN_MAXJOBS; ++k)
    if (k < p.njobs && t >= p.job[k].tile_base) ji = k;
  const TnJob& J = p.job[ji];
  const int tl = t - J.tile_base;
  const int ta = tl / J.tiles_b, tb = tl % J.tiles_b;
  const int a_c0 = J.a0 + TN_TA * ta, a_end = J.a0 + J.na;    // first column of the A tile, end of the job's A range
  const int b_c0 = TN_TB * tb, b_end = J.nb;
  const int c_end = J.nb < (int)J.ldc ? J.nb : (int)J.ldc;     // columns of B beyond a narrower C are padding
  const int wa = wave & 3, wn = wave >> 2;                     // the two waves of a SIMD share their A columns
  int nblk = (b_end - b_c0 + 31) / 32 - 4 * wn;                // 32-column blocks of B this wave multiplies
  nblk = nblk < 0 ? 0 : nblk > 4 ? 4 : nblk;
  // 32-column blocks of A: block wa and block wa + 4 of the tile's eight, so that the blocks of an edge tile spread over
  // the four SIMDs (with blocks 2 wa, 2 wa + 1 an edge tile of 64 columns kept one SIMD as busy as a full tile does)
  int nablk = ((a_end - a_c0 + 31) / 32 - wa + 3) / 4;
  nablk = nablk < 0 ? 0 : nablk > 2 ? 2 : nablk;
  const int64_t m_beg = (int64_t)split * rows_per_split;
  int64_t m_end = m_beg + rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nslab = (int)((m_end - m_beg + TN_BK - 1) / TN_BK);
  if (nslab <= 0) return;

  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_mm16_zero16);
  asm volatile("" : "+s"(zero));

  // ---- fill: an image row is 512 bytes (32 chunks of 16); instruction i of an image covers rows 2i, 2i+1 (1 KiB);
  // lane l -> row 2i + l/32, position l%32, which holds chunk (l%32) ^ 4*(row & 3) of the row: a 32-column block of
  // four consecutive rows is spread over all banks, as the transposing fragment reads need.  A wave issues
  // instructions i = 2*wave, 2*wave+1 of both images.
  const int fsub = lane >> 5;                                      // row inside the instruction's pair
  int ca[2], cb[2];                                                 // first column each lane fetches, per instruction
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int frow = 2 * (2 * wave + j) + fsub;                    // row of the stage: 0..31
    const int fch = (lane & 31) ^ (4 * (frow & 3));
    ca[j] = a_c0 + 8 * fch; if (ca[j] > a_end - 8) ca[j] = a_end - 8;     // beyond the range: any readable column (feeds rows / columns nobody stores)
    cb[j] = b_c0 + 8 * fch; if (cb[j] > b_end - 8) cb[j] = b_end - 8;
  }
  // Source addresses: one 64-bit pointer per lane and piece, advanced by a slab per issue (the slabs are issued in
  // order); everything the loop needs from the job lives in registers from here on (a field of p.job[ji] read inside
  // the loop is a scalar load from the kernel arguments, and the wait for it also waits for every LDS read in flight).
  const int T = p.T, shift = J.shift;
  const int lda = __builtin_amdgcn_readfirstlane((int)J.lda), ldb = __builtin_amdgcn_readfirstlane((int)J.ldb);
  const int nrows = (int)(m_end - m_beg);                           // rows of this split
  const int64_t step_a = (int64_t)TN_BK * lda * 2, step_b = (int64_t)TN_BK * ldb * 2;
  uint64_t pa[2], pb[2];
  int rrow[2], tpos[2];                                             // row inside the split / time position of the lane's pieces
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    rrow[j] = 2 * (2 * wave + j) + fsub;
    const int64_t m = m_beg + rrow[j];
    pa[j] = (uint64_t)(uintptr_t)J.A + (uint64_t)(2 * (m * lda + ca[j]));
    pb[j] = (uint64_t)(uintptr_t)J.B + (uint64_t)(2 * ((m + shift) * ldb + cb[j]));      // only dereferenced for rows inside their window
    tpos[j] = T > 0 ? (int)(m % T) : 0;
  }
  const int tstep = T > 0 ? TN_BK % T : 0;
  const uint64_t zero64 = (uint64_t)(uintptr_t)zero;
  // the wave's four pieces of the next slab (two row pairs of each image); beyond the split's last row -- and beyond its
  // last slab, where the loads only keep the counts uniform -- they read the zero page
  auto issue_half = [&](int stage, int j) {
    unsigned char* st = smem + stage * TN_STAGE + wave * 2048;
    {
      const bool okm = rrow[j] < nrows;
      const bool okb = okm && (T <= 0 || (unsigned)(tpos[j] + shift) < (unsigned)T);
      const uint64_t sa = okm ? pa[j] : zero64, sb = okb ? pb[j] : zero64;
      glds16(reinterpret_cast<const void*>(sa), st + j * 1024);
      glds16(reinterpret_cast<const void*>(sb), st + TN_IMG + j * 1024);
      pa[j] += (uint64_t)step_a; pb[j] += (uint64_t)step_b; rrow[j] += TN_BK;
      if (T > 0) { tpos[j] += tstep; if (tpos[j] >= T) tpos[j] -= T; }
    }
  };
  auto issue = [&](int, int stage) { issue_half(stage, 0); issue_half(stage, 1); };
  issue(0, 0); issue(1, 1); issue(2, 2);

  // ---- fragment reads (transposing): lane -> k-row 8*(g>>1) + qq (+4 for the second half), bytes of the 32-column
  // block blk of an image row: chunk 4*(blk ^ qq) + 2*(g&1) + (pp>>1), + 8*(pp&1).  The reads are issued in assembly:
  // the compiler would hold an LDS read back until every direct-to-LDS load in flight (the next three slabs) has landed.
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const uint32_t frow_off = (uint32_t)(uintptr_t)smem + (uint32_t)((8 * (g >> 1) + qq) * 512 + 8 * (pp & 1));
  uint32_t fa[2], fb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = frow_off + (uint32_t)((4 * ((wa + 4 * i) ^ qq) + 2 * (g & 1) + (pp >> 1)) << 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = frow_off + (uint32_t)((4 * ((4 * wn + j) ^ qq) + 2 * (g & 1) + (pp >> 1)) << 4) + TN_IMG;

  f32x16 acc[2][4], bacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    bacc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j][e] = 0.f;
  }
  const bool want_db = J.db != nullptr && tb == 0;             // workgroup-uniform; wave (wa, wn) sums A block wa + 4 * wn
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  auto mma = [&](const TnFrags& f) {
    bf16x8 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(f.a[i][0], f.a[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nblk) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(f.b[j][0], f.b[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (i < nablk) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i][j], 0, 0, 0);
      }
    if (want_db && wn < nablk) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wn ? a[1] : a[0], ones, bacc, 0, 0, 0);
  };

  // Software pipeline over the slabs, one barrier per slab: a wave multiplies slab q from fragments it fetched during
  // the previous interval, then fetches slab q+1's, issues its share of slab q+4 and waits for its pieces of slab q+2.
  // The LDS reads are never waited for right after they are issued (a burst that is drained at once runs the LDS at
  // half its rate with two waves per SIMD, and that wait used to sit in front of every slab's MFMAs).
  //   landing: every wave waits for ITS pieces of slab q+2 before the barrier that ends interval q; slab q+2 is first
  //   read in interval q+1.  reuse: slab q+4 goes to the stage slab q-1 occupied (five stages, all of the LDS), whose
  //   fragments every wave had in registers, and multiplied, before the barrier that starts interval q.
  // Measured with cycle stamps (-DMM16_STAMPS, scripts/tn_stamps.py), per wave and slab: 24 fragment reads 240-390
  // cycles of issue, 4 direct-to-LDS loads 410-740 (the CU's vector-memory path takes a 1-KiB piece every ~20 cycles and
  // all eight waves queue for it), 16 MFMAs 680: a wave's own instruction stream is ~1700 cycles per slab against 896
  // cycles of matrix-pipe time per SIMD, so running the two waves of a SIMD out of phase (tried three ways: one slab
  // apart, one phase apart with two barriers per slab, opposite order of the two parts) changes nothing, and the fetch and
  // the refill issued in two halves, one behind each half of a slab's MFMAs, is 4-6 % slower.
  if (!(p.dbg & 4)) issue(3, 3);
  wait_vm<2 * 4>();                        // this wave's pieces of slabs 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  TnFrags f0, f1;
  auto fetch = [&](int stage) {
    const uint32_t so = (uint32_t)(stage * TN_STAGE);
    uint32_t aa[2], ab[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) aa[i] = fa[i] + so;
#pragma unroll
    for (int j = 0; j < 4; ++j) ab[j] = fb[j] + so;
    tn_fetch<0>(aa, ab, f0);
    tn_fetch<16 * 512>(aa, ab, f1);
  };
  // (a wave of an edge tile with no block pair to multiply and no column sums to form reads no fragments: it refills and meets the barriers)
  const bool busy = nablk > 0 && (nblk > 0 || (want_db && wn < nablk));
  if (busy) fetch(0);
  int sq = 0;                              // stage of slab q
#ifdef MM16_STAMPS
  unsigned long long ts_[12], sum_[12] = {0};
#define TN_LAND0(f) tn_landed<0>(f)        /* scalar-memory stamps share the counter of the LDS reads: no counted waits */
#else
#define TN_LAND0(f) tn_landed<12>(f)
#endif
  for (int q = 0; q < nslab; ++q) {
    const int s1 = sq + 1 == TN_ST ? 0 : sq + 1;             // stage of slab q+1
    const int s4 = sq == 0 ? TN_ST - 1 : sq - 1;             // stage of slab q+4 = the one slab q-1 occupied
    TSTAMP(0);
    TN_LAND0(f0);
    TSTAMP(1);
    if (!(p.dbg & 2)) mma(f0);
    TSTAMP(2);
    tn_landed<0>(f1);
    TSTAMP(3);
    if (!(p.dbg & 2)) mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(4);
    if (busy) fetch(s1);
    TSTAMP(5);
    if (!(p.dbg & 4)) issue(q + 4, s4);
    TSTAMP(6);
    wait_vm<2 * 4>();                      // this wave's pieces of slab q+2 have landed (those of q+3, q+4 may be in flight)
    TSTAMP(7);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MM16_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // [0] wait f0, [1] mma f0 issue, [2] wait f1, [3] mma f1 issue, [4] fetch issue, [5] refill issue, [6] wait vm
    for (int i = 0; i < 7; ++i) sum_[i] += ts_[i + 1] - ts_[i];
#endif
    sq = s1;
  }
#ifdef MM16_STAMPS
  if (lane == 0 && (wave == 0 || wave == 4)) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_tn_stamps[(wave ? 16 : 0) + i], sum_[i]);
    atomicAdd(&g_tn_stamps[(wave ? 16 : 0) + 15], (unsigned long long)nslab);
  }
#endif
  tn_landed<0>(f0);                        // (fragments of the slab beyond the last: fetched, never multiplied)
  tn_landed<0>(f1);

  // ---- accumulate the tile: lanes run along a row of C (two 128-byte segments per instruction)
  if ((p.dbg & 1) && acc[0][0][0] != 123.456f) return;
  const int lr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int na = TN_TA * ta + 32 * (wa + 4 * i) + (e & 3) + 8 * (e >> 2) + 4 * h;      // row inside the job's A range
      if (na >= J.na_out) continue;
      const int dst = tn_map_row(J, na);
      if (dst < 0) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = b_c0 + 128 * wn + 32 * j + lr;
        if (j < nblk && nb < c_end) atomicAdd(J.C + (int64_t)dst * J.ldc + nb, acc[i][j][e]);
      }
      if (want_db && i == wn && lr == 0) {
        atomicAdd(J.db + dst, bacc[e]);
        if (J.db2) atomicAdd(J.db2 + dst, bacc[e]);
      }
    }
  wait_vm<0>();
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" int tad_mm16_nt(int64_t M, int64_t N, int64_t K, const void* A16, int64_t lda, const void* W16, int64_t ldw,
                           const float* bias, const void* residual, int res_dtype, int64_t ld_res, const void* relu_of,
                           int relu_dtype, int64_t ld_relu, void* C, int c_dtype, int64_t ldc, void* Z16, int64_t ldz, int act,
                           tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "tad_mm16_nt: bad shape M=%ld N=%ld K=%ld lda=%ld ldw=%ld ldc=%ld",
            (long)M, (long)N, (long)K, (long)lda, (long)ldw, (long)ldc);
  if (M == 0) return TAD_OK;                   // an empty batch has no operands to check
  TAD_CHECK(A16 && W16 && C, "tad_mm16_nt: null pointer");
  TAD_CHECK(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && al16(A16) && al16(W16), "tad_mm16_nt: A16 / W16 need 16-byte aligned rows (K, lda, ldw multiples of 8)");
  TAD_CHECK(K <= 1024, "tad_mm16_nt: K = %ld exceeds 1024 (the rows of A a wave keeps in registers)", (long)K);
  TAD_CHECK(N % 8 == 0 && N <= 8192, "tad_mm16_nt: N = %ld must be a multiple of 8, at most 8192", (long)N);
  TAD_CHECK(c_dtype == TAD_F32 || c_dtype == TAD_BF16, "tad_mm16_nt: bad c_dtype %d", c_dtype);
  // every [M,N] operand moves in 16-byte pieces: bf16 ones need 8-element, fp32 ones 4-element aligned rows
  auto mn_ok = [&](const void* q, int64_t ld, int dt) { return al16(q) && ld >= N && ld % (dt == TAD_BF16 ? 8 : 4) == 0 && (dt == TAD_BF16 || dt == TAD_F32); };
  TAD_CHECK(mn_ok(C, ldc, c_dtype), "tad_mm16_nt: C needs a 16-byte aligned base and ldc %% 8 == 0 (bf16) / %% 4 == 0 (fp32)");
  TAD_CHECK(!Z16 || mn_ok(Z16, ldz, act == TAD_ACT_SILU_Z32 ? TAD_F32 : TAD_BF16), "tad_mm16_nt: bad Z16");
  TAD_CHECK(!residual || mn_ok(residual, ld_res, res_dtype), "tad_mm16_nt: bad residual");
  TAD_CHECK(!relu_of || mn_ok(relu_of, ld_relu, relu_dtype), "tad_mm16_nt: bad relu_of");
  TAD_CHECK(act == TAD_ACT_NONE || act == TAD_ACT_RELU || act == TAD_ACT_SILU || (act == TAD_ACT_DSILU && relu_of) || (act == TAD_ACT_SILU_Z32 && Z16),
            "tad_mm16_nt: bad act %d (TAD_ACT_DSILU needs relu_of, TAD_ACT_SILU_Z32 needs Z)", act);
  TAD_CHECK(lda < (1 << 22) && ldw < (1 << 22) && ldc < (1 << 22) && N * ldw < (1ll << 30), "tad_mm16_nt: leading dimension too large");
  // the [M,N] operands are addressed with 32-bit byte offsets
  TAD_CHECK(M * ldc * 4 < 0xfff00000ll && M * ld_res * 4 < 0xfff00000ll && M * ld_relu * 4 < 0xfff00000ll && M * ldz * 2 < 0xfff00000ll,
            "tad_mm16_nt: an [M,N] operand exceeds 4 GiB");
  TAD_CHECK(ldw >= cdiv64(K, NT_KD) * NT_KD, "tad_mm16_nt: W16 must be a padded image: ldw >= K rounded up to a multiple of %d (zeros beyond K), rows readable up to N rounded up to %d", NT_KD, NT_BN);
  if (M == 0) return TAD_OK;
  NtParams p{};
  p.A = reinterpret_cast<const bf16_t*>(A16); p.lda = lda;
  p.W = reinterpret_cast<const bf16_t*>(W16); p.ldw = ldw;
  p.C = C; p.ldc = ldc; p.c16 = c_dtype == TAD_BF16;
  p.Z = Z16; p.ldz = ldz; p.z32 = act == TAD_ACT_SILU_Z32;
  p.bias = bias;
  p.res = residual; p.ld_res = ld_res; p.res16 = res_dtype == TAD_BF16;
  p.relu = relu_of; p.ld_relu = ld_relu; p.relu16 = relu_dtype == TAD_BF16;
  p.M = M; p.N = N; p.K = K; p.act = act;
  p.nks_img = (int)cdiv64(K, NT_KD);
  const int ks = (int)cdiv64(K, 16);
  hipStream_t s = (hipStream_t)stream;
  if (ks <= 8) return launch_nt<8>(p, s);
  if (ks <= 13) return launch_nt<13>(p, s);
  if (ks <= 16) return launch_nt<16>(p, s);
  if (ks <= 25) return launch_nt<25>(p, s);
  if (ks <= 28) return launch_nt<28>(p, s);
  if (ks <= 36) return launch_nt<36>(p, s);
  if (ks <= 50) return launch_nt<50>(p, s);
  return launch_nt<64>(p, s);
}

extern "C" int tad_mm16_tn(int64_t M, int64_t T, const void* A16, int64_t lda, int njobs, const tad_mm16_tn_job_t* jobs, tad_stream_t stream) {
  TAD_CHECK(M >= 0 && M < (1ll << 31) && T >= 0 && T < (1 << 30) && (T == 0 || M % T == 0), "tad_mm16_tn: bad M=%ld T=%ld", (long)M, (long)T);
  TAD_CHECK(jobs && njobs >= 1 && njobs <= TN_MAXJOBS, "tad_mm16_tn: 1..%d jobs", TN_MAXJOBS);
  if (M == 0) return TAD_OK;                   // an empty batch adds nothing
  TAD_CHECK(!A16 || (al16(A16) && lda % 8 == 0 && lda < (1 << 22)), "tad_mm16_tn: A16 needs 16-byte aligned rows (lda %% 8 == 0)");
  TnParams p{};
  p.M = M; p.T = (int)T; p.njobs = njobs;
  int tiles = 0;
  for (int k = 0; k < njobs; ++k) {
    const tad_mm16_tn_job_t& u = jobs[k];
    TAD_CHECK(u.B16 && u.C && al16(u.B16) && u.ldb % 8 == 0 && u.ldb < (1 << 22) && u.nb >= 8 && u.nb % 8 == 0 && u.ldb >= u.nb,
              "tad_mm16_tn: job %d: B16 needs 16-byte aligned rows, nb a multiple of 8", k);
    const void* ja = u.A16 ? u.A16 : A16;
    const int64_t jlda = u.A16 ? u.lda : lda;
    TAD_CHECK(ja && al16(ja) && jlda % 8 == 0 && jlda < (1 << 22), "tad_mm16_tn: job %d: no A operand, or rows not 16-byte aligned", k);
    const int64_t na8 = (u.na + 7) / 8 * 8;               // whole 16-byte pieces are read; columns beyond na produce nothing
    TAD_CHECK(u.a0 >= 0 && u.na >= 1 && u.a0 % 8 == 0 && u.a0 + na8 <= jlda, "tad_mm16_tn: job %d: bad A column range", k);
    TAD_CHECK(u.ldc > u.nb - 8 && u.ldc >= 1 && (T > 0 || u.shift == 0) && u.shift > -(T > 0 ? T : 1) && u.shift < (T > 0 ? T : 1), "tad_mm16_tn: job %d: bad ldc / shift", k);
    TAD_CHECK(u.db || !u.db2, "tad_mm16_tn: job %d: db2 without db", k);
    TnJob& j = p.job[k];
    j.A = reinterpret_cast<const bf16_t*>(ja); j.lda = jlda;
    j.B = reinterpret_cast<const bf16_t*>(u.B16); j.ldb = u.ldb; j.nb = (int)u.nb; j.shift = (int)u.shift;
    j.C = u.C; j.ldc = u.ldc; j.db = u.db; j.db2 = u.db2; j.a0 = (int)u.a0; j.na = (int)na8; j.na_out = (int)u.na;
    j.mapH = 0;
    if (u.gate_map[0] > 0) {
      const int H = u.gate_map[0], G = u.gate_map[1];
      TAD_CHECK((G == 3 || G == 4) && u.na % (4 * (int64_t)H) == 0, "tad_mm16_tn: job %d: bad gate map (na must be a multiple of 4*H)", k);
      for (int sidx = 0; sidx < 4; ++sidx) { TAD_CHECK(u.gate_map[2 + sidx] < G, "tad_mm16_tn: job %d: gate slot beyond G", k); j.slot[sidx] = u.gate_map[2 + sidx]; }
      j.mapH = H; j.mapG = G;
    }
    j.tiles_a = (int)cdiv64(na8, TN_TA); j.tiles_b = (int)cdiv64(u.nb, TN_TB); j.tile_base = tiles;
    tiles += j.tiles_a * j.tiles_b;
  }
  p.tiles_total = tiles;
  // reduction splits: one workgroup per CU in flight, two rounds of them (the atomics of the first round's tiles
  // run under the second round's products); every split adds its tiles with atomics
  static const char* wgs_env = getenv("TAD_MM16_TN_WGS");
  const int64_t target = wgs_env ? atoi(wgs_env) : 256;
  const int64_t slabs = cdiv64(M, TN_BK);
  int64_t splits = target / tiles;
  // a split shorter than a dozen slabs is all pipeline fill and atomics: at 2400 rows (the reference's batch of 100
  // windows) the five launches of a step average 48.7 / 41.0 / 37.4 / 40.0 us with at least 4 / 8 / 12 / 16 slabs per
  // split (6144 rows: 59 / 59 / 59 / 57); from 256 * 12 slabs per tile column on (the bench) the limit does not bind
  static const char* ms_env = getenv("TAD_MM16_TN_MINSLABS");
  const int64_t min_slabs = ms_env && atoi(ms_env) > 0 ? atoi(ms_env) : 12;
  if (splits > slabs / min_slabs) splits = slabs / min_slabs;
  if (splits < 1) splits = 1;
  p.rows_per_split = cdiv64(slabs, splits) * TN_BK;
  p.nsplits = (int)cdiv64(M, p.rows_per_split);
  p.nitems = (int64_t)p.nsplits * tiles;
  // Splits sized tile by tile.  Time of a tile per slab of rows ~ alpha (the refill of the slab and the barrier: the same
  // for every tile) + (1 - alpha) x its share of a full tile's matrix-pipe time: the busiest SIMD multiplies
  // ceil(A blocks / 4) x (B blocks) of the full tile's 2 x 8 block pairs.
  static const char* bal_env = getenv("TAD_MM16_TN_BALANCE");      // experiments: 0 = the same number of splits for every tile
  static const char* alpha_env = getenv("TAD_MM16_TN_ALPHA");
  const double alpha = alpha_env ? atof(alpha_env) : 0.5;      // (0.45-0.5 measured best at the bench shapes: 248-253 -> 235-240 us per launch)
  p.balanced = 0;
  if (!(bal_env && atoi(bal_env) == 0) && tiles <= 1023 && tiles <= target && splits > 1) {
    std::vector<double> cost((size_t)tiles);
    double total = 0.0;
    for (int k = 0; k < njobs; ++k) {
      const TnJob& j = p.job[k];
      for (int ta = 0; ta < j.tiles_a; ++ta)
        for (int tb = 0; tb < j.tiles_b; ++tb) {
          int nA = (j.na - TN_TA * ta + 31) / 32, nB = (j.nb - TN_TB * tb + 31) / 32;
          nA = nA > 8 ? 8 : nA; nB = nB > 8 ? 8 : nB;
          const double c = alpha + (1.0 - alpha) * ((nA + 3) / 4) * nB / 16.0;
          cost[(size_t)(j.tile_base + ta * j.tiles_b + tb)] = c;
          total += c;
        }
    }
    const int64_t max_n = std::max<int64_t>(1, slabs / min_slabs);
    std::vector<int64_t> n((size_t)tiles);
    int64_t used = 0;
    for (int t = 0; t < tiles; ++t) {
      int64_t v = (int64_t)(target * cost[(size_t)t] / total);
      v = v < 1 ? 1 : v > max_n ? max_n : v;
      n[(size_t)t] = v; used += v;
    }
    // what rounding left over goes, one split at a time, to the tile whose items are the longest
    while (used < target) {
      int best = -1; double worst = 0.0;
      for (int t = 0; t < tiles; ++t) {
        const double len = cost[(size_t)t] / (double)n[(size_t)t];
        if (n[(size_t)t] < max_n && len > worst) { worst = len; best = t; }
      }
      if (best < 0) break;
      ++n[(size_t)best]; ++used;
    }
    // no tile keeps more splits than it needs to stay under the longest item (splits cost atomics); tiles that all cost
    // the same end up with the same number, and the launch keeps the split-major order of the plain schedule
    double longest = 0.0;
    for (int t = 0; t < tiles; ++t) longest = std::max(longest, cost[(size_t)t] / (double)n[(size_t)t]);
    bool uniform = true;
    for (int t = 0; t < tiles; ++t) {
      int64_t& v = n[(size_t)t];
      while (v > 1 && cost[(size_t)t] / (double)(v - 1) <= longest * (1.0 + 1e-9)) --v;
      uniform = uniform && v == n[0];
    }
    struct It { int64_t m; uint32_t e; };
    std::vector<It> items;
    bool fits = true;
    for (int t = 0; t < tiles && fits; ++t) {
      const int64_t nt = n[(size_t)t], rps = cdiv64(slabs, nt) * TN_BK;
      fits = nt <= 2047;
      for (int64_t sidx = 0; sidx * rps < M && fits; ++sidx) items.push_back(It{sidx * rps, (uint32_t)t | (uint32_t)sidx << 10 | (uint32_t)nt << 21});
    }
    if (fits && !uniform && items.size() <= (size_t)TN_MAXITEMS) {
      // rows first: the workgroups of an XCD (contiguous runs of items) then share the rows of A and B they fetch
      // (tile-major order fetched 37 % more; with every tile split alike the launch fetched 11 % less and took 7 % longer)
      std::stable_sort(items.begin(), items.end(), [](const It& x, const It& y) { return x.m < y.m; });
      for (size_t i = 0; i < items.size(); ++i) p.item[i] = items[i].e;
      p.nitems = (int64_t)items.size();
      p.balanced = 1;
    }
  }
  { static const char* dbg_env = getenv("TAD_MM16_DBG"); p.dbg = dbg_env ? atoi(dbg_env) : 0; }
  const size_t lds = TN_ST * TN_STAGE;
  static size_t configured[64] = {};
  if (int rc = mm16_set_lds(reinterpret_cast<const void*>(mm16_tn_kernel), lds, configured)) return rc;
  hipLaunchKernelGGL(mm16_tn_kernel, dim3((unsigned)p.nitems), dim3(512), lds, (hipStream_t)stream, p);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("mm16_tn_kernel");
  return TAD_OK;
}
