// Cluster form of the forward LSTM / GRU sequence kernel for gfx950, H = 512, bf16 operands and storage.
//
// At H = 512 one direction's W_hh (4H x H bf16 = 2 MB) fits no CU, and a workgroup that owns all hidden units of its
// batch rows streams it from L2 every step: that stream (44 GB/s per CU), not the matrix pipe, bounded the step.
// Here EIGHT workgroups (a group, one workgroup per CU) share 128 batch rows of one direction and each owns 64 hidden
// units: its slice of W_hh (256 gate rows x 512, 256 KB) sits in the registers of its 8 waves for the whole launch
// (128 VGPRs per lane), as the A operand of v_mfma_f32_32x32x16_bf16; h_{t-1} of all 512 units (the B operand,
// transposed: column = batch row) sits in LDS.  Per step a workgroup computes the gates of its 64 units for the
// group's rows, applies the cell update (c, or the GRU's h, of its units stays in registers) and hands its 64 columns
// of h_t to the seven other workgroups through a buffer in device memory.
//
// The 128 rows are walked as four quarters of 32; a phase = one step of one quarter, and the phases of a step follow
// each other quarter by quarter.  The hand-off of a phase has three phases of other quarters' work to complete in: the
// operands of phase p + 1 (h_{t-1} of its quarter from the exchange buffer, its input-side pre-activations) are loaded
// into registers at the top of phase p and written to a second LDS tile at its end, and the results of phase p leave at
// the top of phase p + 1, so that no wait of the pipeline is exposed as long as the workgroups of a group keep pace.
// Everything that crosses the memory path does so in whole lines: a lane of the product holds 16 gate values of one batch
// row (32 bytes of 64 different rows per store instruction), so the results are transposed through LDS staging tiles
// and leave 16 bytes per lane along rows (8 full lines per wave instruction); the pre-activations arrive the same way.
//
// Hand-off: the slice is stored write-through (sc1), 16 bytes per lane, every 128-byte line whole by one wave
// instruction; every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier follows, then one
// lane adds 1 to the quarter's arrival counter (agent scope).  A consumer polls that counter with sc1 loads from one lane
// until it reaches 8 x (publishes so far), passes a workgroup barrier, and reads the eight slices with sc1 loads
// (they bypass the CU's L1, which other CUs' stores never refresh).  Counters only grow within a launch and are
// zeroed by a small kernel in front of it (tad_zero_async).  Every wait is bounded: a wait that runs out sets the workspace's status word and
// the workgroup leaves, so a launch whose workgroups cannot all be resident ends instead of hanging.
#include "common.h"
#include "rnn_layout.h"
#include <stdlib.h>
#include <algorithm>

// Diagnostic build only (-DCL_STAMPS): cycle counts of the parts of a phase, summed over wave 0 of every workgroup.
#ifdef CL_STAMPS
__device__ unsigned long long g_cl_stamps[8];
#define CSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                       __builtin_amdgcn_sched_barrier(0); acc_st[k] += now_ - last_st; last_st = now_; } while (0)
extern "C" int tad_dbg_cl_stamps(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cl_stamps), 64);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_cl_stamps), z, 64); }
  return 0;
}
#else
#define CSTAMP(k)
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

constexpr int CL_H = 512, CL_NC = 8, CL_KS = CL_H / 16;
constexpr int CL_QR = 32, CL_NQ = 4, CL_ROWS = CL_QR * CL_NQ;     // rows of a quarter, quarters, rows of a group
constexpr int CL_LDH = CL_H + 8;       // row strides in LDS, padded so that the 16-byte accesses of 16 rows meet no bank twice:
constexpr int CL_LDX = 256 + 8;        //   h tile (bf16); pre-activations / gate staging (bf16, 256 gate columns);
constexpr int CL_LDC = 64 + 4;         //   cell state / h staging (fp32, 64 units);
constexpr int CL_LDS16 = 64 + 8;       //   h_t staging (bf16, 64 units)
constexpr size_t CL_T_BYTES = (size_t)CL_QR * CL_LDH * 2, CL_X_BYTES = (size_t)CL_QR * CL_LDX * 2, CL_C_BYTES = (size_t)CL_QR * CL_LDC * 4;
constexpr size_t CL_T_OFF = 0, CL_X_OFF = 2 * CL_T_BYTES, CL_G_OFF = CL_X_OFF + 2 * CL_X_BYTES, CL_C_OFF = CL_G_OFF + CL_X_BYTES,
                 CL_O_OFF = CL_C_OFF + CL_C_BYTES, CL_S_OFF = CL_O_OFF + CL_C_BYTES, CL_BH_OFF = CL_S_OFF + (size_t)CL_QR * CL_LDS16 * 2;
constexpr size_t CL_LDS_BYTES = CL_BH_OFF + 3 * 64 * 4;           // (GRU: b_hh of the workgroup's units, [3][64] fp32)
constexpr size_t CL_XQ_BYTES = (size_t)CL_QR * CL_H * 2;          // h_t of one quarter: 32 KB
constexpr size_t CL_SLOT_BYTES = CL_NQ * 2 * CL_XQ_BYTES;         // [quarter][step parity]
constexpr int CL_CTR_STRIDE = 32;                                 // arrival counters sit 128 bytes apart
constexpr int CL_MAX_SLOTS = 32;
constexpr size_t CL_CTR_BYTES = (size_t)CL_MAX_SLOTS * CL_NQ * CL_CTR_STRIDE * 4;     // 16 KB, zeroed before every launch
constexpr size_t CL_STATUS_OFF = CL_CTR_BYTES, CL_XBUF_OFF = 2 * CL_CTR_BYTES;
constexpr size_t CL_WS_BYTES = CL_XBUF_OFF + (size_t)CL_MAX_SLOTS * CL_SLOT_BYTES;
constexpr long long CL_SPIN_TICKS = 50000000;     // 0.5 s of the 100 MHz wall clock
static_assert(CL_LDS_BYTES <= 160 * 1024, "LDS");

struct ClArgs {
  tad_rnn_dir_t d[2];          // w_packed: the cluster-form image
  tad_rnn_xin_t xi[2];         // XIN launches: the layer's input and input-side weights instead of d[].xproj
  int B, T, ntiles, ndir, nslots, nitems;
  unsigned char* xbuf;
  unsigned* ctr;
  unsigned* status;
};

__device__ __forceinline__ uint32_t cl_pk2(float a, float b) {
  bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
__device__ __forceinline__ float cl_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float cl_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

__device__ __forceinline__ int cl_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// one lane: until *ctr >= target, or the time limit (then the status word is set)
__device__ __forceinline__ bool cl_wait(unsigned* ctr, unsigned target, unsigned* status) {
  if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
  const long long t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if (wall_clock64() - t0 > CL_SPIN_TICKS) {
      __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

// XIN: the layer's input has at most 32 columns (the first layer of the VAE's encoders: the table's F features).  Its
// input-side pre-activations are then not read from memory but formed here: x_t W_ih^T is one or two more MFMA steps on
// the accumulators that W_hh h_{t-1} lands in, seeded with the bias (W_ih's slice: 8 registers).  The product that wrote
// them ([B*T, 4H] bf16 per direction), its launch and a phase's 16 KB of loads through LDS are gone; the pre-activations
// are no longer rounded to bf16 on the way.  GRU: the candidate gate's input side goes to the slot of the accumulator
// that the recurrent image leaves empty (slot 3), its recurrent side stays in slot 2.
constexpr int CL_LDXI = 40;            // row stride of the input tile in LDS (bf16; 32 columns + 8)
constexpr size_t CL_XI_BIAS_OFF = 4096;                    // inside the first buffer of the pre-activation tiles (unused by XIN)

template <int CELL, bool STASH, bool XIN>
__global__ __launch_bounds__(512) void rnn_cluster_fwd_kernel(ClArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hb = lane >> 5, n = lane & 31;
  // (The eight workgroups of a group are consecutive: under the round-robin dispatch that is observed they sit on eight
  // XCDs.  Taking them from one XCD instead, so that its L2 serves seven of the eight reads of a published slice, measured
  // the same to 6 % slower.)
  const int slot = blockIdx.x >> 3, member = blockIdx.x & 7;
  const int mw = member * 8 + wave;                  // the wave's 8 units start at 8 mw
  unsigned* const ctr = a.ctr + (size_t)slot * CL_NQ * CL_CTR_STRIDE;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(a.xbuf + (size_t)slot * CL_SLOT_BYTES, 0, (int)CL_SLOT_BYTES, 0x00020000);
  const int T = a.T;
  if (tid == 0) dead = 0;
  // the lane's places in the LDS tiles.  Product side: batch row n of the quarter, accumulator register 4 j + s = slot s
  // of unit 8 wave + 4 hb + j of the workgroup's 64.  Copy side: 16-byte pieces along rows.
  unsigned char* const xg_lane = smem + ((size_t)n * CL_LDX + 32 * wave + 16 * hb) * 2;           // + CL_X_OFF (+ buffer) or CL_G_OFF
  unsigned char* const co_lane = smem + ((size_t)n * CL_LDC + 8 * wave + 4 * hb) * 4;             // + CL_C_OFF or CL_O_OFF
  unsigned char* const s_lane = smem + CL_S_OFF + ((size_t)n * CL_LDS16 + 8 * wave + 4 * hb) * 2;
  const unsigned char* const h_lane = smem + CL_T_OFF + ((size_t)n * CL_LDH + 8 * hb) * 2;
  // Copy side (16-byte pieces along rows): gate columns, rows gr and gr + 16, piece gc of 32; fp32 columns, row cr, piece cc of
  // 16; bf16 h columns (threads 0..255), row sr, piece sc of 8.  These and the byte offsets made of them are recomputed from
  // the thread index where they are used (a handful of integer operations) instead of living in registers through the
  // matrix product, where 128 registers of W_hh leave no room for them: t2_ is the thread index behind cl_opaque, which keeps
  // the compiler from hoisting them.  (An asm statement makes the compiler wait for memory accesses in flight: cl_opaque
  // is called only where none are, or where the wait is due anyway.)
#define CL_COPY_IDX const int gr = t2_ >> 5, gc = t2_ & 31, cr = t2_ >> 4, cc = t2_ & 15, sr = (t2_ >> 3) & 31, sc = t2_ & 7; \
                    (void)gr; (void)gc; (void)cr; (void)cc; (void)sr; (void)sc

#ifdef CL_STAMPS
  unsigned long long acc_st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_st = __builtin_amdgcn_s_memtime();
#endif
  uint4 wreg[CL_KS];
  uint4 wih[XIN ? 2 : 1];
  int cur_dir = -1;
  unsigned gbase = 0;                                // steps every quarter of this group has published before the current item

  for (int item = slot; item < a.nitems; item += a.nslots) {
    const int dir = item / a.ntiles, tile = item - dir * a.ntiles;
    const tad_rnn_dir_t d = a.d[dir];
    if (dir != cur_dir) {
      const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed) + (size_t)mw * CL_KS * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < CL_KS; ++kk) wreg[kk] = wp[kk * 64];
      if constexpr (XIN) {
        // A operand rows of the wave in the gate-interleaved image of W_ih (row 4 unit + slot; zeros beyond the input width)
        const int m = lane & 31, slot = m & 3, unit = 8 * mw + 4 * ((m >> 2) & 1) + (m >> 3);
        const int srow = (CELL == TAD_CELL_GRU && slot == 3) ? 2 : slot;
        const bf16_t* wrow = reinterpret_cast<const bf16_t*>(a.xi[dir].w_ih16) + (int64_t)(4 * unit + srow) * a.xi[dir].ldw + 8 * hb;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          wih[kk] = *reinterpret_cast<const uint4*>(wrow + 16 * kk);
          if (CELL == TAD_CELL_GRU && slot == 2) wih[kk] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      cur_dir = dir;
    }
    const int64_t row0 = (int64_t)tile * CL_ROWS;
    // Global addresses = tensor base (scalar registers) + 32-bit byte offset = (one that is the same for the whole
    // workgroup) + (the thread's own inside the group's 128 rows: one register per tensor for the whole item).
    const int nrows = a.B - (int)row0;                // rows of the tile that exist (the others are neither loaded nor stored)
#define CL_OOB 0x80000000u      /* beyond the 2 GiB the descriptors span: such a load returns zeros, such a store is dropped */
#define CL_VX (uint32_t)((gr * T * (int)d.ld_xproj + 256 * member + 8 * gc) * 2)
#define CL_VG (uint32_t)((gr * T * (int)d.ld_gates + 256 * member + 8 * gc) * 2)
#define CL_VC (uint32_t)((cr * T * (int)d.ld_c + 64 * member + 4 * cc) * 4)
#define CL_VO (uint32_t)((cr * T * (int)d.ld_out + 64 * member + 4 * cc) * 4)
#define CL_VS (uint32_t)((sr * T * (int)d.ld_out16 + 64 * member + 8 * sc) * 2)
    // (the host admits only tensors of less than 2 GiB to this form: buffer addressing, 32-bit arithmetic throughout)
    auto rsrc_of = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); };
    const __amdgpu_buffer_rsrc_t x_rs = rsrc_of(XIN ? a.xi[dir].x16 : d.xproj), g_rs = rsrc_of(d.gates), c_rs = rsrc_of(d.cstate), o_rs = rsrc_of(d.out), s_rs = rsrc_of(d.out16);
    auto row_off = [&](int q, int tt, int64_t ld, int esz, int extra_rows) -> uint32_t {     // byte offset of (row0 + 32 q + extra_rows, tt)
      return (uint32_t)((((int)row0 + CL_QR * q + extra_rows) * T + tt) * (int)ld * esz);
    };
    // pre-activations of (quarter q, time tt): two 16-byte pieces per thread (XIN: the input rows, one piece for the first 128 threads)
    const int xi_ld = XIN ? (int)a.xi[dir].ldx : 0, xi_nch = XIN ? a.xi[dir].kin >> 3 : 0;
    auto load_x = [&](int t2_, int q, int tt, u32x4_t* xv) {
      CL_COPY_IDX;
      if constexpr (XIN) {
        const int xr = t2_ >> 2, xc = t2_ & 3;           // row of the quarter, 16-byte piece of the row
        const bool ok = t2_ < 128 && xc < xi_nch && CL_QR * q + xr < nrows;
        xv[0] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? (uint32_t)((xr * T * xi_ld + 8 * xc) * 2) : CL_OOB, row_off(q, tt, xi_ld, 2, 0), 0);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)      // (no branch around a load: the wait for its result would drain every access in flight)
          xv[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, CL_QR * q + gr + 16 * i < nrows ? CL_VX : CL_OOB, row_off(q, tt, d.ld_xproj, 2, 16 * i), 0);
      }
    };
    auto put_x = [&](int t2_, int buf, const u32x4_t* xv) {
      CL_COPY_IDX;
      if constexpr (XIN) {
        if (t2_ < 128) *reinterpret_cast<u32x4_t*>(smem + CL_X_OFF + (size_t)buf * CL_X_BYTES + ((size_t)(t2_ >> 2) * CL_LDXI + 8 * (t2_ & 3)) * 2) = xv[0];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          *reinterpret_cast<u32x4_t*>(smem + CL_X_OFF + (size_t)buf * CL_X_BYTES + ((size_t)(gr + 16 * i) * CL_LDX + 8 * gc) * 2) = xv[i];
      }
    };

    for (int i = tid; i < (int)(2 * CL_T_BYTES / 16); i += 512) reinterpret_cast<uint4*>(smem + CL_T_OFF)[i] = make_uint4(0u, 0u, 0u, 0u);
    float st[CL_NQ][4];                              // LSTM: c; GRU: h (fp32) of (quarter, unit)
#pragma unroll
    for (int q = 0; q < CL_NQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) st[q][j] = 0.f;
    float* const bhl = reinterpret_cast<float*>(smem + CL_BH_OFF);
    if constexpr (CELL == TAD_CELL_GRU) {
      if (tid < 192) bhl[tid] = d.b_hh ? d.b_hh[(tid >> 6) * CL_H + 64 * member + (tid & 63)] : 0.f;
    }
    if constexpr (XIN) {
      // the bias of the workgroup's 256 gate rows in accumulator order (4 unit + slot), the GRU's candidate gate moved to slot 3.
      // (Lives in the unused part of the first tile buffer: a tile is 32 x 40 bf16 here.)
      if (tid < 256) {
        const int slot = tid & 3, srow = (CELL == TAD_CELL_GRU && slot == 3) ? 2 : slot;
        const float b = a.xi[dir].bias[4 * (64 * member + (tid >> 2)) + srow];
        reinterpret_cast<float*>(smem + CL_X_OFF + CL_XI_BIAS_OFF)[tid] = (CELL == TAD_CELL_GRU && slot == 2) ? 0.f : b;
      }
    }
    {
      u32x4_t xv[2];
      load_x(tid, 0, d.reverse ? T - 1 : 0, xv);
      put_x(tid, 0, xv);
    }
    __syncthreads();

    // results of the phase before, from the staging tiles to registers (the tiles are free again behind the barrier
    // that follows) and from there to memory
    u32x4_t og[2], oc, oo, os;
    auto take_results = [&](int t2_) {
      CL_COPY_IDX;
      if constexpr (STASH) {
        og[0] = *reinterpret_cast<const u32x4_t*>(smem + CL_G_OFF + ((size_t)gr * CL_LDX + 8 * gc) * 2);
        og[1] = *reinterpret_cast<const u32x4_t*>(smem + CL_G_OFF + ((size_t)(gr + 16) * CL_LDX + 8 * gc) * 2);
        if constexpr (CELL == TAD_CELL_LSTM) oc = *reinterpret_cast<const u32x4_t*>(smem + CL_C_OFF + ((size_t)cr * CL_LDC + 4 * cc) * 4);
      }
      if (d.out) oo = *reinterpret_cast<const u32x4_t*>(smem + CL_O_OFF + ((size_t)cr * CL_LDC + 4 * cc) * 4);
      os = *reinterpret_cast<const u32x4_t*>(smem + CL_S_OFF + ((size_t)sr * CL_LDS16 + 8 * sc) * 2);
    };
    auto publish = [&](int t2_, int q, int step) {            // the workgroup's 64 columns of h_t of a quarter: waves 0..3, 8 rows each
      CL_COPY_IDX;
      if (tid < 256) {
        const uint32_t off = (uint32_t)((q * 2 + (step & 1)) * CL_XQ_BYTES) + (uint32_t)((sr * CL_H + 64 * member + 8 * sc) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(os, xrs, off, 0, 16);
      }
    };
    // (in four parts, issued between the quarters of the matrix product: the memory path takes a wave's 1-KB stores at its
    // own pace, and the products already issued keep the matrix pipe busy meanwhile)
    auto store_results = [&](int t2_, int q, int tt, int part) {
      CL_COPY_IDX;
      if constexpr (STASH) {
        if (part == 0) __builtin_amdgcn_raw_buffer_store_b128(og[0], g_rs, CL_QR * q + gr < nrows ? CL_VG : CL_OOB, row_off(q, tt, d.ld_gates, 2, 0), 0);
        if (part == 1) __builtin_amdgcn_raw_buffer_store_b128(og[1], g_rs, CL_QR * q + gr + 16 < nrows ? CL_VG : CL_OOB, row_off(q, tt, d.ld_gates, 2, 16), 0);
        if constexpr (CELL == TAD_CELL_LSTM) {
          if (part == 2) __builtin_amdgcn_raw_buffer_store_b128(oc, c_rs, CL_QR * q + cr < nrows ? CL_VC : CL_OOB, row_off(q, tt, d.ld_c, 4, 0), 0);
        }
      }
      if (part == 3 && d.out) __builtin_amdgcn_raw_buffer_store_b128(oo, o_rs, CL_QR * q + cr < nrows ? CL_VO : CL_OOB, row_off(q, tt, d.ld_out, 4, 0), 0);
      if (part == 2 && d.out16) __builtin_amdgcn_raw_buffer_store_b128(os, s_rs, (tid < 256 && CL_QR * q + sr < nrows) ? CL_VS : CL_OOB, row_off(q, tt, d.ld_out16, 2, 0), 0);
    };

    unsigned seen = 0;                               // lane 0: an earlier reading of the counter the next poll asks about
    for (int step = 0; step < T; ++step) {
#pragma unroll
      for (int q = 0; q < CL_NQ; ++q) {
        // phase p = 4 step + q works on LDS buffers p & 1 = q & 1; the phase before is (pq, pstep), the next (nq, nstep)
        const int pq = (q + 3) & 3, pstep = q == 0 ? step - 1 : step;
        const int nq = (q + 1) & 3, nstep = q == 3 ? step + 1 : step;
        const bool has_prev = pstep >= 0, has_next = nstep < T;
        CSTAMP(0);
        const int ta = cl_opaque(tid);
        if (has_prev) take_results(ta);
        // h_{t-1} of the next phase's quarter: published three phases ago by every workgroup of the group.  (The poll comes
        // before this phase's stores: the wait for its load would wait for them too.)
        if (has_next && nstep > 0 && tid == 0 && seen < 8u * (gbase + (unsigned)nstep) &&
            !cl_wait(ctr + nq * CL_CTR_STRIDE, 8u * (gbase + (unsigned)nstep), a.status)) dead = 1;
        if (has_prev && pstep + 1 < T) publish(ta, pq, pstep);     // (nobody reads the last step's)
        CSTAMP(1);
        __syncthreads();
        if (dead) return;
        CSTAMP(2);
        u32x4_t fv[4], xv[2];
        if (has_next) {
          if (nstep > 0) {
            const uint32_t xoff = (uint32_t)((nq * 2 + ((nstep - 1) & 1)) * CL_XQ_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) fv[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + (uint32_t)(tid + 512 * i) * 16u, 0, 16);
          }
          load_x(ta, nq, d.reverse ? T - 1 - nstep : nstep, xv);
        }
        // the counter the next phase will ask about, read a phase ahead: its round trip is off that phase's critical path
        // (counters only grow: a value that was enough then is enough later)
        if (tid == 0) seen = __hip_atomic_load(ctr + ((q + 2) & 3) * CL_CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CSTAMP(3);

        // gates^T (32 gate rows of the wave x 32 batch rows) = x side + W_slice h_{t-1}^T
        f32x16 acc;
        float xn[4];
        if constexpr (XIN) {
          const f32x4* bl = reinterpret_cast<const f32x4*>(smem + CL_X_OFF + CL_XI_BIAS_OFF) + 8 * wave + 4 * hb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 b = bl[j];
            acc[4 * j] = b[0]; acc[4 * j + 1] = b[1]; acc[4 * j + 2] = b[2]; acc[4 * j + 3] = b[3];
            xn[j] = 0.f;
          }
          const unsigned char* xt = smem + CL_X_OFF + (size_t)(q & 1) * CL_X_BYTES + ((size_t)n * CL_LDXI + 8 * hb) * 2;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wih[0]), *reinterpret_cast<const bf16x8*>(xt), acc, 0, 0, 0);
          if (xi_nch > 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wih[1]), *reinterpret_cast<const bf16x8*>(xt + 32), acc, 0, 0, 0);
        } else
        {
          const unsigned char* xp = xg_lane + CL_X_OFF + (size_t)(q & 1) * CL_X_BYTES;
          const u32x4_t x0 = *reinterpret_cast<const u32x4_t*>(xp), x1 = *reinterpret_cast<const u32x4_t*>(xp + 16);
          const uint32_t w8[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float s0 = cl_lo(w8[2 * j]), s1 = cl_hi(w8[2 * j]), s2 = cl_lo(w8[2 * j + 1]), s3 = cl_hi(w8[2 * j + 1]);
            if constexpr (CELL == TAD_CELL_LSTM) {
              acc[4 * j] = s0; acc[4 * j + 1] = s1; acc[4 * j + 2] = s2; acc[4 * j + 3] = s3;
              xn[j] = 0.f;
            } else {
              acc[4 * j] = s0; acc[4 * j + 1] = s1; acc[4 * j + 2] = 0.f; acc[4 * j + 3] = 0.f;
              xn[j] = s2;
            }
          }
        }
        {
          const unsigned char* hp = h_lane + (size_t)(q & 1) * CL_T_BYTES;
#pragma unroll
          for (int part = 0; part < 4; ++part) {
#pragma unroll
            for (int kk = part * (CL_KS / 4); kk < (part + 1) * (CL_KS / 4); ++kk)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[kk]), *reinterpret_cast<const bf16x8*>(hp + kk * 32), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_prev) store_results(ta, pq, d.reverse ? T - 1 - pstep : pstep, part);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        CSTAMP(4);

        // gate nonlinearities and cell update of 4 units of one batch row per lane; results to the staging tiles
        {
          const bool ok = CL_QR * q + n < nrows;
          float hv[4], sv[16], cv[4];
          f32x4 bh[3] = {};
          if constexpr (CELL == TAD_CELL_GRU) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bh[g] = *reinterpret_cast<const f32x4*>(bhl + g * 64 + 8 * wave + 4 * hb);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float h;
            if constexpr (CELL == TAD_CELL_LSTM) {
              const float ig = fast_sigmoid(acc[4 * j]), fg = fast_sigmoid(acc[4 * j + 1]);
              const float gg = fast_tanh(acc[4 * j + 2]), og_ = fast_sigmoid(acc[4 * j + 3]);
              const float c = fg * st[q][j] + ig * gg;
              st[q][j] = c;
              h = og_ * fast_tanh(c);
              cv[j] = c;
              sv[4 * j] = ig; sv[4 * j + 1] = fg; sv[4 * j + 2] = gg; sv[4 * j + 3] = og_;
            } else {
              const float rg = fast_sigmoid(acc[4 * j] + bh[0][j]), zg = fast_sigmoid(acc[4 * j + 1] + bh[1][j]);
              const float hpn = acc[4 * j + 2] + bh[2][j];
              const float ng = fast_tanh((XIN ? acc[4 * j + 3] : xn[j]) + rg * hpn);
              h = (1.f - zg) * ng + zg * st[q][j];
              st[q][j] = h;
              cv[j] = 0.f;
              sv[4 * j] = rg; sv[4 * j + 1] = zg; sv[4 * j + 2] = ng; sv[4 * j + 3] = hpn;
            }
            if (!ok) h = 0.f;
            hv[j] = h;
          }
          *reinterpret_cast<uint2*>(s_lane) = make_uint2(cl_pk2(hv[0], hv[1]), cl_pk2(hv[2], hv[3]));
          if (d.out) *reinterpret_cast<float4*>(co_lane + CL_O_OFF) = make_float4(hv[0], hv[1], hv[2], hv[3]);
          if constexpr (STASH) {
            *reinterpret_cast<uint4*>(xg_lane + CL_G_OFF) = make_uint4(cl_pk2(sv[0], sv[1]), cl_pk2(sv[2], sv[3]), cl_pk2(sv[4], sv[5]), cl_pk2(sv[6], sv[7]));
            *reinterpret_cast<uint4*>(xg_lane + CL_G_OFF + 16) = make_uint4(cl_pk2(sv[8], sv[9]), cl_pk2(sv[10], sv[11]), cl_pk2(sv[12], sv[13]), cl_pk2(sv[14], sv[15]));
            if constexpr (CELL == TAD_CELL_LSTM) *reinterpret_cast<float4*>(co_lane + CL_C_OFF) = make_float4(cv[0], cv[1], cv[2], cv[3]);
          }
        }
        CSTAMP(5);
        // the next phase's operands into the other LDS buffers (last read one phase ago)
        if (has_next) {
          const int tb = cl_opaque(tid);
          if (nstep > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int idx = tb + 512 * i, r = idx >> 6, c = idx & 63;
              *reinterpret_cast<u32x4_t*>(smem + CL_T_OFF + (size_t)(nq & 1) * CL_T_BYTES + ((size_t)r * CL_LDH + 8 * c) * 2) = fv[i];
            }
          }
          put_x(tb, nq & 1, xv);
        }
        // (drains the hand-off stores issued at the top of this phase: they have had the whole phase)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CSTAMP(6);
        __syncthreads();
        CSTAMP(7);
        if (has_prev && tid == 0) __hip_atomic_fetch_add(ctr + pq * CL_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the last phase's results; its signal tells the group that this workgroup has taken its last read of the buffers
    take_results(tid);
    __syncthreads();
    for (int part = 0; part < 4; ++part) store_results(tid, CL_NQ - 1, d.reverse ? 0 : T - 1, part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr + (CL_NQ - 1) * CL_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gbase += (unsigned)T;
    // The next item's first publishes overwrite the exchange buffers: every workgroup of the group must be done with them.
    if (item + a.nslots < a.nitems) {
      if (tid == 0) {
        for (int q = 0; q < CL_NQ; ++q)
          if (!cl_wait(ctr + q * CL_CTR_STRIDE, 8u * gbase, a.status)) dead = 1;
      }
      __syncthreads();
      if (dead) return;
    }
  }
#ifdef CL_STAMPS
  if (tid == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_cl_stamps[k], acc_st[k]);
#endif
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cl_max_slots(const void* kernel) {
  // workgroups that can be resident at once, in groups of eight
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    int per_cu = 0;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, CL_LDS_BYTES) != hipSuccess) return 0;
    int s = prop.multiProcessorCount * per_cu / CL_NC;
    cached[dev] = s > CL_MAX_SLOTS ? CL_MAX_SLOTS : (s > 0 ? s : -1);
  }
  return cached[dev] > 0 ? cached[dev] : 0;
}

template <int CELL, bool STASH, bool XIN>
int launch_cluster(ClArgs a, hipStream_t s) {
  auto k = rnn_cluster_fwd_kernel<CELL, STASH, XIN>;
  static bool configured[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { tad_set_error("tad_rnn_fwd_ws: hipGetDevice failed"); return TAD_ERR_LAUNCH; }
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_LDS_BYTES);
    if (e != hipSuccess) { tad_set_error("tad_rnn_fwd_ws: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return TAD_ERR_LAUNCH; }
    configured[dev] = true;
  }
  const int max_slots = cl_max_slots(reinterpret_cast<const void*>(k));
  if (max_slots <= 0) { tad_set_error("tad_rnn_fwd_ws: no room for a group of %d workgroups on this device", CL_NC); return TAD_ERR_LAUNCH; }
  a.nslots = a.nitems < max_slots ? a.nitems : max_slots;
  hipError_t e = tad_zero_async(a.ctr, CL_CTR_BYTES, s);
  if (e != hipSuccess) { tad_set_error("tad_rnn_fwd_ws: zeroing the counters failed: %s", hipGetErrorString(e)); return TAD_ERR_LAUNCH; }
  hipLaunchKernelGGL(k, dim3((unsigned)(a.nslots * CL_NC)), dim3(512), CL_LDS_BYTES, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_cluster_fwd_kernel<%d, %s, %s>", CELL, STASH ? "true" : "false", XIN ? "true" : "false");
  return TAD_OK;
}

}  // namespace

extern "C" int64_t tad_rnn_workspace_bytes(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir) {
  (void)cell; (void)T; (void)ndir;
  static const char* off = getenv("TAD_RNN_CLUSTER");      // experiments: 0 = never the cluster form
  if (off && atoi(off) == 0) return 0;
  if (!rnn_cluster_shape(prec, H) || store != TAD_BF16 || B <= 0) return 0;
  return (int64_t)CL_WS_BYTES;
}

// xin == nullptr: the pre-activations come from dirs[].xproj; a call that the cluster form cannot take is tad_rnn_fwd.
// xin != nullptr: the form multiplies the layer's narrow input itself; a call that it cannot take is an error.
static int rnn_fwd_ws_any(const char* fn, int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                          const tad_rnn_xin_t* xin, void* workspace, int64_t workspace_bytes, tad_stream_t stream) {
  const int64_t need = tad_rnn_workspace_bytes(cell, prec, store, B, T, H, ndir);
  bool cluster = workspace && need > 0 && workspace_bytes >= need && (cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU) &&
                 (ndir == 1 || ndir == 2) && dirs && T > 0 && al16(workspace);
  bool stash = false, all_stash = true;
  if (cluster) {
    for (int i = 0; i < ndir; ++i) {
      const tad_rnn_dir_t& d = dirs[i];
      stash = stash || d.gates; all_stash = all_stash && d.gates;
      // every tensor moves in 16-byte pieces along its rows
      cluster = cluster && d.w_packed && (d.out || d.out16) &&
                (xin ? (xin[i].x16 && al16(xin[i].x16) && xin[i].ldx % 8 == 0 && xin[i].kin >= 8 && xin[i].kin <= 32 && xin[i].kin % 8 == 0 &&
                        xin[i].ldx >= xin[i].kin && xin[i].w_ih16 && al16(xin[i].w_ih16) && xin[i].ldw % 8 == 0 && xin[i].ldw >= 32 && xin[i].bias)
                     : (d.xproj && al16(d.xproj) && d.ld_xproj % 8 == 0 && d.ld_xproj >= 4 * H)) &&
                (!d.out || (al16(d.out) && d.ld_out % 4 == 0 && d.ld_out >= H)) &&
                (!d.out16 || (al16(d.out16) && d.ld_out16 % 8 == 0 && d.ld_out16 >= H)) &&
                (!d.gates || (al16(d.gates) && d.ld_gates % 8 == 0 && d.ld_gates >= 4 * H &&
                              (cell == TAD_CELL_GRU || (d.cstate && al16(d.cstate) && d.ld_c % 4 == 0 && d.ld_c >= H))));
    }
    // (the kernel's byte offsets are 32-bit)
    for (int i = 0; i < ndir && cluster; ++i) {
      const tad_rnn_dir_t& d = dirs[i];
      const int64_t ldmax = std::max(std::max(xin ? xin[i].ldx : d.ld_xproj, d.ld_gates), std::max(std::max(d.ld_out, d.ld_c) * 2, d.ld_out16));
      cluster = (B + CL_ROWS) * T * ldmax * 2 < ((int64_t)1 << 31);
    }
    cluster = cluster && stash == all_stash;
  }
  if (!cluster) {
    if (!xin) return tad_rnn_fwd(cell, prec, store, B, T, H, ndir, dirs, stream);
    tad_set_error("%s: the cluster form does not take this call (shape, workspace, alignment or leading dimensions)", fn);
    return TAD_ERR_INVALID;
  }

  ClArgs a{};
  const int G = cell == TAD_CELL_LSTM ? 4 : 3;
  for (int i = 0; i < ndir; ++i) {
    a.d[i] = dirs[i];
    a.d[i].w_packed = reinterpret_cast<const unsigned char*>(dirs[i].w_packed) + rnn_fwd_image_blocks(prec, H, G) * 1024;
    if (xin) a.xi[i] = xin[i];
  }
  a.B = (int)B; a.T = (int)T; a.ndir = ndir;
  a.ntiles = (int)cdiv64(B, CL_ROWS);
  a.nitems = a.ntiles * ndir;
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  a.ctr = reinterpret_cast<unsigned*>(ws);
  a.status = reinterpret_cast<unsigned*>(ws + CL_STATUS_OFF);
  a.xbuf = ws + CL_XBUF_OFF;
  hipStream_t s = (hipStream_t)stream;
  if (xin) {
    if (cell == TAD_CELL_LSTM) return stash ? launch_cluster<TAD_CELL_LSTM, true, true>(a, s) : launch_cluster<TAD_CELL_LSTM, false, true>(a, s);
    return stash ? launch_cluster<TAD_CELL_GRU, true, true>(a, s) : launch_cluster<TAD_CELL_GRU, false, true>(a, s);
  }
  if (cell == TAD_CELL_LSTM) return stash ? launch_cluster<TAD_CELL_LSTM, true, false>(a, s) : launch_cluster<TAD_CELL_LSTM, false, false>(a, s);
  return stash ? launch_cluster<TAD_CELL_GRU, true, false>(a, s) : launch_cluster<TAD_CELL_GRU, false, false>(a, s);
}

extern "C" int tad_rnn_fwd_ws(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                              void* workspace, int64_t workspace_bytes, tad_stream_t stream) {
  return rnn_fwd_ws_any("tad_rnn_fwd_ws", cell, prec, store, B, T, H, ndir, dirs, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int tad_rnn_fwd_ws_x(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                                const tad_rnn_xin_t* xin, void* workspace, int64_t workspace_bytes, tad_stream_t stream) {
  TAD_CHECK(xin, "tad_rnn_fwd_ws_x: null xin");
  return rnn_fwd_ws_any("tad_rnn_fwd_ws_x", cell, prec, store, B, T, H, ndir, dirs, xin, workspace, workspace_bytes, stream);
}

extern "C" int tad_rnn_ws_status(const void* workspace, tad_stream_t stream) {
  TAD_CHECK(workspace, "tad_rnn_ws_status: null workspace");
  unsigned v = 0;
  hipError_t e = hipMemcpyAsync(&v, reinterpret_cast<const unsigned char*>(workspace) + CL_STATUS_OFF, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { tad_set_error("tad_rnn_ws_status: %s", hipGetErrorString(e)); return TAD_ERR_LAUNCH; }
  if (v != 0) { tad_set_error("a wait between the workgroups of a cluster-form sequence launch ran out of time: its results are not valid"); return TAD_ERR_LAUNCH; }
  return TAD_OK;
}
