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

// Diagnostic build only (-DEXP_STAMPS): per-phase cycle counts of wave 0, summed over workgroups and steps.
#ifdef EXP_STAMPS
__device__ unsigned long long g_rstamps[8];
#define RSTAMP(var) __builtin_amdgcn_sched_barrier(0); unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0)
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
__global__ void pack_kernel(PackArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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

// ---------------------------------------------------------------------------
// acc[n] += A(32 x K from LDS) * B(block stream): blocks ordered [kk][n], 64 uint4 each.
template <bool BF16, int NT, int PF>
__device__ __forceinline__ void mma_stream(f32x16* acc, const unsigned char* lds_a, int lds_ld,
                                           const uint4* __restrict__ wp, int nk, int lane) {
  const int row = lane & 31, half = lane >> 5;
  uint4 ring[PF][NT];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < nk) {
#pragma unroll
      for (int n = 0; n < NT; ++n) ring[p][n] = wp[((int64_t)p * NT + n) * 64 + lane];
    }
  for (int kk0 = 0; kk0 < nk; kk0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kk = kk0 + p;
      if (kk < nk) {
        if constexpr (BF16) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(lds_a + ((size_t)row * lds_ld + 16 * kk + 8 * half) * 2);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const bf16x8 bv = __builtin_bit_cast(bf16x8, ring[p][n]);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[n], 0, 0, 0);
          }
        } else {
          const f32x4 av = *reinterpret_cast<const f32x4*>(lds_a + ((size_t)row * lds_ld + 8 * kk + 4 * half) * 4);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const f32x4 bv = __builtin_bit_cast(f32x4, ring[p][n]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[n], 0, 0, 0);
          }
        }
        if (kk + PF < nk) {
#pragma unroll
          for (int n = 0; n < NT; ++n) ring[p][n] = wp[((int64_t)(kk + PF) * NT + n) * 64 + lane];
        }
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
struct RnnFwdArgs {
  tad_rnn_dir_t d[MAXDIR];
  int B, T, H, NJ, nk, lds_ld, rpw;
};

template <int CELL, bool BF16, int JPW>
__global__ __launch_bounds__(512) void rnn_fwd_kernel(RnnFwdArgs a) {
  constexpr int G = CELL == TAD_CELL_LSTM ? 4 : 3;
  constexpr int S = CELL == TAD_CELL_LSTM ? 5 : 4;
  constexpr int ESZ = BF16 ? 2 : 4;
  constexpr int PF = BF16 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const tad_rnn_dir_t d = a.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = lane & 31;
  const int64_t b0 = (int64_t)blockIdx.x * a.rpw;   // rows >= rpw of the 32-row MFMA tile stay empty
  const size_t buf_bytes = (size_t)32 * a.lds_ld * ESZ;
  const int H = a.H, T = a.T;
  const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed);

  for (size_t i = threadIdx.x; i < 2 * buf_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;

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
  // offset shared by all 16 rows); lanes 0-31 / 32-63 each cover one full 128-byte line.
  const int half = lane >> 5;
  // input-side pre-activations of one step for chunk slot jj (the accumulator seeds)
  auto load_xp = [&](int step_, int jj, f32x16* acc, float* xn) {
    const int tt_ = d.reverse ? T - 1 - step_ : step_;
    const int u = 32 * (wave + jj * nw) + col;
    const bool uok = u < H;
    const int voff = (4 * half * T + tt_) * (int)d.ld_xproj + u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      const bool ok = uok && (cr + 4 * half) < a.rpw && (b0 + cr + 4 * half) < a.B;
      const float* xp = d.xproj + (b0 + cr) * (int64_t)T * d.ld_xproj;
      if constexpr (CELL == TAD_CELL_LSTM) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][r] = ok ? xp[voff + g * H] : 0.f;
      } else {
        acc[0][r] = ok ? xp[voff] : 0.f;            // b_hh is added in the epilogue so that the
        acc[1][r] = ok ? xp[voff + H] : 0.f;        // prefetched values stay untouched (in flight)
        acc[2][r] = 0.f;
        xn[r] = ok ? xp[voff + 2 * H] : 0.f;
      }
    }
  };

  // With one chunk per wave the next step's xproj is fetched right after this step's MFMAs,
  // so its HBM latency hides under the gate math, the stores and the barrier.
  constexpr bool PREFETCH = (JPW == 1);
  f32x16 nacc[G];
  float nxn[16];
  if constexpr (PREFETCH) {
    if (wave < a.NJ) load_xp(0, 0, nacc, nxn);
  }

#ifdef EXP_STAMPS
  unsigned long long stamp_prev = 0;
#endif
  for (int step = 0; step < T; ++step) {
    const int tt = d.reverse ? T - 1 - step : step;
    const unsigned char* cur = smem + (size_t)(step & 1) * buf_bytes;
    unsigned char* nxt = smem + (size_t)((step + 1) & 1) * buf_bytes;
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int j = wave + jj * nw;
      if (j < a.NJ) {
        const int u = 32 * j + col;
        const bool uok = u < H;
        f32x16 acc[G];
        float xn[16];
        RSTAMP(q0);
        if constexpr (PREFETCH) {
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = nacc[g];
#pragma unroll
          for (int r = 0; r < 16; ++r) xn[r] = nxn[r];
#ifdef EXP_STAMPS
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the prefetched values have landed
#endif
        } else {
          load_xp(step, jj, acc, xn);
        }
        RSTAMP(q1);
        mma_stream<BF16, G, PF>(acc, cur, a.lds_ld, wp + (int64_t)j * a.nk * G * 64, a.nk, lane);
        RSTAMP(q2);
        if constexpr (PREFETCH) {
          if (step + 1 < T) load_xp(step + 1, 0, nacc, nxn);
        }
        RSTAMP(q3);
#ifdef EXP_STAMPS
        if (threadIdx.x == 0) { atomicAdd(&g_rstamps[0], q1 - q0); atomicAdd(&g_rstamps[1], q2 - q1); atomicAdd(&g_rstamps[2], q3 - q2); }
        stamp_prev = q3;
#endif
        const int ooff = (4 * half * T + tt) * (int)d.ld_out + u;
        const int soff = (4 * half * T + tt) * (S * H) + u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const int row = cr + 4 * half;
          const bool ok = uok && row < a.rpw && (b0 + row) < a.B;
          float h;
          float sv[S];
          if constexpr (CELL == TAD_CELL_LSTM) {
            const float ig = fast_sigmoid(acc[0][r]), fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]), og = fast_sigmoid(acc[3][r]);
            const float c = fg * st[jj][r] + ig * gg;
            st[jj][r] = c;
            h = og * fast_tanh(c);
            sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og; sv[4] = c;
          } else {
            const float rg = fast_sigmoid(acc[0][r] + bh[jj][0]), zg = fast_sigmoid(acc[1][r] + bh[jj][1]);
            const float hpn = acc[2][r] + bh[jj][2];
            const float ng = fast_tanh(xn[r] + rg * hpn);
            h = (1.f - zg) * ng + zg * st[jj][r];
            st[jj][r] = h;
            sv[0] = rg; sv[1] = zg; sv[2] = ng; sv[3] = hpn;
          }
          if (!ok) h = 0.f;
          lds_put<BF16>(nxt, a.lds_ld, row, u, h);
          if (ok) {
            (d.out + (b0 + cr) * (int64_t)T * d.ld_out)[ooff] = h;
            if (d.stash) {
              float* sp = d.stash + (b0 + cr) * (int64_t)T * (S * H);
#pragma unroll
              for (int q = 0; q < S; ++q) sp[soff + q * H] = sv[q];
            }
          }
        }
      }
    }
#ifdef EXP_STAMPS
    { RSTAMP(q4); if (threadIdx.x == 0) atomicAdd(&g_rstamps[3], q4 - stamp_prev); stamp_prev = q4; }
#endif
    __syncthreads();
#ifdef EXP_STAMPS
    { RSTAMP(q5); if (threadIdx.x == 0) { atomicAdd(&g_rstamps[4], q5 - stamp_prev); atomicAdd(&g_rstamps[7], 1ull); } }
#endif
  }
}

// ---------------------------------------------------------------------------
struct RnnBwdArgs {
  tad_rnn_bwd_dir_t d[MAXDIR];
  int B, T, H, NJ, width, lds_ld, nk_total, nk_gate, rpw;
};

template <int CELL, bool BF16, int JPW, bool PHASED>
__global__ __launch_bounds__(512) void rnn_bwd_kernel(RnnBwdArgs a) {
  constexpr int G = CELL == TAD_CELL_LSTM ? 4 : 3;
  constexpr int S = CELL == TAD_CELL_LSTM ? 5 : 4;
  constexpr int PF = BF16 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const tad_rnn_bwd_dir_t d = a.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = lane & 31;
  const int64_t b0 = (int64_t)blockIdx.x * a.rpw;   // rows >= rpw of the 32-row MFMA tile stay empty
  const int H = a.H, T = a.T;
  const uint4* wp = reinterpret_cast<const uint4*>(d.w_packed_t);
  const float* dsrc = (CELL == TAD_CELL_GRU) ? d.dgates_h : d.dgates;   // operand of the recurrent product
  const int half = lane >> 5;

  f32x16 acc[JPW];
  float dcs[JPW][16];
#pragma unroll
  for (int jj = 0; jj < JPW; ++jj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[jj][r] = 0.f; dcs[jj][r] = 0.f; }

  for (int step = 0; step < T; ++step) {
    const int n = T - 1 - step;                       // forward step being undone
    const int tt = d.reverse ? T - 1 - n : n;
    const int tprev = (n > 0) ? (d.reverse ? tt + 1 : tt - 1) : -1;
    // ---- phase 1: gate gradients for the owned (row, unit) elements
    // (wave-uniform row bases + one 32-bit per-lane offset per buffer, as in the forward)
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int j = wave + jj * nw;
      if (j < a.NJ) {
        const int u = 32 * j + col;
        const bool uok = u < H;
        const int rowT = 4 * half * T;
        const int doff = (rowT + tt) * (int)d.ld_dout + u;
        const int soff = (rowT + tt) * (S * H) + u;
        const int spoff = (rowT + tprev) * (S * H) + u;
        const int ooff = (rowT + tprev) * (int)d.ld_out + u;
        const int goff = (rowT + tt) * (int)d.ld_dgates + u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          const int row = cr + 4 * half;
          const bool ok = uok && row < a.rpw && (b0 + row) < a.B;
          float da[G];
          float nacc = 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) da[g] = 0.f;
          if (ok) {
            const int64_t rb = (b0 + cr) * (int64_t)T;
            const float dh = (d.dout + rb * d.ld_dout)[doff] + acc[jj][r];
            const float* sp = d.stash + rb * (S * H);
            if constexpr (CELL == TAD_CELL_LSTM) {
              const float ig = sp[soff], fg = sp[soff + H], gg = sp[soff + 2 * H], og = sp[soff + 3 * H], c = sp[soff + 4 * H];
              const float cprev = (tprev >= 0) ? sp[spoff + 4 * H] : 0.f;
              const float tc = fast_tanh(c);
              const float dog = dh * tc;
              const float dc = dh * og * (1.f - tc * tc) + dcs[jj][r];
              dcs[jj][r] = dc * fg;
              da[0] = dc * gg * ig * (1.f - ig);
              da[1] = dc * cprev * fg * (1.f - fg);
              da[2] = dc * ig * (1.f - gg * gg);
              da[3] = dog * og * (1.f - og);
              float* dg = d.dgates + rb * d.ld_dgates;
#pragma unroll
              for (int g = 0; g < 4; ++g) dg[goff + g * H] = da[g];
            } else {
              const float rg = sp[soff], zg = sp[soff + H], ng = sp[soff + 2 * H], hpn = sp[soff + 3 * H];
              const float hprev = (tprev >= 0) ? (d.out + rb * d.ld_out)[ooff] : 0.f;
              const float dn = dh * (1.f - zg);
              const float dz = dh * (hprev - ng);
              const float dan = dn * (1.f - ng * ng);
              const float dar = dan * hpn * rg * (1.f - rg);
              const float daz = dz * zg * (1.f - zg);
              da[0] = dar; da[1] = daz; da[2] = dan * rg;
              float* dg = d.dgates + rb * d.ld_dgates;
              dg[goff] = dar; dg[goff + H] = daz; dg[goff + 2 * H] = dan;
              float* dgh = d.dgates_h + rb * d.ld_dgates;
              dgh[goff] = dar; dgh[goff + H] = daz; dgh[goff + 2 * H] = da[2];
              nacc = dh * zg;
            }
          }
          acc[jj][r] = nacc;
          if constexpr (!PHASED) {
#pragma unroll
            for (int g = 0; g < G; ++g) lds_put<BF16>(smem, a.lds_ld, row, g * a.width + u, da[g]);
          }
        }
      }
    }
    // ---- phase 2: dh_rec = dgates * W_hh (K = G * width), A operand from LDS
    if constexpr (!PHASED) {
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < JPW; ++jj) {
        const int j = wave + jj * nw;
        if (j < a.NJ) mma_stream<BF16, 1, PF>(&acc[jj], smem, a.lds_ld, wp + (int64_t)j * a.nk_total * 64, a.nk_total, lane);
      }
      __syncthreads();
    } else {
      for (int g = 0; g < G; ++g) {
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
              if (u < H && row < a.rpw && b < a.B) v = dsrc[(b * T + tt) * d.ld_dgates + (int64_t)g * H + u];
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
}

// ---------------------------------------------------------------------------
struct Geom {
  int NJ, width, nkf, nkb, jpw, nw;
};

// Rows of the batch one workgroup owns.  The MFMA tile is 32 rows; when the batch is too
// small to give every CU a workgroup, fewer real rows per workgroup spread the per-step
// HBM traffic (xproj in, h / stash out) over more CUs -- the step time is bound by the
// per-CU memory path, not by the matrix cores, in that regime.
int rows_per_wg(int64_t B, int ndir) {
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
  g.jpw = g.NJ <= 8 ? 1 : (g.NJ <= 16 ? 2 : 4);
  g.nw = (int)cdiv64(g.NJ, g.jpw);
  return g;
}

constexpr size_t LDS_LIMIT = 160 * 1024;

// raises the dynamic-LDS limit of a kernel; `configured` caches the largest size already set
template <typename K>
int set_lds(K kernel, size_t bytes, size_t& configured) {
  if (bytes <= configured) return TAD_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    tad_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", bytes, hipGetErrorString(e));
    return TAD_ERR_LAUNCH;
  }
  configured = bytes;
  return TAD_OK;
}

template <int CELL, bool BF16, int JPW>
int launch_fwd(const RnnFwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto k = rnn_fwd_kernel<CELL, BF16, JPW>;
  static size_t configured = 0;
  if (int rc = set_lds(k, lds, configured)) return rc;
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_fwd_kernel<%d, %s, %d>", CELL, BF16 ? "true" : "false", JPW);
  return TAD_OK;
}

template <int CELL, bool BF16, int JPW, bool PHASED>
int launch_bwd(const RnnBwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto k = rnn_bwd_kernel<CELL, BF16, JPW, PHASED>;
  static size_t configured = 0;
  if (int rc = set_lds(k, lds, configured)) return rc;
  hipLaunchKernelGGL(k, grid, block, lds, s, a);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("rnn_bwd_kernel<%d, %s, %d, %s>", CELL, BF16 ? "true" : "false", JPW, PHASED ? "true" : "false");
  return TAD_OK;
}

}  // namespace

extern "C" int64_t tad_rnn_packed_bytes(int prec, int64_t H, int G) {
  if (H <= 0 || (G != 3 && G != 4) || (prec != TAD_F32 && prec != TAD_BF16)) return -1;
  const Geom g = geom(prec, H, G);
  const int64_t f = (int64_t)g.NJ * g.nkf * G, b = (int64_t)g.NJ * g.nkb;
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
  return TAD_OK;
}

extern "C" int tad_rnn_fwd(int cell, int prec, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                           tad_stream_t stream) {
  TAD_CHECK(cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU, "tad_rnn_fwd: bad cell %d", cell);
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_fwd: bad prec %d", prec);
  TAD_CHECK(B > 0 && T > 0 && H > 0 && H <= 1024 && (ndir == 1 || ndir == 2) && dirs, "tad_rnn_fwd: bad shape B=%ld T=%ld H=%ld ndir=%d",
            (long)B, (long)T, (long)H, ndir);
  const int G = cell == TAD_CELL_LSTM ? 4 : 3;
  const bool bf = prec == TAD_BF16;
  const Geom g = geom(prec, H, G);
  RnnFwdArgs a{};
  for (int i = 0; i < ndir; ++i) {
    a.d[i] = dirs[i];
    TAD_CHECK(dirs[i].xproj && dirs[i].w_packed && dirs[i].out, "tad_rnn_fwd: null pointer in dir %d", i);
    TAD_CHECK(dirs[i].ld_xproj >= G * H && dirs[i].ld_out >= H, "tad_rnn_fwd: bad leading dimension in dir %d", i);
    TAD_CHECK((5 * T + 1) * (dirs[i].ld_xproj + dirs[i].ld_out + 5 * H) < (int64_t)1 << 31, "tad_rnn_fwd: T * ld too large for 32-bit lane offsets");
  }
  a.B = (int)B; a.T = (int)T; a.H = (int)H; a.NJ = g.NJ; a.nk = g.nkf;
  a.rpw = rows_per_wg(B, ndir);
  a.lds_ld = g.width + (bf ? 8 : 4);
  const size_t lds = (size_t)2 * 32 * a.lds_ld * (bf ? 2 : 4);
  TAD_CHECK(lds <= LDS_LIMIT, "tad_rnn_fwd: H=%ld needs %zu bytes of LDS in this precision (limit %zu)", (long)H, lds, LDS_LIMIT);
  dim3 grid((unsigned)cdiv64(B, a.rpw), (unsigned)ndir), block((unsigned)g.nw * 64);
  hipStream_t s = (hipStream_t)stream;
#define TAD_FWD(C, BFV)                                                                  \
  (g.jpw == 1 ? launch_fwd<C, BFV, 1>(a, grid, block, lds, s)                             \
              : (g.jpw == 2 ? launch_fwd<C, BFV, 2>(a, grid, block, lds, s) : launch_fwd<C, BFV, 4>(a, grid, block, lds, s)))
  if (cell == TAD_CELL_LSTM) return bf ? TAD_FWD(TAD_CELL_LSTM, true) : TAD_FWD(TAD_CELL_LSTM, false);
  return bf ? TAD_FWD(TAD_CELL_GRU, true) : TAD_FWD(TAD_CELL_GRU, false);
#undef TAD_FWD
}

extern "C" int tad_rnn_bwd(int cell, int prec, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_bwd_dir_t* dirs,
                           tad_stream_t stream) {
  TAD_CHECK(cell == TAD_CELL_LSTM || cell == TAD_CELL_GRU, "tad_rnn_bwd: bad cell %d", cell);
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_rnn_bwd: bad prec %d", prec);
  TAD_CHECK(B > 0 && T > 0 && H > 0 && H <= 1024 && (ndir == 1 || ndir == 2) && dirs, "tad_rnn_bwd: bad shape B=%ld T=%ld H=%ld ndir=%d",
            (long)B, (long)T, (long)H, ndir);
  const int G = cell == TAD_CELL_LSTM ? 4 : 3;
  const bool bf = prec == TAD_BF16;
  const Geom g = geom(prec, H, G);
  RnnBwdArgs a{};
  for (int i = 0; i < ndir; ++i) {
    a.d[i] = dirs[i];
    TAD_CHECK(dirs[i].dout && dirs[i].stash && dirs[i].w_packed_t && dirs[i].dgates, "tad_rnn_bwd: null pointer in dir %d", i);
    TAD_CHECK(cell == TAD_CELL_LSTM || (dirs[i].dgates_h && dirs[i].out), "tad_rnn_bwd: GRU needs dgates_h and out (dir %d)", i);
    TAD_CHECK(dirs[i].ld_dout >= H && (cell == TAD_CELL_LSTM || dirs[i].ld_out >= H) && dirs[i].ld_dgates >= G * H,
              "tad_rnn_bwd: bad leading dimension in dir %d", i);
  }
  const int ks = ks_of(bf);
  a.B = (int)B; a.T = (int)T; a.H = (int)H; a.NJ = g.NJ; a.width = g.width;
  a.rpw = rows_per_wg(B, ndir);
  a.nk_total = g.nkb; a.nk_gate = g.width / ks;
  const int pad = bf ? 8 : 4;
  const size_t esz = bf ? 2 : 4;
  size_t lds = (size_t)32 * (G * g.width + pad) * esz;
  const bool phased = lds > LDS_LIMIT;
  a.lds_ld = (phased ? g.width : G * g.width) + pad;
  lds = (size_t)32 * a.lds_ld * esz;
  TAD_CHECK(lds <= LDS_LIMIT, "tad_rnn_bwd: H=%ld needs %zu bytes of LDS (limit %zu)", (long)H, lds, LDS_LIMIT);
  dim3 grid((unsigned)cdiv64(B, a.rpw), (unsigned)ndir), block((unsigned)g.nw * 64);
  hipStream_t s = (hipStream_t)stream;
#define TAD_BWD2(C, BFV, J) (phased ? launch_bwd<C, BFV, J, true>(a, grid, block, lds, s) : launch_bwd<C, BFV, J, false>(a, grid, block, lds, s))
#define TAD_BWD(C, BFV) (g.jpw == 1 ? TAD_BWD2(C, BFV, 1) : (g.jpw == 2 ? TAD_BWD2(C, BFV, 2) : TAD_BWD2(C, BFV, 4)))
  if (cell == TAD_CELL_LSTM) return bf ? TAD_BWD(TAD_CELL_LSTM, true) : TAD_BWD(TAD_CELL_LSTM, false);
  return bf ? TAD_BWD(TAD_CELL_GRU, true) : TAD_BWD(TAD_CELL_GRU, false);
#undef TAD_BWD
#undef TAD_BWD2
}

#ifdef EXP_STAMPS
// diagnostic build only: forward kernel, wave 0, summed {seed+wait, mfma stream, prefetch issue, gate math+stores, barrier, -, -, steps}
extern "C" int tad_dbg_rnn_stamps(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rstamps), 64);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_rstamps), z, 64); }
  return 0;
}
#endif
