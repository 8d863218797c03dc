// MFMA GEMM for the nn.Linear family on gfx950: C[i,j] = sum_k A(i,k) * B(j,k).
//
// One kernel serves the three products of a linear layer by describing each operand
// with two strides: "k-contiguous" (activation rows, weight rows) or "row-contiguous"
// (the same memory read as its transpose).  Tiles are staged global -> registers ->
// LDS as [row][k] images, so the MFMA fragment reads are identical in all cases.
//
//   workgroup = 256 threads = 4 waves, tile 128 (i) x 64 (j), k-step 32
//   wave w owns rows [32w, 32w+32) x 64 columns = two 32x32 accumulators
//   fp32 : v_mfma_f32_32x32x2_f32, LDS rows of 32+4 floats
//   bf16 : v_mfma_f32_32x32x16_bf16, LDS rows of 32+8 bf16 (operands rounded on staging)
// Both paddings make the 16-byte fragment reads bank-conflict free (row stride = 4*odd dwords).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDF = BK + 4;   // floats per LDS row (fp32 path)
constexpr int LDH = BK + 8;   // bf16 per LDS row (bf16 path)

struct GemmParams {
  const float* A; int64_t sa_row, sa_k;
  const float* B; int64_t sb_row, sb_k;
  float* C; int64_t ldc;
  int64_t M, N, K;
  const float* bias; const float* bias2;
  const float* relu_of; int64_t ld_relu;
  const float* residual; int64_t ld_res;
  int act, atomic;
  int64_t k_per_split;
  int64_t shiftT, shift;     // time-shifted rows of the B operand along k (see header)
  int64_t tiles_n;
};

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16_t x = (bf16_t)a, y = (bf16_t)b;
  uint16_t ux = __builtin_bit_cast(uint16_t, x), uy = __builtin_bit_cast(uint16_t, y);
  return (uint32_t)ux | ((uint32_t)uy << 16);
}

// ---- k-contiguous operand: tile R rows x 32 k; thread -> (row = tid/8 + 32*it, 4 floats at k = 4*(tid%8))
template <int R>
__device__ __forceinline__ void load_kc(float* reg, const float* __restrict__ P, int64_t s_row, int64_t row0,
                                        int64_t rows, int64_t k0, int64_t kend, int tid, bool vec_ok) {
  const int c4 = tid & 7, r = tid >> 3;
#pragma unroll
  for (int it = 0; it < R / 32; ++it) {
    const int64_t row = row0 + r + 32 * it;
    const int64_t k = k0 + 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
      const float* p = P + row * s_row + k;
      if (vec_ok && k + 3 < kend) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (k + 0 < kend) v.x = p[0];
        if (k + 1 < kend) v.y = p[1];
        if (k + 2 < kend) v.z = p[2];
        if (k + 3 < kend) v.w = p[3];
      }
    }
    reg[4 * it + 0] = v.x; reg[4 * it + 1] = v.y; reg[4 * it + 2] = v.z; reg[4 * it + 3] = v.w;
  }
}

template <int R, bool BF16>
__device__ __forceinline__ void store_kc(const float* reg, void* lds, int tid) {
  const int c4 = tid & 7, r = tid >> 3;
#pragma unroll
  for (int it = 0; it < R / 32; ++it) {
    const int row = r + 32 * it;
    if constexpr (BF16) {
      uint2 u;
      u.x = pack_bf16x2(reg[4 * it + 0], reg[4 * it + 1]);
      u.y = pack_bf16x2(reg[4 * it + 2], reg[4 * it + 3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(lds) + row * LDH + 4 * c4) = u;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(lds) + row * LDF + 4 * c4) =
          make_float4(reg[4 * it + 0], reg[4 * it + 1], reg[4 * it + 2], reg[4 * it + 3]);
    }
  }
}

// ---- row-contiguous operand (element (row,k) at P[k*s_k + row]): thread -> one row, NKR consecutive k
template <int R>
__device__ __forceinline__ void load_rc(float* reg, const float* __restrict__ P, int64_t s_k, int64_t row0,
                                        int64_t rows, int64_t k0, int64_t kend, int tid, int64_t shiftT,
                                        int64_t shift) {
  constexpr int NKR = BK * R / 256;      // 16 for R=128, 8 for R=64
  const int r = tid % R, kq = tid / R;
  const int64_t row = row0 + r;
#pragma unroll
  for (int e = 0; e < NKR; ++e) {
    int64_t k = k0 + kq * NKR + e;
    float v = 0.f;
    if (row < rows && k < kend) {
      bool ok = true;
      if (shiftT > 0) {
        const int64_t t = k % shiftT + shift;
        ok = (t >= 0 && t < shiftT);
        k += shift;
      }
      if (ok) v = P[k * s_k + row];
    }
    reg[e] = v;
  }
}

template <int R, bool BF16>
__device__ __forceinline__ void store_rc(const float* reg, void* lds, int tid) {
  constexpr int NKR = BK * R / 256;
  const int r = tid % R, kq = tid / R;
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < NKR / 8; ++j) {
      uint4 u;
      u.x = pack_bf16x2(reg[8 * j + 0], reg[8 * j + 1]);
      u.y = pack_bf16x2(reg[8 * j + 2], reg[8 * j + 3]);
      u.z = pack_bf16x2(reg[8 * j + 4], reg[8 * j + 5]);
      u.w = pack_bf16x2(reg[8 * j + 6], reg[8 * j + 7]);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(lds) + r * LDH + kq * NKR + 8 * j) = u;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NKR / 4; ++j) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(lds) + r * LDF + kq * NKR + 4 * j) =
          make_float4(reg[4 * j + 0], reg[4 * j + 1], reg[4 * j + 2], reg[4 * j + 3]);
    }
  }
}

template <bool BF16, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int ESZ = BF16 ? 2 : 4;
  constexpr int LD = BF16 ? LDH : LDF;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LD * ESZ];
  void* As = smem;
  void* Bs = smem + BM * LD * ESZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int64_t i0 = tm * BM, j0 = tn * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;

  const bool a_vec = A_KC && (p.sa_row % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool b_vec = B_KC && (p.sb_row % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  float ra[16], rb[8];
  auto gload = [&](int64_t k0) {
    if constexpr (A_KC) load_kc<BM>(ra, p.A, p.sa_row, i0, p.M, k0, kend, tid, a_vec);
    else load_rc<BM>(ra, p.A, p.sa_k, i0, p.M, k0, kend, tid, 0, 0);
    if constexpr (B_KC) load_kc<BN>(rb, p.B, p.sb_row, j0, p.N, k0, kend, tid, b_vec);
    else load_rc<BN>(rb, p.B, p.sb_k, j0, p.N, k0, kend, tid, p.shiftT, p.shift);
  };
  auto lstore = [&]() {
    if constexpr (A_KC) store_kc<BM, BF16>(ra, As, tid); else store_rc<BM, BF16>(ra, As, tid);
    if constexpr (B_KC) store_kc<BN, BF16>(rb, Bs, tid); else store_rc<BN, BF16>(rb, Bs, tid);
  };

  if (kbeg < kend) gload(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();              // previous tile's fragment reads are done
    lstore();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);   // prefetch the next tile under the MFMAs
    const int arow = 32 * wave + (lane & 31), half = lane >> 5;
    if constexpr (BF16) {
      const bf16_t* a16 = reinterpret_cast<const bf16_t*>(As);
      const bf16_t* b16 = reinterpret_cast<const bf16_t*>(Bs);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a16 + arow * LDH + 16 * kk + 8 * half);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(b16 + (32 * n + (lane & 31)) * LDH + 16 * kk + 8 * half);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        }
      }
    } else {
      const float* a32 = reinterpret_cast<const float*>(As);
      const float* b32 = reinterpret_cast<const float*>(Bs);
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a32 + arow * LDF + 8 * kk + 4 * half);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(b32 + (32 * n + (lane & 31)) * LDF + 8 * kk + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[n], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: accumulator (reg r, lane) -> C[i0 + 32*wave + acc_row32(r,lane)][j0 + 32*n + (lane&31)]
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int64_t j = j0 + 32 * n + (lane & 31);
    if (j >= p.N) continue;
    float bsum = 0.f;
    if (p.bias) bsum += p.bias[j];
    if (p.bias2) bsum += p.bias2[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = i0 + 32 * wave + acc_row32(r, lane);
      if (i >= p.M) continue;
      float* c = p.C + i * p.ldc + j;
      float v = acc[n][r];
      if (p.atomic) {
        atomicAdd(c, v);
        continue;
      }
      v += bsum;
      if (p.residual) v += p.residual[i * p.ld_res + j];
      if (p.act == TAD_ACT_RELU) v = fmaxf(v, 0.f);
      if (p.relu_of && !(p.relu_of[i * p.ld_relu + j] > 0.f)) v = 0.f;
      *c = v;
    }
  }
}

template <bool A_KC, bool B_KC>
int launch(int prec, const GemmParams& p, int splits, hipStream_t s) {
  dim3 grid((unsigned)(cdiv64(p.M, BM) * p.tiles_n), 1, (unsigned)splits);
  if (prec == TAD_BF16) hipLaunchKernelGGL((gemm_kernel<true, A_KC, B_KC>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<false, A_KC, B_KC>), grid, dim3(256), 0, s, p);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

}  // namespace

extern "C" int tad_linear_fwd(int prec, int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx,
                              const float* W, int64_t ldw, const float* bias, const float* bias2,
                              const float* residual, int64_t ld_res, float* Y, int64_t ldy, int act,
                              tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_fwd: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "tad_linear_fwd: bad shape M=%ld N=%ld K=%ld ldx=%ld ldw=%ld ldy=%ld",
            (long)M, (long)N, (long)K, (long)ldx, (long)ldw, (long)ldy);
  TAD_CHECK(X && W && Y, "tad_linear_fwd: null pointer");
  TAD_CHECK(act == TAD_ACT_NONE || act == TAD_ACT_RELU, "tad_linear_fwd: bad act %d", act);
  TAD_CHECK(!residual || ld_res >= N, "tad_linear_fwd: bad ld_res");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  p.A = X; p.sa_row = ldx; p.sa_k = 1;
  p.B = W; p.sb_row = ldw; p.sb_k = 1;
  p.C = Y; p.ldc = ldy; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.bias2 = bias2; p.act = act; p.residual = residual; p.ld_res = ld_res;
  p.k_per_split = cdiv64(K, BK) * BK; p.tiles_n = cdiv64(N, BN);
  return launch<true, true>(prec, p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_data(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy,
                                   const float* W, int64_t ldw, float* dX, int64_t ldx, const float* relu_of,
                                   int64_t ld_relu, int accumulate, tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_bwd_data: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldw >= K && ldx >= K, "tad_linear_bwd_data: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY && W && dX, "tad_linear_bwd_data: null pointer");
  TAD_CHECK(!relu_of || ld_relu >= K, "tad_linear_bwd_data: bad ld_relu");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dX[m,k] = sum_n dY[m,n] W[n,k]: output columns are k, the reduction runs over n
  p.A = dY; p.sa_row = ldy; p.sa_k = 1;
  p.B = W; p.sb_row = 1; p.sb_k = ldw;
  p.C = dX; p.ldc = ldx; p.M = M; p.N = K; p.K = N;
  p.relu_of = relu_of; p.ld_relu = ld_relu;
  if (accumulate) { p.residual = dX; p.ld_res = ldx; }
  p.k_per_split = cdiv64(N, BK) * BK; p.tiles_n = cdiv64(K, BN);
  return launch<true, false>(prec, p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_weight(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy,
                                     const float* X, int64_t ldx, float* dW, int64_t ldw, int64_t T,
                                     int64_t tshift, tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_bwd_weight: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && ldw >= K, "tad_linear_bwd_weight: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY && X && dW, "tad_linear_bwd_weight: null pointer");
  TAD_CHECK(T >= 0 && (T == 0 || (M % T == 0 && tshift > -T && tshift < T)), "tad_linear_bwd_weight: bad T/tshift");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dW[n,k] = sum_m dY[m,n] X[m,k]: output rows n, columns k, reduction over m
  p.A = dY; p.sa_row = 1; p.sa_k = ldy;
  p.B = X; p.sb_row = 1; p.sb_k = ldx;
  p.C = dW; p.ldc = ldw; p.M = N; p.N = K; p.K = M;
  p.atomic = 1; p.shiftT = T; p.shift = tshift;
  p.tiles_n = cdiv64(K, BN);
  // split the long reduction so that roughly 1024 workgroups are in flight
  const int64_t tiles = cdiv64(N, BM) * p.tiles_n;
  int64_t splits = cdiv64(1024, tiles);
  const int64_t ksteps = cdiv64(M, BK);
  if (splits > ksteps) splits = ksteps;
  if (splits < 1) splits = 1;
  p.k_per_split = cdiv64(ksteps, splits) * BK;
  splits = cdiv64(M, p.k_per_split);
  return launch<false, false>(prec, p, (int)splits, (hipStream_t)stream);
}
