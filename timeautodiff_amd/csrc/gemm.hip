// nn.Linear family, fp32-stored operands (either matrix-core precision): entry points over the kernel in gemm_core.h.
#include "gemm_core.h"

extern "C" int tad_linear_fwd(int prec, int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx,
                              const float* W, int64_t ldw, const float* bias, const float* bias2,
                              const float* residual, int64_t ld_res, float* Y, int64_t ldy, int act,
                              tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_fwd: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "tad_linear_fwd: bad shape M=%ld N=%ld K=%ld ldx=%ld ldw=%ld ldy=%ld",
            (long)M, (long)N, (long)K, (long)ldx, (long)ldw, (long)ldy);
  TAD_CHECK(X && W && Y, "tad_linear_fwd: null pointer");
  TAD_CHECK(ldx < LD_MAX && ldw < LD_MAX && ldy < LD_MAX && ld_res < LD_MAX, "tad_linear_fwd: leading dimension too large");
  TAD_CHECK(act == TAD_ACT_NONE || act == TAD_ACT_RELU, "tad_linear_fwd: bad act %d", act);
  TAD_CHECK(!residual || ld_res >= N, "tad_linear_fwd: bad ld_res");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  p.A = X; p.sa_row = ldx; p.sa_k = 1;
  p.B = W; p.sb_row = ldw; p.sb_k = 1;
  p.C = Y; p.ldc = ldy; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.bias2 = bias2; p.act = act; p.residual = residual; p.ld_res = ld_res;
  p.k_per_split = cdiv64(K, BK) * BK;
  return launch<KC32, KC32>(prec, pick_cfg(prec, M, N, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_data(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy,
                                   const float* W, int64_t ldw, float* dX, int64_t ldx, const float* relu_of,
                                   int64_t ld_relu, int accumulate, tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_bwd_data: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldw >= K && ldx >= K, "tad_linear_bwd_data: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY && W && dX, "tad_linear_bwd_data: null pointer");
  TAD_CHECK(ldy < LD_MAX && ldw < LD_MAX && ldx < LD_MAX && ld_relu < LD_MAX, "tad_linear_bwd_data: leading dimension too large");
  TAD_CHECK(!relu_of || ld_relu >= K, "tad_linear_bwd_data: bad ld_relu");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dX[m,k] = sum_n dY[m,n] W[n,k]: output columns are k, the reduction runs over n
  p.A = dY; p.sa_row = ldy; p.sa_k = 1;
  p.B = W; p.sb_row = 1; p.sb_k = ldw;
  p.C = dX; p.ldc = ldx; p.M = M; p.N = K; p.K = N;
  p.relu_of = relu_of; p.ld_relu = ld_relu;
  if (accumulate) { p.residual = dX; p.ld_res = ldx; }
  p.k_per_split = cdiv64(N, BK) * BK;
  return launch<KC32, RC32>(prec, pick_cfg(prec, M, K, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_weight(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy,
                                     const float* X, int64_t ldx, float* dW, int64_t ldw, float* db, float* db2,
                                     int64_t T, int64_t tshift, const int32_t* gate_map, tad_stream_t stream) {
  TAD_CHECK(prec == TAD_F32 || prec == TAD_BF16, "tad_linear_bwd_weight: bad prec %d", prec);
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && ldw >= K, "tad_linear_bwd_weight: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY && X && dW, "tad_linear_bwd_weight: null pointer");
  TAD_CHECK(ldy < LD_MAX && ldw < LD_MAX && ldx < LD_MAX, "tad_linear_bwd_weight: leading dimension too large");
  TAD_CHECK(db || !db2, "tad_linear_bwd_weight: db2 without db");
  TAD_CHECK(M < (1ll << 31) && T >= 0 && T < (1 << 30) && (T == 0 || (M % T == 0 && tshift > -T && tshift < T)), "tad_linear_bwd_weight: bad T/tshift");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dW[n,k] = sum_m dY[m,n] X[m,k]: output rows n, columns k, reduction over m
  p.A = dY; p.sa_row = 1; p.sa_k = ldy;
  p.B = X; p.sb_row = 1; p.sb_k = ldx;
  p.C = dW; p.ldc = ldw; p.M = N; p.N = K; p.K = M;
  p.atomic = 1; p.shiftT = (int)T; p.shift = T > 0 ? (int)tshift : 0;
  p.rowsum = db; p.rowsum2 = db2;
  TAD_CHECK(set_gate_map(p, gate_map, N), "tad_linear_bwd_weight: bad gate map (N must be a multiple of 4*H, slots < G)");
  const Cfg cfg = pick_cfg_bwd_weight(prec, N, K);
  const int splits = plan_splits(p, cfg);
  return launch<RC32, RC32>(prec, cfg, p, splits, (hipStream_t)stream);
}

#ifdef EXP_STAMPS
// diagnostic build only: summed per-workgroup cycle counts {prologue, load issue, mfma, wait+lds store, barrier, epilogue, total, n}
extern "C" int tad_dbg_stamps(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 64);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, 64); }
  return 0;
}
#endif
