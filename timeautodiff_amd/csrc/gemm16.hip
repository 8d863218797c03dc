// nn.Linear family with bf16-STORED operands (bf16 matrix-core precision only): the input-side products of the
// recurrent layers, whose big tensors (pre-activations, their gradients) and weight images live in HBM as bf16.
// bf16 operands are staged into LDS in 16-byte chunks without touching the values (see gemm_core.h).
#include "gemm_core.h"

namespace {
bool rows16(const void* p, int64_t ld) { return p && aligned16(p) && ld % 8 == 0; }
}  // namespace

extern "C" int tad_linear_fwd_b16(int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx, const void* W16,
                                  int64_t ldw, const float* bias, void* Y, int y_dtype, int64_t ldy, int act,
                                  tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "tad_linear_fwd_b16: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(X && W16 && Y, "tad_linear_fwd_b16: null pointer");
  TAD_CHECK(ldx < LD_MAX && ldw < LD_MAX && ldy < LD_MAX, "tad_linear_fwd_b16: leading dimension too large");
  TAD_CHECK(act == TAD_ACT_NONE || act == TAD_ACT_RELU, "tad_linear_fwd_b16: bad act %d", act);
  TAD_CHECK(y_dtype == TAD_F32 || y_dtype == TAD_BF16, "tad_linear_fwd_b16: bad y_dtype %d", y_dtype);
  TAD_CHECK(rows16(W16, ldw) && K % 8 == 0, "tad_linear_fwd_b16: W16 needs 16-byte aligned rows (ldw %% 8 == 0) and K %% 8 == 0");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  p.A = X; p.sa_row = ldx; p.sa_k = 1;
  p.B = W16; p.sb_row = ldw; p.sb_k = 1;
  p.C = Y; p.ldc = ldy; p.c16 = y_dtype == TAD_BF16; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.act = act;
  p.k_per_split = cdiv64(K, BK) * BK;
  TAD_CHECK(!p.c16 || wide_epilogue_ok(p), "tad_linear_fwd_b16: a bf16 Y needs N %% 4 == 0, ldy %% 4 == 0 and 16-byte aligned Y and bias");
  return launch_bf16<KC32, KC16>(pick_cfg(TAD_BF16, M, N, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_fwd_b16x(int64_t M, int64_t N, int64_t K, const void* X16, int64_t ldx, const void* W16, int64_t ldw,
                                   const float* bias, void* Y, int y_dtype, int64_t ldy, int act, tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "tad_linear_fwd_b16x: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(X16 && W16 && Y, "tad_linear_fwd_b16x: null pointer");
  TAD_CHECK(ldx < LD_MAX && ldw < LD_MAX && ldy < LD_MAX, "tad_linear_fwd_b16x: leading dimension too large");
  TAD_CHECK(act == TAD_ACT_NONE || act == TAD_ACT_RELU, "tad_linear_fwd_b16x: bad act %d", act);
  TAD_CHECK(y_dtype == TAD_F32 || y_dtype == TAD_BF16, "tad_linear_fwd_b16x: bad y_dtype %d", y_dtype);
  TAD_CHECK(rows16(X16, ldx) && rows16(W16, ldw) && K % 8 == 0,
            "tad_linear_fwd_b16x: X16 and W16 need 16-byte aligned rows (ld %% 8 == 0) and K %% 8 == 0");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  p.A = X16; p.sa_row = ldx; p.sa_k = 1;
  p.B = W16; p.sb_row = ldw; p.sb_k = 1;
  p.C = Y; p.ldc = ldy; p.c16 = y_dtype == TAD_BF16; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.act = act;
  p.k_per_split = cdiv64(K, BK) * BK;
  TAD_CHECK(!p.c16 || wide_epilogue_ok(p), "tad_linear_fwd_b16x: a bf16 Y needs N %% 4 == 0, ldy %% 4 == 0 and 16-byte aligned Y and bias");
  return launch_bf16<KC16, KC16>(pick_cfg(TAD_BF16, M, N, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_data_b16(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const void* W16,
                                       int64_t ldw, float* dX, int64_t ldx, const float* relu_of, int64_t ld_relu,
                                       int accumulate, tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldw >= K && ldx >= K, "tad_linear_bwd_data_b16: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY16 && W16 && dX, "tad_linear_bwd_data_b16: null pointer");
  TAD_CHECK(ldy < LD_MAX && ldw < LD_MAX && ldx < LD_MAX && ld_relu < LD_MAX, "tad_linear_bwd_data_b16: leading dimension too large");
  TAD_CHECK(!relu_of || ld_relu >= K, "tad_linear_bwd_data_b16: bad ld_relu");
  TAD_CHECK(rows16(dY16, ldy) && N % 8 == 0, "tad_linear_bwd_data_b16: dY16 needs 16-byte aligned rows (ldy %% 8 == 0) and N %% 8 == 0");
  TAD_CHECK(rows16(W16, ldw) && K % 8 == 0, "tad_linear_bwd_data_b16: W16 needs 16-byte aligned rows (ldw %% 8 == 0) and K %% 8 == 0");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dX[m,k] = sum_n dY[m,n] W[n,k]: output columns are k, the reduction runs over n
  p.A = dY16; p.sa_row = ldy; p.sa_k = 1;
  p.B = W16; p.sb_row = 1; p.sb_k = ldw;
  p.C = dX; p.ldc = ldx; p.M = M; p.N = K; p.K = N;
  p.relu_of = relu_of; p.ld_relu = ld_relu;
  if (accumulate) { p.residual = dX; p.ld_res = ldx; }
  p.k_per_split = cdiv64(N, BK) * BK;
  return launch_bf16<KC16, RC16>(pick_cfg(TAD_BF16, M, K, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_data_b16o(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const void* W16,
                                        int64_t ldw, void* dX16, int64_t ldx, const void* relu_of16, int64_t ld_relu,
                                        tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldw >= K && ldx >= K, "tad_linear_bwd_data_b16o: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY16 && W16 && dX16, "tad_linear_bwd_data_b16o: null pointer");
  TAD_CHECK(ldy < LD_MAX && ldw < LD_MAX && ldx < LD_MAX && ld_relu < LD_MAX, "tad_linear_bwd_data_b16o: leading dimension too large");
  TAD_CHECK(!relu_of16 || ld_relu >= K, "tad_linear_bwd_data_b16o: bad ld_relu");
  TAD_CHECK(rows16(dY16, ldy) && N % 8 == 0, "tad_linear_bwd_data_b16o: dY16 needs 16-byte aligned rows (ldy %% 8 == 0) and N %% 8 == 0");
  TAD_CHECK(rows16(W16, ldw) && K % 8 == 0, "tad_linear_bwd_data_b16o: W16 needs 16-byte aligned rows (ldw %% 8 == 0) and K %% 8 == 0");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  p.A = dY16; p.sa_row = ldy; p.sa_k = 1;
  p.B = W16; p.sb_row = 1; p.sb_k = ldw;
  p.C = dX16; p.ldc = ldx; p.c16 = 1; p.M = M; p.N = K; p.K = N;
  p.relu_of = reinterpret_cast<const float*>(relu_of16); p.ld_relu = ld_relu; p.relu16 = 1;
  p.k_per_split = cdiv64(N, BK) * BK;
  TAD_CHECK(wide_epilogue_ok(p), "tad_linear_bwd_data_b16o: dX16 and relu_of16 need 16-byte aligned bases and leading dimensions that are multiples of 4");
  return launch_bf16<KC16, RC16>(pick_cfg(TAD_BF16, M, K, MIN_WGS), p, 1, (hipStream_t)stream);
}

extern "C" int tad_linear_bwd_weight_b16(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const float* X,
                                         int64_t ldx, float* dW, int64_t ldw, float* db, float* db2, int64_t T,
                                         int64_t tshift, const int32_t* gate_map, tad_stream_t stream) {
  TAD_CHECK(M >= 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && ldw >= K, "tad_linear_bwd_weight_b16: bad shape M=%ld N=%ld K=%ld",
            (long)M, (long)N, (long)K);
  TAD_CHECK(dY16 && X && dW, "tad_linear_bwd_weight_b16: null pointer");
  TAD_CHECK(ldy < LD_MAX && ldw < LD_MAX && ldx < LD_MAX, "tad_linear_bwd_weight_b16: leading dimension too large");
  TAD_CHECK(db || !db2, "tad_linear_bwd_weight_b16: db2 without db");
  TAD_CHECK(M < (1ll << 31) && T >= 0 && T < (1 << 30) && (T == 0 || (M % T == 0 && tshift > -T && tshift < T)),
            "tad_linear_bwd_weight_b16: bad T/tshift");
  TAD_CHECK(rows16(dY16, ldy) && N % 8 == 0, "tad_linear_bwd_weight_b16: dY16 needs 16-byte aligned rows (ldy %% 8 == 0) and N %% 8 == 0");
  if (M == 0) return TAD_OK;
  GemmParams p{};
  // dW[n,k] = sum_m dY[m,n] X[m,k]: output rows n, columns k, reduction over m
  p.A = dY16; p.sa_row = 1; p.sa_k = ldy;
  p.B = X; p.sb_row = 1; p.sb_k = ldx;
  p.C = dW; p.ldc = ldw; p.M = N; p.N = K; p.K = M;
  p.atomic = 1; p.shiftT = (int)T; p.shift = T > 0 ? (int)tshift : 0;
  p.rowsum = db; p.rowsum2 = db2;
  TAD_CHECK(set_gate_map(p, gate_map, N), "tad_linear_bwd_weight_b16: bad gate map (N must be a multiple of 4*H, slots < G)");
  const Cfg cfg = pick_cfg_bwd_weight(TAD_BF16, N, K);
  const int splits = plan_splits(p, cfg);
  return launch_bf16<RC16, RC32>(cfg, p, splits, (hipStream_t)stream);
}
