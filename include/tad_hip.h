/*
 * tad_hip.h -- C ABI of libtad_hip.so: hand-written gfx950 (MI355X) kernels for the
 * TimeAutoDiff hot path (denoiser training step, reverse-diffusion sampler, mixed-type
 * VAE forward/backward).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 data unless stated otherwise;
 *     tensors are row-major with an explicit leading dimension (ld*) in elements;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it, never
 *     synchronise and never allocate;
 *   - return value: TAD_OK or an error code; tad_last_error() gives the message of the
 *     calling thread's last failure;
 *   - `prec` selects the matrix-core operand type: TAD_F32 (v_mfma_f32_32x32x2_f32,
 *     exact fp32 products) or TAD_BF16 (v_mfma_f32_32x32x16_bf16, operands rounded to
 *     bf16, fp32 accumulation).  Everything outside the matrix products is fp32;
 *   - the big per-timestep tensors of the recurrent layers (input-side pre-activations,
 *     gate stash, pre-activation gradients) are stored in the "gate-interleaved" layout:
 *     column 4*u + s of a direction's block of 4*H columns holds slot s of hidden unit u
 *     (LSTM slots i,f,g,o; GRU r,z,n and a fourth slot described per tensor), so that a
 *     lane of the sequence kernels moves all slots of its unit in one access.  Their
 *     element type (`store`) is TAD_F32 or TAD_BF16 (bf16 only together with prec = TAD_BF16).
 *
 * Each entry point names the reference code it replaces (paths relative to the
 * TimeAutoDiff checkout).  The reference has no FFI layer of its own: these are the
 * ATen calls its nn.Module forwards / autograd / optimizer dispatch on this path.
 */
#ifndef TAD_HIP_H
#define TAD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tad_stream_t;

enum { TAD_OK = 0, TAD_ERR_INVALID = 1, TAD_ERR_LAUNCH = 2 };
enum { TAD_F32 = 0, TAD_BF16 = 1 };
enum { TAD_ACT_NONE = 0, TAD_ACT_RELU = 1, TAD_ACT_SILU = 2 /* tad_mm16_nt only */,
       TAD_ACT_DSILU = 3 /* tad_mm16_nt only: the result is multiplied by silu'(relu_of) instead of being masked by it */,
       TAD_ACT_SILU_Z32 = 4 /* tad_mm16_nt only: TAD_ACT_SILU whose pre-activation output Z holds float32 (what SiLU's backward reads) */ };
enum { TAD_CELL_LSTM = 0, TAD_CELL_GRU = 1 };

int tad_abi_version(void);
const char* tad_last_error(void);
/* Name (template arguments included, as a profiler prints it) of the device kernel that the calling
 * thread's most recent entry point launched, or "" if it launched none it names; reading clears it.
 * No counterpart in the reference: it lets a benchmark attribute device time per kernel. */
const char* tad_last_kernel(void);

/* ------------------------------------------------------------------ linear family
 * nn.Linear forward: Y[M,N] = act(X[M,K] W[N,K]^T + bias + bias2 + residual).
 * Replaces every nn.Linear / FeedForward on the path: DIFF.py:57-63, 141, 155-164,
 * 184, 188-194; VAE.py:80-86, 122-133, 143-145; and the input projections inside
 * nn.LSTM / nn.GRU (DIFF.py:182, 189; VAE.py:119-120).  bias, bias2 ([N]) and residual
 * ([M,N], row stride ld_res; may alias Y, e.g. the `x + cond_output(...)` sum of
 * DIFF.py:216) may be NULL. */
int tad_linear_fwd(int prec, int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx, const float* W,
                   int64_t ldw, const float* bias, const float* bias2, const float* residual, int64_t ld_res,
                   float* Y, int64_t ldy, int act, tad_stream_t stream);

/* autograd of nn.Linear wrt its input: dX[M,K] = dY[M,N] W[N,K] (+ dX if accumulate);
 * if relu_of != NULL the result is zeroed where relu_of[m,k] <= 0 (backward of the
 * ReLU that produced this layer's input, e.g. DIFF.py:62, 179, 193; VAE.py:129). */
int tad_linear_bwd_data(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy, const float* W,
                        int64_t ldw, float* dX, int64_t ldx, const float* relu_of, int64_t ld_relu,
                        int accumulate, tad_stream_t stream);

/* autograd of nn.Linear wrt its weight: dW[N,K] += dY[M,N]^T X'[M,K].  Always
 * accumulates (float atomics across M-slabs).  If T > 0 the rows of X are read
 * time-shifted inside windows of T rows: X'[b*T+t] = X[b*T+t+tshift] when
 * 0 <= t+tshift < T, else 0 -- the h_{t-1} operand of the recurrent weight gradient
 * in BPTT (autograd of DIFF.py:182, 189; VAE.py:119-120).  If db != NULL the bias
 * gradient db[n] += sum_m dY[m,n] is accumulated in the same pass (and into db2 as
 * well when that is given: the b_ih / b_hh pair of an LSTM receives the same sum).
 * gate_map (HOST array {H, G, slot0, slot1, slot2, slot3} or NULL): dY is a
 * gate-interleaved pre-activation gradient (N a multiple of 4*H; column blk*4H + 4u + s);
 * its product row lands in row blk*G*H + slot_s*H + u of dW (and db, db2), torch's
 * gate-major parameter order, or is dropped when slot_s < 0. */
int tad_linear_bwd_weight(int prec, int64_t M, int64_t N, int64_t K, const float* dY, int64_t ldy,
                          const float* X, int64_t ldx, float* dW, int64_t ldw, float* db, float* db2,
                          int64_t T, int64_t tshift, const int32_t* gate_map, tad_stream_t stream);

/* The same three products with bf16-STORED operands (bf16 matrix cores, fp32 accumulation):
 * the input-side products of nn.LSTM / nn.GRU (DIFF.py:182, 189; VAE.py:119-120) when the
 * pre-activations, their gradients and the weight images live in HBM as bf16.  bf16
 * operands need 16-byte aligned rows, leading dimensions and extents that are multiples of 8.
 *   fwd_b16:        Y = act(X W16^T + bias), X fp32, W16 bf16 [N,K], Y fp32 or bf16 (y_dtype);
 *   fwd_b16x:       the same with X16 stored as bf16 too, through the square-tiled kernel: for weights too large for
 *                   tad_mm16_nt's scheme, in which every workgroup streams the whole weight (the input side of a
 *                   recurrent layer at hidden size 512: 4096 x 1024);
 *   bwd_data_b16:   dX = dY16 W16 (+ dX) with the ReLU mask, dY16 bf16 [M,N], W16 bf16 [N,K], dX fp32;
 *   bwd_data_b16o:  the same product with a bf16 result dX16 and a bf16 mask operand (no accumulate): for a dX
 *                   that only matrix products read (dX16, relu_of16: 16-byte aligned, ld multiples of 4);
 *   bwd_weight_b16: dW += dY16^T X' with db/db2, T/tshift and gate_map as above; the bias
 *                   sums are taken over the bf16 values of dY16. */
int tad_linear_fwd_b16(int64_t M, int64_t N, int64_t K, const float* X, int64_t ldx, const void* W16, int64_t ldw,
                       const float* bias, void* Y, int y_dtype, int64_t ldy, int act, tad_stream_t stream);
int tad_linear_fwd_b16x(int64_t M, int64_t N, int64_t K, const void* X16, int64_t ldx, const void* W16, int64_t ldw,
                        const float* bias, void* Y, int y_dtype, int64_t ldy, int act, tad_stream_t stream);
int tad_linear_bwd_data_b16(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const void* W16,
                            int64_t ldw, float* dX, int64_t ldx, const float* relu_of, int64_t ld_relu,
                            int accumulate, tad_stream_t stream);
int tad_linear_bwd_data_b16o(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const void* W16,
                             int64_t ldw, void* dX16, int64_t ldx, const void* relu_of16, int64_t ld_relu,
                             tad_stream_t stream);
int tad_linear_bwd_weight_b16(int64_t M, int64_t N, int64_t K, const void* dY16, int64_t ldy, const float* X,
                              int64_t ldx, float* dW, int64_t ldw, float* db, float* db2, int64_t T,
                              int64_t tshift, const int32_t* gate_map, tad_stream_t stream);

/* Products with BOTH operands stored as bf16 (bf16 matrix cores, fp32 accumulation), staged global -> LDS by
 * direct-to-LDS loads.  Same nn.Linear call sites as above, for activations that are kept in HBM as bf16.
 *   C[M,N] = act(A16[M,K] W16[N,K]^T + bias + residual), then zeroed where relu_of[m,n] <= 0.
 * A16, W16: bf16, 16-byte aligned rows (K, lda, ldw multiples of 8), K <= 1024; N a multiple of 8.  W16 is a PADDED
 * image: allocated (readable) for N rounded up to a multiple of 64 rows and K rounded up to a multiple of 128
 * columns (ldw >= that), zeros in columns K.. of every row; rows N.. may hold anything.  bias fp32 [N] or NULL.  residual / relu_of: [M,N] operands of dtype res_dtype / relu_dtype (TAD_F32 or TAD_BF16) or NULL; residual
 * may alias C.  C: fp32 or bf16 (c_dtype), ldc a multiple of 4.  Z16 (optional, bf16 [M,N]): receives the value
 * before `act` (the pre-activation a SiLU's backward needs, DIFF.py:58 / VAE.py:81; float32 [M,N] with act =
 * SILU_Z32).  act: NONE, RELU, SILU, SILU_Z32, or DSILU (the result times silu'(relu_of): a SiLU's backward).
 * With W16 = the transposed weight image this is also the input gradient dX = dY W. */
int tad_mm16_nt(int64_t M, int64_t N, int64_t K, const void* A16, int64_t lda, const void* W16, int64_t ldw,
                const float* bias, const void* residual, int res_dtype, int64_t ld_res, const void* relu_of,
                int relu_dtype, int64_t ld_relu, void* C, int c_dtype, int64_t ldc, void* Z16, int64_t ldz, int act,
                tad_stream_t stream);

/* bf16 copy of an fp32 matrix: dst16[r][c] = bf16(src[r][c]) (round to nearest even), r < rows, c < cols; with
 * relu_of16 (bf16 [rows, cols], row stride ld_relu) the copy is zero where relu_of16 <= 0 (the ReLU backward of the
 * layer whose output relu_of16 is).  The bf16
 * operand of tad_mm16_nt / tad_mm16_tn for activations another kernel left in fp32 (no counterpart in the reference,
 * which has no reduced-precision mode: the rounding is the one every bf16 product applies to its operands). */
int tad_cvt_bf16(int64_t rows, int64_t cols, const float* src, int64_t ld_src, void* dst16, int64_t ld_dst,
                 const void* relu_of16, int64_t ld_relu, tad_stream_t stream);

typedef struct {
  const float* src;   /* parameter matrix W [n, k], row stride ld_src */
  int64_t ld_src;
  int64_t n, k;
  void* dst;          /* bf16 image: transpose == 0: dst[r][c] = W[r][c] (ld_dst >= k); else dst[c][r] = W[r][c] (ld_dst >= n) */
  int64_t ld_dst;
  int32_t transpose;
  int32_t pad_;
} tad_pack_w16_job_t;
/* The bf16 weight images tad_mm16_nt reads (forward: W itself; input gradient: W^T), any number of them in one
 * launch per 48 jobs.  Only the n x k elements are written: the padding of an image (rows up to a multiple of 64,
 * row stride a multiple of 128) keeps what its storage was initialised with (zeros).  No counterpart in the
 * reference: the images are the bf16 compute mode's copies of the nn.Linear / nn.LSTM / nn.GRU weights. */
int tad_pack_w16(int njobs, const tad_pack_w16_job_t* jobs, tad_stream_t stream);

/* Weight gradients with both operands stored as bf16 (autograd of nn.Linear / nn.LSTM / nn.GRU wrt their weights,
 * same call sites as tad_linear_bwd_weight): for every job k
 *   C_k[map(a), b] += sum_m A16[m, a0_k + a] * B16_k[m + shift_k, b],   a < na_k, b < nb_k,
 * where a row m + shift_k that leaves its window of T rows (0 <= (m mod T) + shift_k < T fails) reads as zero: the
 * h_{t-1} operand of BPTT.  All jobs of a call share A16 (the gate gradients of a recurrent layer feed dW_ih and both
 * dW_hh), which is then read once from HBM.  db / db2 (or NULL) += the column sums of the job's columns of A16.
 * gate_map = {H, G, slot0..slot3} maps product row blk*4H + 4u + s to row blk*G*H + slot_s*H + u of C_k / db
 * (dropped when slot_s < 0); gate_map[0] == 0: rows as they are.  A16, B16: bf16, 16-byte aligned rows; a0, nb
 * multiples of 8; na any (a0 + na rounded up to 8 must stay inside lda: whole 16-byte pieces are read).  C_k fp32
 * (float atomics: sums differ in the last bits from run to run).  At most 12 jobs; the launch's
 * A16 may be NULL when every job brings its own.  A weight with fewer than a multiple of 8 inputs (Linear(6 -> H))
 * passes B16 padded with zero columns up to nb and ldc_k = its true width, nb_k - 8 < ldc_k < nb_k: the columns of the
 * product beyond ldc_k are not written. */
typedef struct {
  const void* B16; int64_t ldb; int64_t nb; int64_t shift;
  float* C; int64_t ldc; float* db; float* db2;
  int64_t a0, na;
  int32_t gate_map[6];
  const void* A16;       /* NULL: the launch's A16 / lda; else this job's own A operand [M, lda] (the small products of
                            a backward pass -- different layers, same number of rows -- then share one launch) */
  int64_t lda;
} tad_mm16_tn_job_t;
int tad_mm16_tn(int64_t M, int64_t T, const void* A16, int64_t lda, int njobs, const tad_mm16_tn_job_t* jobs, tad_stream_t stream);

/* Column sums: out_sum[n] += sum_m dY[m,n]; if out_wsum != NULL also
 * out_wsum[n] += sum_m dY[m,n] * row_scale[m*row_scale_stride] * scale_mul.
 * Bias gradients of every nn.Linear / RNN and the weight gradient of the
 * Linear(1 -> H/2) inside PositionalEncoding (DIFF.py:141-145). */
int tad_colsum(int64_t M, int64_t N, const float* dY, int64_t ldy, const float* row_scale,
               int64_t row_scale_stride, float scale_mul, float* out_sum, float* out_sum2, float* out_wsum,
               tad_stream_t stream);

/* Gradients of the linear halves of the two positional encodings (reference DIFF.py: PositionalEncoding.linear_proj of
 * t_enc and i_enc, whose outputs are columns [col1, col1+L) and [col2, col2+L) of the input of proj, weight W [N, ldw]
 * fp32) from g16 = the bf16-stored gradient of proj's pre-activations [M, N]: gb += W[:, col..]^T sum_r g[r],
 * gw += mul * W[:, col..]^T sum_r s[r] g[r], s1 / s2 the per-row scalars the encodings were evaluated at.  Equal to
 * the column sums of g W[:, col..] without forming that product.  round_w != 0: W is rounded to bf16 first (the operand
 * of the bf16 mode).  sums: 3 N floats of workspace.  N a multiple of 8, at most 512; ldg a multiple of 8. */
int tad_enc_grad_b16(int64_t M, int N, const void* g16, int64_t ldg, const float* s1, int64_t s1_stride, const float* s2,
                     int64_t s2_stride, const float* W, int64_t ldw, int col1, int col2, int L, float mul1, float mul2,
                     int round_w, float* sums, float* gb1, float* gw1, float* gb2, float* gw2, tad_stream_t stream);

/* ------------------------------------------------------------------ recurrent family
 * Per-direction arguments of the persistent LSTM / GRU sequence kernels.  xproj holds
 * the input-side pre-activations x_t W_ih^T (+ b_ih, and + b_hh for LSTM) for all
 * time steps, gate-interleaved; the kernel adds h_{t-1} W_hh^T, applies the gate
 * nonlinearities and walks T steps with h (and c) resident on chip.  Typed pointers
 * (void*) hold `store` elements; they need 16-byte aligned bases and leading
 * dimensions that are multiples of 4. */
typedef struct {
  const void* xproj;    /* [B*T, 4H] of this direction (slots i,f,g,o / r,z,n,unused), row stride ld_xproj */
  int64_t ld_xproj;
  const void* w_packed; /* from tad_rnn_pack_weights (forward image) */
  const float* b_hh;    /* GRU: recurrent bias [3H] (r,z,n); LSTM: NULL */
  float* out;           /* h_t -> out[(b*T+t)*ld_out + u], u < H; may be NULL when out16 is given */
  int64_t ld_out;
  void* gates;          /* training: [B*T, 4H] gate stash (LSTM i,f,g,o; GRU r,z,n,W_hn h+b_hn), row stride ld_gates; NULL = inference */
  int64_t ld_gates;
  float* cstate;        /* LSTM training: c_t -> cstate[(b*T+t)*ld_c + u] (fp32); GRU: NULL */
  int64_t ld_c;
  int32_t reverse;      /* 0: t = 0..T-1, 1: t = T-1..0 */
  int32_t pad_;
  void* out16;          /* optional: h_t once more, rounded to bf16 (the operand of the matrix products that read h), row stride ld_out16 */
  int64_t ld_out16;
} tad_rnn_dir_t;

/* nn.LSTM / nn.GRU forward over one layer, ndir (1 or 2) independent directions or
 * heads in one launch (DIFF.py:182 -> 220, DIFF.py:189 -> 215, VAE.py:119-120 -> 154-155). */
int tad_rnn_fwd(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                tad_stream_t stream);

typedef struct {
  const float* dout;      /* grad wrt h_t outputs, row stride ld_dout */
  int64_t ld_dout;
  const void* gates;      /* written by tad_rnn_fwd */
  int64_t ld_gates;
  const float* cstate;    /* LSTM: written by tad_rnn_fwd */
  int64_t ld_c;
  const float* out;       /* GRU: forward outputs h_t (h_{t-1} is read); row stride ld_out */
  int64_t ld_out;
  const void* w_packed_t; /* from tad_rnn_pack_weights (backward image) */
  void* dgates;           /* [B*T, 4H] grads of the pre-activations: LSTM i,f,g,o; GRU r, z, n (input side), n (hidden side: r * dn) */
  int64_t ld_dgates;
  int32_t reverse;
  int32_t pad_;
} tad_rnn_bwd_dir_t;

/* BPTT through one LSTM / GRU layer (autograd of the calls above): emits the
 * pre-activation gradients; the weight/bias/input gradients follow from them with
 * tad_linear_bwd_weight(_b16) (gate maps {0,1,2,3} LSTM; GRU input side {0,1,2,-1},
 * hidden side {0,1,-1,2}) and tad_linear_bwd_data(_b16) on the gate-interleaved weights. */
int tad_rnn_bwd(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir,
                const tad_rnn_bwd_dir_t* dirs, tad_stream_t stream);

/* tad_rnn_fwd with device memory for the forms that split the hidden units of a batch tile over several workgroups
 * (bf16 precision and storage at H = 512, where one direction's W_hh fits no CU: eight workgroups of 64 units each keep
 * it in registers and hand their columns of h_t to each other through the workspace every step; DIFF.py:182 -> 220 and
 * VAE.py:119-120 -> 154-155 at hidden_size 512).  tad_rnn_workspace_bytes: what such a form needs for this call, 0 when
 * the shape has none.  workspace: that many bytes, 16-byte aligned, zeroed once by the caller after allocating it,
 * used by one launch at a time (launches that overlap on the device need workspaces of their own); NULL or too small,
 * or pointers / leading dimensions the form cannot take (xproj and gate stash: 16-byte aligned, ld % 8 == 0): the call
 * is tad_rnn_fwd.  Same results as tad_rnn_fwd up to the order of the fp32 sums.  The workgroups of such a launch wait
 * for each other; a wait that runs out of time (0.5 s: the device could not hold the launch's workgroups at once) ends
 * the launch and sets a word in the workspace that tad_rnn_ws_status (synchronises the stream) reports as an error. */
int64_t tad_rnn_workspace_bytes(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir);
int tad_rnn_fwd_ws(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                   void* workspace, int64_t workspace_bytes, tad_stream_t stream);
int tad_rnn_ws_status(const void* workspace, tad_stream_t stream);

/* tad_rnn_fwd_ws for a layer whose input is narrow (at most 32 columns: the first layer of the VAE's two encoder GRUs reads
 * the table's features, VAE.py:119-120 -> 154-155): the cluster form then multiplies x_t by its slice of W_ih itself, and
 * the input-side pre-activations [B*T, 4H] are neither written nor read (dirs[i].xproj is ignored and may be NULL).
 * Per direction: x16 = the layer's input [B*T, kin] bf16 (row stride ldx, a multiple of 8; 16-byte aligned; kin = 8, 16, 24
 * or 32), w_ih16 = the gate-interleaved bf16 image of W_ih that tad_rnn_pack_input writes ([4H, ldw], ldw >= 32, zeros
 * in columns kin..31), bias = tad_rnn_pack_input's bias_gl [4H].  The pre-activations are not rounded to bf16 on the way
 * to the gates as they are in a stored xproj; everything else as tad_rnn_fwd_ws.  Where the cluster form does not apply
 * (see tad_rnn_fwd_ws) the call fails with TAD_ERR_INVALID: the caller forms xproj and calls tad_rnn_fwd_ws. */
typedef struct {
  const void* x16;
  int64_t ldx;
  const void* w_ih16;
  int64_t ldw;
  const float* bias;
  int32_t kin;
  int32_t pad_;
} tad_rnn_xin_t;
int tad_rnn_fwd_ws_x(int cell, int prec, int store, int64_t B, int64_t T, int64_t H, int ndir, const tad_rnn_dir_t* dirs,
                     const tad_rnn_xin_t* xin, void* workspace, int64_t workspace_bytes, tad_stream_t stream);

/* Bytes of ONE packed buffer (forward or backward) of a recurrent weight matrix W_hh[G*H,H].  In bf16 precision
 * the forward buffer holds up to three images: the one the 32-row MFMA forms read, behind it the one of the 16-row form
 * that launches of 8 batch rows per workgroup take, and at H = 512 the one of tad_rnn_fwd_ws's cluster form (every reverse step of `sample`, DIFF.py:340-351, at up to 2047
 * sequences per call; training at the reference's batch of 100, DIFF.py:305-328). */
int64_t tad_rnn_packed_bytes(int prec, int64_t H, int G);
/* Re-lay W_hh[G*H,H] (row stride H) into the MFMA-fragment-ordered images the
 * sequence kernels stream: packed_fwd for h W_hh^T, packed_bwd for dgates W_hh.
 * Either output may be NULL. */
int tad_rnn_pack_weights(int prec, int64_t H, int G, const float* w_hh, void* packed_fwd, void* packed_bwd,
                         tad_stream_t stream);

/* The same for up to 16 matrices of one shape (the layers and directions of a stack) in ONE launch: the images are
 * rebuilt after every optimizer step (DIFF.py:325 / VAE.py:345 `optimizer.step()` changes weight_hh_*), and a launch
 * per image was 2 % of a training step.  packed_fwd / packed_bwd of a job may be NULL. */
typedef struct {
  const float* w_hh;
  void* packed_fwd;
  void* packed_bwd;
} tad_rnn_pack_job_t;
int tad_rnn_pack_weights_multi(int prec, int64_t H, int G, int njobs, const tad_rnn_pack_job_t* jobs, tad_stream_t stream);
/* Gate-interleaved image of the input-side parameters of nblk directions / heads:
 * w_ih [nblk*G*H, in_dim] (torch's gate-major rows, blocks back to back), b_ih and b_hh
 * [nblk*G*H] -> w_gl [nblk*4H, in_dim] with row blk*4H + 4u + s = row blk*G*H + s*H + u
 * of w_ih (zero rows for the unused slot of a GRU), as fp32 or bf16 (w_dtype; row stride
 * ld_gl >= in_dim, columns in_dim..ld_gl-1 zero), and bias_gl [nblk*4H] fp32 = b_ih
 * (+ b_hh for LSTM, whose two biases always appear as a sum) in the same order. */
int tad_rnn_pack_input(int cell, int64_t H, int64_t in_dim, int nblk, const float* w_ih, const float* b_ih,
                       const float* b_hh, void* w_gl, int w_dtype, int64_t ld_gl, float* bias_gl, tad_stream_t stream);

/* ------------------------------------------------------------------ elementwise family
 * add_noise (DIFF.py:102-118; get_gp_covariance DIFF.py:97-100 is the identity):
 * x_noisy[b,:] = sqrt(alphas[i[b]]) x0[b,:] + sqrt(1-alphas[i[b]]) noise[b,:];
 * also writes i_f[b*T+t] = (float) i[b] (the `i` tensor of DIFF.py:254).  row = T*K. */
int tad_add_noise(int64_t B, int64_t T, int64_t K, const float* x0, const float* noise, const int32_t* i_idx,
                  const float* alphas, float* x_noisy, float* i_f, tad_stream_t stream);

/* One reverse-diffusion update (DIFF.py:348-351), in place on x:
 * x = c1*(x - c2*pred) + c3*z;  if last: x = c1*(x - c2*pred) again.
 * c1 = 1/sqrt(1-beta), c2 = beta/sqrt(1-alpha), c3 = sqrt(beta). */
int tad_sampler_update(int64_t n, float* x, const float* pred, const float* z, float c1, float c2, float c3,
                       int last, tad_stream_t stream);

/* Tail of one reverse step fused into one pass over the recurrent output h [rows, D] (row stride ldh):
 * LayerNorm (DIFF.py:183, 221; eps) -> fc (DIFF.py:184, 223; fc_w [K, D], fc_b [K]) -> the update of
 * tad_sampler_update on x [rows, K] with z [rows, K] (DIFF.py:348-351) -> input_proj of the UPDATED x for the
 * next step (DIFF.py:198; w_in [H, K], b_in [H]) into xh [rows, H] (row stride ldxh).  prec = TAD_BF16 rounds
 * the operands of the two projections to bf16 like the matrix products they replace.  D <= 1024, K <= 8,
 * H <= 1024 and 10 H + 3 + 640 n <= 16384 (the launch's LDS), n = 2, 8 or 16: the least of them that is at least
 * D / 64; other shapes take tad_layernorm_fwd + tad_linear_fwd + tad_sampler_update. */
int tad_sampler_tail(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                     const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                     float c1, float c2, float c3, int last, const float* w_in, const float* b_in, float* xh,
                     int64_t ldxh, tad_stream_t stream);

/* tad_sampler_tail with the step's share of `proj` folded in (DIFF.py:203 and :218 for the next reverse step):
 * input_proj has K <= 8 inputs, so proj's product over its output, x_h W_1^T, is a K-column product of the updated x
 * with W_fold = W_1 W_in [H, K] (b_fold = W_1 b_in), which the caller prepares once per call of `sample`.  The row
 * leaves as relu(x_new W_fold^T + b_fold + vrow + pinv[r]) in bf16: the operand of the LSTM's layer-0 input product
 * (vrow = the step's row of the i_enc table, pinv = the step-invariant share of proj incl. its bias, row stride
 * ld_pinv).  One [rows, H] x [H, H] product and one launch fewer per reverse step; rounding differs from the two
 * products only in that x_h is not rounded to bf16 in between.  prec = TAD_BF16 rounds LayerNorm's output, fc's
 * weight and the updated x as tad_sampler_tail does; W_fold and b_fold are used as given. */
int tad_sampler_tail_proj_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                              const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                              float c1, float c2, float c3, int last, const float* w_fold, const float* b_fold,
                              const float* vrow, const float* pinv, int64_t ld_pinv, void* pa16, int64_t ldpa,
                              tad_stream_t stream);
/* tad_sampler_tail_proj_b16 drawing the step's noise itself instead of reading z: lane k of a row's wave evaluates the
 * value tad_sampler_noise(seed, counter, ...) gives element k of global row row0 + r (bit for bit), while the row's loads
 * are in flight -- the noise tensor of a reverse step (DIFF.py:346 `torch.randn_like(x)`) is neither written nor read and
 * its launch disappears. */
int tad_sampler_tail_proj_cn_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                                 const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, uint64_t seed,
                                 uint64_t counter, int64_t row0, float c1, float c2, float c3, int last, const float* w_fold,
                                 const float* b_fold, const float* vrow, const float* pinv, int64_t ld_pinv, void* pa16,
                                 int64_t ldpa, tad_stream_t stream);
/* The same launch with the next step's input projection written as bf16 (it feeds `proj`'s product only). */
int tad_sampler_tail_b16(int prec, int64_t rows, int D, int K, int H, const float* h, int64_t ldh, const float* gamma,
                     const float* beta, float eps, const float* fc_w, const float* fc_b, float* x, const float* z,
                     float c1, float c2, float c3, int last, const float* w_in, const float* b_in, void* xh16,
                     int64_t ldxh, tad_stream_t stream);

/* Denoiser loss (DIFF.py:259-260) and its gradient: loss_sum[0] += sum (pred-target)^2,
 * dpred = grad_scale * (pred - target)  (grad_scale = 2/numel of the GLOBAL batch). */
int tad_mse_loss(int64_t n, const float* pred, const float* target, float grad_scale, float* dpred,
                 float* loss_sum, tad_stream_t stream);

/* nn.L1Loss of the post-hoc predictor (Evaluation/predictive_metrics.py:27, 38-40) and its gradient:
 * loss_sum[0] += sum |pred - target|, dpred = grad_scale * sign(pred - target) (grad_scale = 1 / numel). */
int tad_l1_loss(int64_t n, const float* pred, const float* target, float grad_scale, float* dpred, float* loss_sum,
                tad_stream_t stream);

/* Squared L2 norm of a flat buffer: out_sum[0] += sum g[i]^2, one launch (wave reductions, one atomic per
 * workgroup).  For the gradient norm of a step over the flat gradient buffer -- the check a data-parallel run uses to
 * see that every rank holds the same reduced gradient, and what torch.nn.utils.clip_grad_norm_ would compute; the
 * reference's loops (DIFF.py:318-325, VAE.py:341-345) do not clip. */
int tad_grad_norm_sq(int64_t n, const float* g, float* out_sum, tad_stream_t stream);

/* compute_sine_cosine (DIFF.py:14-32, VAE.py:24-42) with num_terms = 8:
 * out[r, j*8+k] = sin(2^k pi x[r,j]), out[r, 8n + j*8+k] = cos(...). */
int tad_sincos_features(int64_t rows, int n_nums, const float* x, int64_t ldx, float* out, int64_t ldo,
                        tad_stream_t stream);
/* The same values rounded to bf16 (the features feed a matrix product only). */
int tad_sincos_features_b16(int64_t rows, int n_nums, const float* x, int64_t ldx, void* out16, int64_t ldo,
                            tad_stream_t stream);

/* Embedding lookups of Embedding_data.forward (DIFF.py:66-75, VAE.py:89-98):
 * out[r, d*E + e] = tables[(tab_off[d] + (int)x[r,d]) * E + e], d < n_disc.
 * Codes are clamped to [0, cards[d]-1] (the reference raises on out-of-range codes).
 * tab_off / cards are HOST arrays of n_disc (<= 64) entries, copied into the launch. */
int tad_embed_fwd(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* tables,
                  const int32_t* tab_off, const int32_t* cards, float* out, int64_t ldo, tad_stream_t stream);
/* The same rows rounded to bf16 (they feed a matrix product only). */
int tad_embed_fwd_b16(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* tables,
                      const int32_t* tab_off, const int32_t* cards, void* out16, int64_t ldo, tad_stream_t stream);
/* ... and its autograd: dtables[(tab_off[d]+code)*E + e] += dout[r, d*E+e]. */
int tad_embed_bwd(int64_t rows, int n_disc, int E, const float* x, int64_t ldx, const float* dout, int64_t ldo,
                  float* dtables, const int32_t* tab_off, const int32_t* cards, tad_stream_t stream);

/* PositionalEncoding.forward (DIFF.py:143-146): out[r, :L] = w[:]*(t[r]*inv_max) + b[:],
 * out[r, L+j] = sin(t[r]*scale[j] + shift[j]); L = H/2, j < H-L.  t[r] = t[r*t_stride]. */
int tad_posenc_fwd(int64_t rows, int H, const float* t, int64_t t_stride, float inv_max, const float* w,
                   const float* b, const float* scale, const float* shift, float* out, int64_t ldo,
                   tad_stream_t stream);
/* The same values rounded to bf16 (in the training step the encodings feed `proj` only, DIFF.py:207-208). */
int tad_posenc_fwd_b16(int64_t rows, int H, const float* t, int64_t t_stride, float inv_max, const float* w,
                       const float* b, const float* scale, const float* shift, void* out16, int64_t ldo,
                       tad_stream_t stream);

/* nn.SiLU (DIFF.py:58, VAE.py:81) forward / backward. */
int tad_silu_fwd(int64_t n, const float* z, float* a, tad_stream_t stream);
int tad_silu_bwd(int64_t n, const float* z, const float* da, float* dz, tad_stream_t stream);
/* Results rounded to bf16: a = silu(z) and dz feed matrix products only (mlp_nums.2, the gradient of mlp_nums.0). */
int tad_silu_fwd_b16(int64_t n, const float* z, void* a16, tad_stream_t stream);
int tad_silu_bwd_b16(int64_t n, const float* z, const float* da, void* dz16, tad_stream_t stream);

/* `layer_norm` + `fc` of BiRNN_score.forward (DIFF.py:221-223) in one pass over the recurrent output: LayerNorm of
 * a row (eps 1e-5; mean_rstd[r] = {mean, rstd} for the backward), then pred[r, :] = fc_w [K, D] applied to it + fc_b
 * (K <= 8, D <= 1024).  The normalised row is written only if y != NULL.  prec = TAD_BF16 rounds the projection's
 * operands to bf16 like the matrix kernels do. */
int tad_ln_fc_fwd(int prec, int64_t rows, int D, int K, const float* x, int64_t ldx, const float* gamma, const float* beta,
                  float eps, const float* fc_w, const float* fc_b, float* y, int64_t ldy, float* mean_rstd, float* pred,
                  tad_stream_t stream);
/* ... and its autograd in one pass: the normalised row is recomputed from x and mean_rstd; dx [rows, D] =
 * LayerNorm-backward(dpred fc_w); dgamma, dbeta, dfc_w [K, D], dfc_b [K] are accumulated into. */
int tad_ln_fc_bwd(int prec, int64_t rows, int D, int K, const float* dpred, const float* x, int64_t ldx, const float* gamma,
                  const float* beta, const float* mean_rstd, const float* fc_w, float* dx, int64_t lddx, float* dgamma,
                  float* dbeta, float* dfc_w, float* dfc_b, tad_stream_t stream);

/* nn.LayerNorm over the last dim D (DIFF.py:183, 221), eps 1e-5; mean_rstd[r] = {mean, rstd}. */
int tad_layernorm_fwd(int64_t rows, int D, const float* x, int64_t ldx, const float* gamma, const float* beta,
                      float eps, float* y, int64_t ldy, float* mean_rstd, tad_stream_t stream);
int tad_layernorm_bwd(int64_t rows, int D, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                      const float* gamma, const float* mean_rstd, float* dx, int64_t lddx, float* dgamma,
                      float* dbeta, tad_stream_t stream);

/* Row gather of the minibatch (DIFF.py:320, VAE.py:340-348): out[r,:] = src[idx[r],:]. */
int tad_gather_rows(int64_t rows, int64_t row_len, const float* src, const int64_t* idx, float* out,
                    tad_stream_t stream);

/* The random draws of one training iteration on the device (DIFF.py:306 `random.sample(all_indices, batch_size)`,
 * DIFF.py:317 `torch.rand(B, T, 1).sort(1)[0]`, DIFF.py:253 `torch.randint(0, diffusion_steps, size=(B,))`,
 * DIFF.py:108 `torch.randn_like(x)`), for the trainers' `draws="device"` mode: idx[B] = B distinct window indices in
 * [0, N) (a keyed bijection of [0, N) applied to 0..B-1), t[B, T] = T uniforms in [0, 1) per sample in ascending
 * order, i_idx[B] uniform in [0, n_steps), noise[B, T, K] standard normals.  Counter-based (Philox4x32-10 keyed by
 * `seed`, iteration `counter` < 2^32): the same (seed, counter) gives the same batch on every rank and in every
 * replay; nothing is kept between calls.  Any of the four outputs may be NULL.  The values are NOT the reference's
 * (its CPU generators cannot be replayed on the device); `draws="reference"` keeps those. */
int tad_draw_batch(uint64_t seed, uint64_t counter, int64_t N, int64_t B, int64_t T, int64_t K, int n_steps,
                   int64_t* idx, float* t, int32_t* i_idx, float* noise, tad_stream_t stream);

/* The noise of the reverse loop for `noise_source="device"` (DIFF.py:335 `x = torch.randn(B, T, F)`, DIFF.py:347
 * `z = torch.randn(B, T, F)` once per step): out[r, k], r < rows, k < K = standard normals of rows row0 + r of the
 * GLOBAL [B*T, K] tensor at step `counter` (< 2^32; the caller uses the step index, and the number of steps for x_T) --
 * counter-based like tad_draw_batch, so a rank or a concurrent chunk draws exactly its rows, one step at a time,
 * instead of the whole [steps, B, T, F] block up front.  Not the reference's values (`noise_source="reference"` replays
 * its CPU generator). */
int tad_sampler_noise(uint64_t seed, uint64_t counter, int64_t row0, int64_t rows, int K, float* out, tad_stream_t stream);

/* `optimizer.zero_grad()` over the flat gradient buffer (DIFF.py:321, VAE.py:343) and the other clears of a step
 * (an unused column block of the concatenated encodings, DIFF.py:200-203 with the step encoding left out): zero
 * `rows` rows of `row_bytes` bytes, `ld_bytes` apart (4-byte granularity). */
int tad_zero_2d(int64_t rows, int64_t row_bytes, void* p, int64_t ld_bytes, tad_stream_t stream);
/* dst[r, :cols] = src[r, :cols] (float32): the copy of a layer input into its zero-padded buffer (VAE.py:147-151:
 * the encoders read x with F columns; the bf16 kernels want a multiple of 8). */
int tad_copy_2d(int64_t rows, int64_t cols, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                tad_stream_t stream);
/* out = a + b (float32 vectors): the bias of `input_proj(x) + cond_output(h)` computed as one product
 * (DIFF.py:197-198). */
int tad_add_vec(int64_t n, const float* a, const float* b, float* out, tad_stream_t stream);
/* p[0..n) = value: `torch.ones_like` masks of auto_loss's callers (VAE.py:222-253 evaluates the loss on outputs
 * that already carry the mask) and of the post-hoc discriminator's labels (Evaluation/Metrics.py:101-106). */
int tad_fill_f32(int64_t n, float* p, float value, tad_stream_t stream);
/* p = 1 / (1 + exp(-p)) in place: `torch.sigmoid(logit)` of the post-hoc discriminator (Evaluation/Metrics.py:62-65). */
int tad_sigmoid_f32(int64_t n, float* p, tad_stream_t stream);

/* Minibatch of sliding windows straight from the [rows, F] table they are cut from
 * (dataprovider_pypots.py:75-87 `sliding_window` followed by the trainers' index gather, DIFF.py:320,
 * VAE.py:340-348): out[b, t, :] = table[idx[b] * stride + t, :], b < B, t < T.  The caller guarantees
 * idx[b] * stride + T <= rows of the table.  Window i of the reference's stacked tensor is idx = i. */
int tad_gather_windows(int64_t B, int64_t T, int64_t F, int64_t stride, const float* table, int64_t ld,
                       const int64_t* idx, float* out, tad_stream_t stream);

/* torch.optim.Adam step on a flat buffer (DIFF.py:291, 325; VAE.py:295, 354), coupled
 * L2 weight decay, bias correction for step number `step` (1-based). */
int tad_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, tad_stream_t stream);
/* The same step with its two step-dependent factors read from device memory: coef2 = {lr / (1 - beta1^step),
 * 1 / sqrt(1 - beta2^step)} as tad_adam_coefficients (host function, the arithmetic of tad_adam_step) computes them.
 * For launches captured into a HIP graph: the host refreshes coef2 before every replay. */
void tad_adam_coefficients(float lr, float beta1, float beta2, int step, float* out2);
int tad_adam_step_dev(int64_t n, float* p, const float* g, float* m, float* v, float beta1, float beta2, float eps,
                      float weight_decay, const float* coef2, tad_stream_t stream);
/* The step counter of `optimizer.step()` (torch.optim.Adam's state['step'], DIFF.py:325, VAE.py:354) kept on the
 * device for captured steps: coef4 = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step), step (int32 bits), unused};
 * one single-thread launch increments the step and refreshes the two factors tad_adam_step_dev reads, so that a
 * replayed graph needs nothing from the host. */
int tad_adam_tick(float lr, float beta1, float beta2, float* coef4, tad_stream_t stream);

/* ------------------------------------------------------------------ VAE family
 * reparametrize (VAE.py:147-150) + KLD partial sum (VAE.py:350-351):
 * z = mu + eps*exp(lv/2); kld_sum[0] += sum(1 + lv - mu^2 - exp(lv)). */
int tad_vae_reparam_fwd(int64_t n, const float* mu, const float* lv, const float* eps, float* z, float* kld_sum,
                        tad_stream_t stream);
/* dmu = dz + kld_scale*mu; dlv = dz*eps*0.5*exp(lv/2) - 0.5*kld_scale*(1-exp(lv));
 * kld_scale = beta / numel (global). */
int tad_vae_reparam_bwd(int64_t n, const float* dz, const float* mu, const float* lv, const float* eps,
                        float kld_scale, float* dmu, float* dlv, tad_stream_t stream);

/* Decoder heads epilogue + auto_loss (VAE.py:162-191 mask injection, 206-213 heads,
 * 222-253 loss).  `heads` [rows, NH] holds the raw head pre-activations in the column
 * order bins | cat_0 | ... | cat_{nc-1} | nums (NH = nb + sum cards + nn) and is
 * overwritten IN PLACE with the reference-visible outputs (bins + mask offset,
 * cat logits + mask offset on class 0, sigmoid(nums)*mask).  If dheads != NULL it
 * receives d(loss)/d(raw pre-activations) with the mean reductions' denominators
 * given by scale_bce, scale_ce (already divided by n_cats) and scale_mse.
 * loss_parts[0] += sum BCE, [1] += sum MSE numerators, [2+j] += sum CE of cat j.
 * mask/target are [rows, F] with F = nb + nc + nn.  cards is a HOST array of nc (<= 64)
 * cardinalities. */
int tad_vae_heads_loss(int64_t rows, int nb, int nc, int nn, const int32_t* cards, float* heads, int64_t ldh,
                       const float* mask, const float* target, int64_t ldf, float scale_bce, float scale_ce,
                       float scale_mse, float* dheads, float* loss_parts, tad_stream_t stream);

/* convert_to_tensor (process_edited.py:252-301), the step every driver runs on the decoder's
 * outputs after sampling: `heads` [rows, NH] holds the reference-visible head outputs in
 * the column order bins | cat_0 | ... | nums (as tad_vae_heads_loss leaves them);
 * out[r, :] = [ sigmoid(bin) > 0.5 ? 1 : 0 | first argmax of softmax(cat_j) | nums ],
 * rows of F = nb + nc + nn floats, row stride ldo. */
int tad_heads_to_tensor(int64_t rows, int nb, int nc, int nn, const int32_t* cards, const float* heads, int64_t ldh,
                        float* out, int64_t ldo, tad_stream_t stream);

/* inp = a + b and target = (a + b) * mask (VAE.py:331-336), elementwise over n. */
int tad_vae_prepare(int64_t n, const float* resp, const float* cond, const float* mask, float* inp, float* target,
                    tad_stream_t stream);

/* ------------------------------------------------------------------ TV-MCG metadata corruption
 * add_continuous_metadata_noise_subset (dataprovider_pypots.py:149-180) on cond [rows = B*T, F]:
 * out = cond, except out[r, cols[j]] = cond[r, cols[j]] + noise[r*n_cols + j] * (level * (std_j + 1e-8)),
 * std_j = unbiased standard deviation of column cols[j] over all rows (float64 two-pass reduction rounded
 * to float32).  cols is a HOST array of n_cols (<= 64) distinct column indices; noise [rows, n_cols] holds
 * the standard-normal draws (the reference's randn_like(x_cont)); stats is device scratch of 2*n_cols
 * doubles owned by the caller (sums, then sums of squared deviations; cleared by the call).  The reference's
 * early return for level <= 0 or no columns is the caller's. */
int tad_metadata_noise(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const float* cond, int64_t ld,
                       const float* noise, float level, double* stats, float* out, int64_t ldo, tad_stream_t stream);

/* tad_metadata_noise without the caller-provided workspace: the column statistics live in a 1 KiB array inside
 * the library's device code (one copy per GPU, held until the process ends), so the call neither takes nor makes
 * an allocation.  No counterpart in the reference: it is for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  Same arguments and results as
 * tad_metadata_noise; n_cols beyond 64 is rejected.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams) race; calls on one stream are ordered and each clears what it uses.  The
 * array used is that of the CURRENT device: a non-null stream of another device is rejected (TAD_ERR_INVALID). */
int tad_metadata_noise_static(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const float* cond, int64_t ld,
                              const float* noise, float level, float* out, int64_t ldo, tad_stream_t stream);

/* flip_categorical_metadata_subset (dataprovider_pypots.py:182-229): out = cond, except for the n_cols
 * (<= 64, HOST arrays cols / classes, classes[j] >= 2) selected columns, where
 * code = (int64) cond[r, col] (truncation, as .long()), and
 * out[r, col] = rnd[j*rows + r] < p ? base + (base >= code) : code, base = base_draw[j*rows + r]
 * (the reference's rand_like and randint(0, classes-1) draws of that column, [n_cols, rows] each).
 * Columns with fewer than 2 classes are skipped by the caller, as the reference skips them. */
int tad_flip_categories(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const int32_t* classes,
                        const float* cond, int64_t ld, const float* rnd, const int64_t* base_draw, float p, float* out,
                        int64_t ldo, tad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAD_HIP_H */
