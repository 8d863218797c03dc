// VAE-specific fused kernels (gfx950): reparametrisation + KLD, and the decoder-head
// epilogue (mask injection, sigmoid) fused with the mixed BCE / CE / MSE loss and its
// gradient.  HBM-bound, one pass over the head pre-activations.
#include <math.h>

#include "common.h"

namespace {

constexpr int VB = 256;
constexpr int MAX_CATS = 64;

__device__ __forceinline__ void block_sum_to(float v, float* out) {
  __shared__ float part[VB / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < VB / 64; ++w) s += part[w];
    if (s != 0.f) atomicAdd(out, s);
  }
}

__global__ void reparam_fwd_kernel(int64_t n, const float* __restrict__ mu, const float* __restrict__ lv,
                                   const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ kld_sum) {
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[e], l = lv[e];
    z[e] = m + eps[e] * expf(0.5f * l);
    acc += 1.f + l - m * m - expf(l);
  }
  if (kld_sum) block_sum_to(acc, kld_sum);
}

__global__ void reparam_bwd_kernel(int64_t n, const float* __restrict__ dz, const float* __restrict__ mu,
                                   const float* __restrict__ lv, const float* __restrict__ eps, float ks,
                                   float* __restrict__ dmu, float* __restrict__ dlv) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float g = dz[e], l = lv[e];
    dmu[e] = g + ks * mu[e];
    dlv[e] = g * eps[e] * 0.5f * expf(0.5f * l) - 0.5f * ks * (1.f - expf(l));
  }
}

struct HeadMeta {
  int32_t card[MAX_CATS];
};

// one thread per row
__global__ void heads_loss_kernel(int64_t rows, int nb, int nc, int nn, HeadMeta hm, float* __restrict__ heads, int64_t ldh,
                                  const float* __restrict__ mask, const float* __restrict__ target, int64_t ldf,
                                  float s_bce, float s_ce, float s_mse, float* __restrict__ dheads,
                                  float* __restrict__ parts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < rows;
  float bce = 0.f, mse = 0.f;
  float* h = heads + (live ? r : 0) * ldh;
  float* dh = dheads ? dheads + (live ? r : 0) * ldh : nullptr;
  const float* mk = mask + (live ? r : 0) * ldf;
  const float* tg = target + (live ? r : 0) * ldf;
  int col = 0;
  if (live) {
    for (int j = 0; j < nb; ++j, ++col) {
      const float x = h[col] + (mk[j] == 0.f ? -100.f : 0.f);
      const float y = tg[j];
      h[col] = x;
      bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      if (dh) dh[col] = (sigmoid_acc(x) - y) * s_bce;
    }
  } else {
    col += nb;
  }
  for (int j = 0; j < nc; ++j) {
    const int card = hm.card[j];
    float ce = 0.f;
    if (live) {
      h[col] += (mk[nb + j] == 0.f ? 100.f : 0.f);
      float mx = h[col];
      for (int q = 1; q < card; ++q) mx = fmaxf(mx, h[col + q]);
      float se = 0.f;
      for (int q = 0; q < card; ++q) se += expf(h[col + q] - mx);
      const float lse = mx + logf(se);
      int t = (int)tg[nb + j];
      t = min(max(t, 0), card - 1);
      ce = lse - h[col + t];
      if (dh) {
        for (int q = 0; q < card; ++q) dh[col + q] = (expf(h[col + q] - lse) - (q == t ? 1.f : 0.f)) * s_ce;
      }
    }
    col += card;
    if (parts) block_sum_to(ce, parts + 2 + j);
  }
  if (live) {
    for (int j = 0; j < nn; ++j, ++col) {
      const float s = sigmoid_acc(h[col]);
      const float m = mk[nb + nc + j];
      const float o = s * m;
      h[col] = o;
      const float d = o - tg[nb + nc + j];
      mse += d * d;
      if (dh) dh[col] = s_mse * d * m * s * (1.f - s);
    }
  }
  if (parts) {
    block_sum_to(bce, parts + 0);
    block_sum_to(mse, parts + 1);
  }
}

// The same, one thread per (row, head column), rpb = VB / nh whole rows per workgroup: a row's columns are read by
// neighbouring lanes (the one-thread-per-row form walks 35 columns one by one, every lane in another cache line, with six
// waves per CU to hide it: 135 us for 98 k rows against 15 us of memory time; this form: 78 us).  Every element is computed with the
// expressions of the kernel above; a categorical column evaluates its feature's softmax terms itself (a handful of values
// that its neighbours load too).  heads is updated in place: all loads of a workgroup come before a barrier, all stores
// after it, and a row belongs to one workgroup.
__global__ void heads_loss_cols_kernel(int64_t rows, int nb, int nc, int nn, int nh, int rpb, HeadMeta hm, float* __restrict__ heads,
                                       int64_t ldh, const float* __restrict__ mask, const float* __restrict__ target, int64_t ldf,
                                       float s_bce, float s_ce, float s_mse, float* __restrict__ dheads, float* __restrict__ parts) {
  const int rl = threadIdx.x / nh, col = threadIdx.x - rl * nh;
  float bce = 0.f, mse = 0.f, ce = 0.f;          // summed over the workgroup's row groups: one reduction and atomic per sum at the end
  int my_cat = -1;
  const int64_t ngroups = (rows + rpb - 1) / rpb;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const int64_t r = grp * rpb + rl;
  const bool live = rl < rpb && r < rows;
  float hv = 0.f, dv = 0.f;
  bool write_h = false;
  if (live) {
    const float* h = heads + r * ldh;
    const float* mk = mask + r * ldf;
    const float* tg = target + r * ldf;
    if (col < nb) {
      const float x = h[col] + (mk[col] == 0.f ? -100.f : 0.f);
      const float y = tg[col];
      hv = x; write_h = true;
      bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      dv = (sigmoid_acc(x) - y) * s_bce;
    } else if (col >= nh - nn) {
      const int j = col - (nh - nn);
      const float sg = sigmoid_acc(h[col]);
      const float m = mk[nb + nc + j];
      const float o = sg * m;
      hv = o; write_h = true;
      const float d = o - tg[nb + nc + j];
      mse += d * d;
      dv = s_mse * d * m * sg * (1.f - sg);
    } else {
      int c0 = nb, j = 0;
      while (col >= c0 + hm.card[j]) { c0 += hm.card[j]; ++j; }
      const int card = hm.card[j], q = col - c0;
      const float h0 = h[c0] + (mk[nb + j] == 0.f ? 100.f : 0.f);       // the feature's first logit carries the mask
      float mx = h0;
      for (int i = 1; i < card; ++i) mx = fmaxf(mx, h[c0 + i]);
      float se = expf(h0 - mx);
      for (int i = 1; i < card; ++i) se += expf(h[c0 + i] - mx);
      const float lse = mx + logf(se);
      int t = (int)tg[nb + j];
      t = min(max(t, 0), card - 1);
      const float mine = q == 0 ? h0 : h[col];
      if (q == 0) {
        hv = h0; write_h = true;
        ce += lse - (t == 0 ? h0 : h[c0 + t]);
        my_cat = j;
      }
      dv = (expf(mine - lse) - (q == t ? 1.f : 0.f)) * s_ce;
    }
  }
  __syncthreads();
  if (live) {
    if (write_h) heads[r * ldh + col] = hv;
    if (dheads) dheads[r * ldh + col] = dv;
  }
  }
  if (parts) {
    block_sum_to(bce, parts + 0);
    block_sum_to(mse, parts + 1);
    for (int j = 0; j < nc; ++j) block_sum_to(my_cat == j ? ce : 0.f, parts + 2 + j);
  }
}

__global__ void prepare_kernel(int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ mask, float* __restrict__ inp, float* __restrict__ target) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = a[e] + (b ? b[e] : 0.f);
    if (inp) inp[e] = v;
    if (target) target[e] = v * mask[e];
  }
}

// Decoder outputs -> data table row (convert_to_tensor): one thread per row.
//   bins: 1 where sigmoid(logit) > 0.5, evaluated in fp32 as 1 / (1 + exp(-x)) like torch.sigmoid;
//   cats: index of the first maximum of the fp32 softmax exp(x - max) / sum (torch.argmax of F.softmax);
//   nums: copied.
__global__ void heads_to_tensor_kernel(int64_t rows, int nb, int nc, int nn, HeadMeta hm, const float* __restrict__ heads,
                                       int64_t ldh, float* __restrict__ out, int64_t ldo) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* h = heads + r * ldh;
  float* o = out + r * ldo;
  int col = 0;
  for (int j = 0; j < nb; ++j, ++col) o[j] = sigmoid_acc(h[col]) > 0.5f ? 1.f : 0.f;
  for (int j = 0; j < nc; ++j) {
    const int card = hm.card[j];
    float mx = h[col];
    for (int q = 1; q < card; ++q) mx = fmaxf(mx, h[col + q]);
    float se = 0.f;
    for (int q = 0; q < card; ++q) se += expf(h[col + q] - mx);
    int best = 0;
    float pb = expf(h[col] - mx) / se;
    for (int q = 1; q < card; ++q) {
      const float pq = expf(h[col + q] - mx) / se;
      if (pq > pb) { pb = pq; best = q; }
    }
    o[nb + j] = (float)best;
    col += card;
  }
  for (int j = 0; j < nn; ++j, ++col) o[nb + nc + j] = h[col];
}

inline unsigned vgrid(int64_t n) {
  int64_t g = cdiv64(n, VB);
  if (g > 2048) g = 2048;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int tad_vae_reparam_fwd(int64_t n, const float* mu, const float* lv, const float* eps, float* z, float* kld_sum,
                                   tad_stream_t stream) {
  TAD_CHECK(n >= 0 && mu && lv && eps && z, "tad_vae_reparam_fwd: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(vgrid(n)), dim3(VB), 0, (hipStream_t)stream, n, mu, lv, eps, z, kld_sum);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_vae_reparam_bwd(int64_t n, const float* dz, const float* mu, const float* lv, const float* eps,
                                   float kld_scale, float* dmu, float* dlv, tad_stream_t stream) {
  TAD_CHECK(n >= 0 && dz && mu && lv && eps && dmu && dlv, "tad_vae_reparam_bwd: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(vgrid(n)), dim3(VB), 0, (hipStream_t)stream, n, dz, mu, lv, eps, kld_scale, dmu, dlv);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_vae_heads_loss(int64_t rows, int nb, int nc, int nn, const int32_t* cards, float* heads, int64_t ldh,
                                  const float* mask, const float* target, int64_t ldf, float scale_bce, float scale_ce,
                                  float scale_mse, float* dheads, float* loss_parts, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && nb >= 0 && nc >= 0 && nn >= 0 && nc <= MAX_CATS && heads && mask && target, "tad_vae_heads_loss: bad arguments");
  TAD_CHECK(nc == 0 || cards, "tad_vae_heads_loss: cards missing");
  HeadMeta hm{};
  int64_t nh = nb + nn;
  for (int j = 0; j < nc; ++j) {
    TAD_CHECK(cards[j] > 0, "tad_vae_heads_loss: bad cardinality");
    hm.card[j] = cards[j];
    nh += cards[j];
  }
  TAD_CHECK(ldh >= nh && ldf >= nb + nc + nn, "tad_vae_heads_loss: bad leading dimension");
  if (rows == 0) return TAD_OK;
  if (nh >= 1 && nh <= VB && nc <= 8) {
    // (up to eight categorical features: their loss sums are one workgroup reduction each)
    const int rpb = (int)(VB / nh);
    int64_t grid = cdiv64(rows, rpb);
    if (grid > 1024) grid = 1024;      // (98 k rows: 160 / 94 / 78 / 85 / 125 us with 256 / 512 / 1024 / 2048 / 4096 workgroups)
    hipLaunchKernelGGL(heads_loss_cols_kernel, dim3((unsigned)grid), dim3(VB), 0, (hipStream_t)stream, rows, nb, nc, nn,
                       (int)nh, rpb, hm, heads, ldh, mask, target, ldf, scale_bce, scale_ce, scale_mse, dheads, loss_parts);
  } else {
    hipLaunchKernelGGL(heads_loss_kernel, dim3((unsigned)cdiv64(rows, VB)), dim3(VB), 0, (hipStream_t)stream, rows, nb, nc, nn, hm,
                       heads, ldh, mask, target, ldf, scale_bce, scale_ce, scale_mse, dheads, loss_parts);
  }
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_vae_prepare(int64_t n, const float* resp, const float* cond, const float* mask, float* inp, float* target,
                               tad_stream_t stream) {
  TAD_CHECK(n >= 0 && resp && (!target || mask), "tad_vae_prepare: bad arguments");
  if (n == 0) return TAD_OK;
  hipLaunchKernelGGL(prepare_kernel, dim3(vgrid(n)), dim3(VB), 0, (hipStream_t)stream, n, resp, cond, mask, inp, target);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

extern "C" int tad_heads_to_tensor(int64_t rows, int nb, int nc, int nn, const int32_t* cards, const float* heads, int64_t ldh,
                                   float* out, int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && nb >= 0 && nc >= 0 && nn >= 0 && nc <= MAX_CATS && heads && out, "tad_heads_to_tensor: bad arguments");
  TAD_CHECK(nc == 0 || cards, "tad_heads_to_tensor: cards missing");
  HeadMeta hm{};
  int64_t nh = nb + nn;
  for (int j = 0; j < nc; ++j) {
    TAD_CHECK(cards[j] > 0, "tad_heads_to_tensor: bad cardinality");
    hm.card[j] = cards[j];
    nh += cards[j];
  }
  TAD_CHECK(ldh >= nh && ldo >= nb + nc + nn, "tad_heads_to_tensor: bad leading dimension");
  if (rows == 0) return TAD_OK;
  hipLaunchKernelGGL(heads_to_tensor_kernel, dim3((unsigned)cdiv64(rows, VB)), dim3(VB), 0, (hipStream_t)stream, rows, nb, nc, nn, hm,
                     heads, ldh, out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
