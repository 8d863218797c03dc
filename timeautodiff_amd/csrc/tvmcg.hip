// TV-MCG metadata corruption on the device (gfx950): Gaussian noise scaled by the per-column standard
// deviation on selected continuous columns, and random class flips on selected categorical columns of the
// conditioning tensor cond [rows = B*T, F] (reference dataprovider_pypots.py:149-229).  HBM-bound: the
// statistics take two passes over the selected columns (float64 accumulation: sum, then squared deviations
// from the mean, as a two-pass float64 standard deviation rounds to the float32 torch reports), the update
// one pass over all of cond.
#include <math.h>

#include <atomic>

#include "common.h"

namespace {

constexpr int TV_BLOCK = 256;
constexpr int TV_MAX_COLS = 64;

struct ColSet {
  int32_t col[TV_MAX_COLS];
  int32_t classes[TV_MAX_COLS];
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// PASS 0: acc[j] += sum_r x[r, col_j];  PASS 1: acc[n + j] += sum_r (x[r, col_j] - acc[j] / rows)^2
template <int PASS>
__global__ void colstats_kernel(int64_t rows, int n, ColSet cs, const float* __restrict__ x, int64_t ld, double* acc) {
  const int lane = threadIdx.x & 63;
  for (int j = 0; j < n; ++j) {
    const int c = cs.col[j];
    const double mean = PASS ? acc[j] / (double)rows : 0.0;
    double s = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
      const double v = (double)x[r * ld + c] - mean;
      s += PASS ? v * v : v;
    }
    s = wave_sum_f64(s);
    if (lane == 0) atomicAdd(&acc[PASS * n + j], s);
  }
}

__global__ void metadata_noise_kernel(int64_t rows, int F, int n, ColSet cs, const float* __restrict__ cond, int64_t ld,
                                      const float* __restrict__ noise, float level, const double* __restrict__ acc,
                                      float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / F;
    const int c = (int)(e - r * F);
    float v = cond[r * ld + c];
    for (int j = 0; j < n; ++j) {
      if (cs.col[j] == c) {
        // unbiased standard deviation (rows == 1 gives 0/0 = NaN like torch.std), rounded to float32, then the
        // reference's float32 sequence: (std + 1e-8), level * that, noise * that, x + that
        const float sd = (float)sqrt(acc[n + j] / (double)(rows - 1)) + 1e-8f;
        const float scale = level * sd;
        v = v + noise[r * n + j] * scale;
        break;
      }
    }
    out[r * ldo + c] = v;
  }
}

__global__ void flip_categories_kernel(int64_t rows, int F, int n, ColSet cs, const float* __restrict__ cond, int64_t ld,
                                       const float* __restrict__ rnd, const int64_t* __restrict__ base, float p,
                                       float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / F;
    const int c = (int)(e - r * F);
    float v = cond[r * ld + c];
    for (int j = 0; j < n; ++j) {
      if (cs.col[j] == c) {
        const int64_t code = (int64_t)v;                       // .long(): truncation toward zero, flipped or not
        int64_t nv = code;
        if (rnd[(int64_t)j * rows + r] < p) {
          const int64_t b = base[(int64_t)j * rows + r];       // drawn from 0 .. classes-2: skip over the current class
          nv = b + (b >= code ? 1 : 0);
        }
        v = (float)nv;
        break;
      }
    }
    out[r * ldo + c] = v;
  }
}

inline unsigned tv_grid(int64_t n) {
  int64_t g = cdiv64(n, (int64_t)TV_BLOCK);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

int fill_cols(ColSet& cs, int n, int F, const int32_t* cols, const int32_t* classes, const char* who) {
  TAD_CHECK(n >= 0 && n <= TV_MAX_COLS && (n == 0 || cols), "%s: 0..%d selected columns supported (got %d)", who, TV_MAX_COLS, n);
  for (int j = 0; j < n; ++j) {
    TAD_CHECK(cols[j] >= 0 && cols[j] < F, "%s: column %d outside 0..%d", who, (int)cols[j], F - 1);
    for (int k = 0; k < j; ++k) TAD_CHECK(cols[k] != cols[j], "%s: column %d listed twice", who, (int)cols[j]);
    cs.col[j] = cols[j];
    cs.classes[j] = classes ? classes[j] : 0;
  }
  return TAD_OK;
}

}  // namespace

#define TV_STREAM ((hipStream_t)stream)

namespace {

// The three launches of the metadata-noise entry points; `stats` is device memory of 2*n_cols doubles.
int metadata_noise_launch(const char* who, int64_t rows, int64_t F, int n_cols, const int32_t* cols, const float* cond, int64_t ld,
                          const float* noise, float level, double* stats, float* out, int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && F > 0 && F < (1 << 30) && cond && out && ld >= F && ldo >= F, "%s: bad arguments rows=%ld F=%ld ld=%ld ldo=%ld",
            who, (long)rows, (long)F, (long)ld, (long)ldo);
  TAD_CHECK(n_cols == 0 || (noise && stats), "%s: noise and stats are required when columns are selected", who);
  ColSet cs;
  int rc = fill_cols(cs, n_cols, (int)F, cols, nullptr, who);
  if (rc != TAD_OK) return rc;
  if (rows == 0) return TAD_OK;
  if (n_cols > 0) {
    hipError_t e = tad_zero_async(stats, sizeof(double) * 2 * n_cols, TV_STREAM);
    TAD_CHECK(e == hipSuccess, "%s: zeroing the statistics failed: %s", who, hipGetErrorString(e));
    hipLaunchKernelGGL(colstats_kernel<0>, dim3(tv_grid(rows)), dim3(TV_BLOCK), 0, TV_STREAM, rows, n_cols, cs, cond, ld, stats);
    hipLaunchKernelGGL(colstats_kernel<1>, dim3(tv_grid(rows)), dim3(TV_BLOCK), 0, TV_STREAM, rows, n_cols, cs, cond, ld, stats);
  }
  hipLaunchKernelGGL(metadata_noise_kernel, dim3(tv_grid(rows * F)), dim3(TV_BLOCK), 0, TV_STREAM, rows, (int)F, n_cols, cs, cond, ld,
                     noise, level, stats, out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}

}  // namespace

extern "C" int tad_metadata_noise(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const float* cond, int64_t ld,
                                  const float* noise, float level, double* stats, float* out, int64_t ldo,
                                  tad_stream_t stream) {
  return metadata_noise_launch("tad_metadata_noise", rows, F, n_cols, cols, cond, ld, noise, level, stats, out, ldo, stream);
}

// ---------------------------------------------------------------------------
// The same operation for callers that cannot allocate: the column statistics live in a device array of this
// code object instead of a caller-provided workspace.  Every GPU holds its own copy of the array at its own
// address; in host code the array's name is only a handle for hipGetSymbolAddress, never a device pointer.
namespace {

static __device__ __align__(256) double scratch[2 * TV_MAX_COLS];     // 1 KiB per GPU, held until the process ends

constexpr int TV_MAX_DEVICES = 64;
std::atomic<double*> g_scratch_ptr[TV_MAX_DEVICES];                    // device address of `scratch`, per device

// Device address of this GPU's copy of `scratch` (looked up once per device, with that device current).
int scratch_address(double** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  TAD_CHECK(e == hipSuccess && dev >= 0 && dev < TV_MAX_DEVICES, "tad_metadata_noise_static: hipGetDevice failed or device %d beyond %d: %s",
            dev, TV_MAX_DEVICES - 1, hipGetErrorString(e));
  double* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* sym = nullptr;
    e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
    TAD_CHECK(e == hipSuccess && sym, "tad_metadata_noise_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
    p = reinterpret_cast<double*>(sym);
    g_scratch_ptr[dev].store(p, std::memory_order_release);            // racing first calls store the same address
  }
  *out = p;
  return TAD_OK;
}

}  // namespace

extern "C" int tad_metadata_noise_static(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const float* cond,
                                         int64_t ld, const float* noise, float level, float* out, int64_t ldo,
                                         tad_stream_t stream) {
  TAD_CHECK(n_cols >= 0 && n_cols <= TV_MAX_COLS, "tad_metadata_noise_static: %d selected columns exceed the static scratch (at most %d)",
            n_cols, TV_MAX_COLS);
  // The scratch array that is used is the CURRENT device's copy, and so is the device the kernels are launched on: a
  // stream that belongs to another GPU is rejected here (its kernels could not reach this GPU's array).
  if (stream) {
    hipDevice_t sdev = -1;
    int cur = -1;
    const hipError_t e1 = hipStreamGetDevice((hipStream_t)stream, &sdev), e2 = hipGetDevice(&cur);
    TAD_CHECK(e1 == hipSuccess && e2 == hipSuccess, "tad_metadata_noise_static: cannot tell the stream's device: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    TAD_CHECK((int)sdev == cur, "tad_metadata_noise_static: the stream belongs to device %d, the current device is %d (make the stream's device current)", (int)sdev, cur);
  }
  double* stats = nullptr;
  if (int rc = scratch_address(&stats)) return rc;
  return metadata_noise_launch("tad_metadata_noise_static", rows, F, n_cols, cols, cond, ld, noise, level, stats, out, ldo, stream);
}

extern "C" int tad_flip_categories(int64_t rows, int64_t F, int n_cols, const int32_t* cols, const int32_t* classes,
                                   const float* cond, int64_t ld, const float* rnd, const int64_t* base, float p, float* out,
                                   int64_t ldo, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && F > 0 && F < (1 << 30) && cond && out && ld >= F && ldo >= F, "tad_flip_categories: bad arguments rows=%ld F=%ld ld=%ld ldo=%ld",
            (long)rows, (long)F, (long)ld, (long)ldo);
  TAD_CHECK(n_cols == 0 || (rnd && base && classes), "tad_flip_categories: draws and class counts are required when columns are selected");
  ColSet cs;
  int rc = fill_cols(cs, n_cols, (int)F, cols, classes, "tad_flip_categories");
  if (rc != TAD_OK) return rc;
  for (int j = 0; j < n_cols; ++j)
    TAD_CHECK(classes[j] >= 2, "tad_flip_categories: column %d has %d classes (columns with fewer than 2 are skipped by the caller)",
              (int)cols[j], (int)classes[j]);
  if (rows == 0) return TAD_OK;
  hipLaunchKernelGGL(flip_categories_kernel, dim3(tv_grid(rows * F)), dim3(TV_BLOCK), 0, TV_STREAM, rows, (int)F, n_cols, cs, cond, ld, rnd,
                     base, p, out, ldo);
  TAD_LAUNCH_CHECK();
  return TAD_OK;
}
