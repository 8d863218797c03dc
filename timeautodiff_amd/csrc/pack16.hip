// bf16 images for the bf16-stored-operand matrix products: row-major conversion of activations that another
// kernel left in fp32, and the padded (optionally transposed) weight images tad_mm16_nt reads.
#include "common.h"

namespace {

// 4 columns per thread: 16-byte loads, 8-byte stores (cols % 4 == 0 and 16-byte aligned rows), else element-wise
template <bool VEC>
__global__ void cvt_bf16_kernel(int64_t rows, int64_t cols, const float* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst, int64_t ldd,
                                const bf16_t* __restrict__ mask, int64_t ldm) {
  if constexpr (VEC) {
    const int64_t c4 = cols >> 2, n = rows * c4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = e / c4, c = (e - r * c4) << 2;
      float4 v = *reinterpret_cast<const float4*>(src + r * lds_ + c);
      if (mask) {
        const bf16x4 m = *reinterpret_cast<const bf16x4*>(mask + r * ldm + c);
        v.x = (float)m[0] > 0.f ? v.x : 0.f; v.y = (float)m[1] > 0.f ? v.y : 0.f;
        v.z = (float)m[2] > 0.f ? v.z : 0.f; v.w = (float)m[3] > 0.f ? v.w : 0.f;
      }
      bf16x4 o;
      o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
      *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = o;
    }
  } else {
    const int64_t n = rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = e / cols, c = e - r * cols;
      float v = src[r * lds_ + c];
      if (mask && !((float)mask[r * ldm + c] > 0.f)) v = 0.f;
      dst[r * ldd + c] = (bf16_t)v;
    }
  }
}

constexpr int PACK_MAXJOBS = 48;
struct PackJobs {
  tad_pack_w16_job_t j[PACK_MAXJOBS];
};

// blockIdx.y = job; the images are small (at most a few hundred KiB) and L2-resident, so the strided reads of a
// transposing job are not worth an LDS tile
__global__ void pack_w16_kernel(PackJobs a) {
  const tad_pack_w16_job_t& j = a.j[blockIdx.y];
  const int64_t n = j.n * j.k;
  bf16_t* dst = reinterpret_cast<bf16_t*>(j.dst);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    if (j.transpose) {          // dst [k, n]: consecutive threads write consecutive n
      const int64_t c = e / j.n, r = e - c * j.n;
      dst[c * j.ld_dst + r] = (bf16_t)j.src[r * j.ld_src + c];
    } else {
      const int64_t r = e / j.k, c = e - r * j.k;
      dst[r * j.ld_dst + c] = (bf16_t)j.src[r * j.ld_src + c];
    }
  }
}

}  // namespace

extern "C" int tad_cvt_bf16(int64_t rows, int64_t cols, const float* src, int64_t ld_src, void* dst16, int64_t ld_dst,
                            const void* relu_of16, int64_t ld_relu, tad_stream_t stream) {
  TAD_CHECK(rows >= 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "tad_cvt_bf16: bad shape rows=%ld cols=%ld ld_src=%ld ld_dst=%ld",
            (long)rows, (long)cols, (long)ld_src, (long)ld_dst);
  if (rows == 0) return TAD_OK;
  TAD_CHECK(src && dst16 && (!relu_of16 || ld_relu >= cols), "tad_cvt_bf16: null pointer or bad ld_relu");
  const bool vec = cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(dst16) & 7) == 0 && ld_relu % 4 == 0 && (reinterpret_cast<uintptr_t>(relu_of16) & 7) == 0;
  int64_t g = cdiv64(rows * (vec ? cols / 4 : cols), 256);
  if (g > 8192) g = 8192;
  if (vec)
    hipLaunchKernelGGL(cvt_bf16_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, rows, cols, src, ld_src, reinterpret_cast<bf16_t*>(dst16), ld_dst,
                       reinterpret_cast<const bf16_t*>(relu_of16), ld_relu);
  else
    hipLaunchKernelGGL(cvt_bf16_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, rows, cols, src, ld_src, reinterpret_cast<bf16_t*>(dst16), ld_dst,
                       reinterpret_cast<const bf16_t*>(relu_of16), ld_relu);
  TAD_LAUNCH_CHECK();
  tad_note_kernel("cvt_bf16_kernel");
  return TAD_OK;
}

extern "C" int tad_pack_w16(int njobs, const tad_pack_w16_job_t* jobs, tad_stream_t stream) {
  TAD_CHECK(njobs >= 0 && (njobs == 0 || jobs), "tad_pack_w16: bad job list");
  for (int base = 0; base < njobs; base += PACK_MAXJOBS) {
    const int cnt = njobs - base < PACK_MAXJOBS ? njobs - base : PACK_MAXJOBS;
    PackJobs a{};
    int64_t biggest = 0;
    for (int i = 0; i < cnt; ++i) {
      const tad_pack_w16_job_t& j = jobs[base + i];
      TAD_CHECK(j.src && j.dst && j.n > 0 && j.k > 0 && j.ld_src >= j.k && j.ld_dst >= (j.transpose ? j.n : j.k),
                "tad_pack_w16: job %d: bad shape n=%ld k=%ld ld_src=%ld ld_dst=%ld", base + i, (long)j.n, (long)j.k, (long)j.ld_src, (long)j.ld_dst);
      a.j[i] = j;
      if (j.n * j.k > biggest) biggest = j.n * j.k;
    }
    int64_t g = cdiv64(biggest, 256 * 4);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(pack_w16_kernel, dim3((unsigned)g, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a);
    TAD_LAUNCH_CHECK();
  }
  tad_note_kernel("pack_w16_kernel");
  return TAD_OK;
}
