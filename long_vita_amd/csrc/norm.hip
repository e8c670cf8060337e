// RMSNorm / LayerNorm forward for gfx950.  HBM-bound: one 64-lane wave owns one row, every lane
// keeps its 16-byte vectors in registers (single read of x, single write of y), reductions are
// wave shuffles only (no LDS, no barrier).  Algorithmic bytes per row: 2 * cols * 2 B (+ w).
//
// Reference arithmetic restated:
//   RMSNorm  M/core/transformer/custom_layers/transformer_engine.py:74-79
//            y = bf16( bf16( float(x) * rsqrt(mean(float(x)^2) + eps) ) * w )
//   LayerNorm torch.nn.LayerNorm on bf16 (fp32 statistics, one rounding at the end)
#include "vita_common.h"

namespace {

constexpr int kRowsPerBlock = 4;  // 4 waves / 256 threads

template <int VPL>  // 16-byte vectors per lane; cols <= VPL * 512
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y,
                                                          float* __restrict__ rstd_out,
                                                          int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = cols >> 3;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
  u32x4 v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      v[i] = xr[vi];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16lo_to_f32(v[i][j]), b = bf16hi_to_f32(v[i][j]);
        ss += a * a + b * b;
      }
    }
  }
  ss = wave_reduce_sum(ss);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  u32x4* yr = reinterpret_cast<u32x4*>(y + row * (int64_t)cols);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      const u32x4 wv = wr[vi];
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_round(bf16lo_to_f32(v[i][j]) * rstd) * bf16lo_to_f32(wv[j]);
        const float b = bf16_round(bf16hi_to_f32(v[i][j]) * rstd) * bf16hi_to_f32(wv[j]);
        o[j] = pack_bf16x2(a, b);
      }
      yr[vi] = o;
    }
  }
}

template <int VPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b,
                                                            bf16_t* __restrict__ y, int64_t rows,
                                                            int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = cols >> 3;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
  u32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      v[i] = xr[vi];
#pragma unroll
      for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(v[i][j]) + bf16hi_to_f32(v[i][j]);
    }
  }
  const float mean = wave_reduce_sum(s) / (float)cols;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16lo_to_f32(v[i][j]) - mean, c = bf16hi_to_f32(v[i][j]) - mean;
        ss += a * a + c * c;
      }
    }
  }
  const float rstd = rsqrtf(wave_reduce_sum(ss) / (float)cols + eps);
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  const u32x4* br = reinterpret_cast<const u32x4*>(b);
  u32x4* yr = reinterpret_cast<u32x4*>(y + row * (int64_t)cols);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      const u32x4 wv = wr[vi];
      u32x4 bv = {0u, 0u, 0u, 0u};
      if (b) bv = br[vi];
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a0 = (bf16lo_to_f32(v[i][j]) - mean) * rstd * bf16lo_to_f32(wv[j]) + bf16lo_to_f32(bv[j]);
        const float a1 = (bf16hi_to_f32(v[i][j]) - mean) * rstd * bf16hi_to_f32(wv[j]) + bf16hi_to_f32(bv[j]);
        o[j] = pack_bf16x2(a0, a1);
      }
      yr[vi] = o;
    }
  }
}

}  // namespace

#define VITA_NORM_DISPATCH(KERNEL, ...)                                                          \
  do {                                                                                           \
    const int vpl = (cols + 511) / 512;                                                          \
    dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);               \
    hipStream_t st = (hipStream_t)stream;                                                        \
    if (vpl <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, st, __VA_ARGS__);                \
    else if (vpl <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, st, __VA_ARGS__);           \
    else if (vpl <= 8) hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, st, __VA_ARGS__);           \
    else if (vpl <= 10) hipLaunchKernelGGL(KERNEL<10>, grid, block, 0, st, __VA_ARGS__);         \
    else hipLaunchKernelGGL(KERNEL<16>, grid, block, 0, st, __VA_ARGS__);                        \
  } while (0)

extern "C" int vita_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd_out,
                                int64_t rows, int cols, float eps, void* stream) {
  if (!x || !w || !y || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || cols > 8192) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  VITA_NORM_DISPATCH(rmsnorm_fwd_kernel, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd_out,
                     rows, cols, eps);
  return vita_check_launch();
}

extern "C" int vita_layernorm_fwd(const void* x, const void* w, const void* b, void* y,
                                  int64_t rows, int cols, float eps, void* stream) {
  if (!x || !w || !y || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || cols > 8192) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  VITA_NORM_DISPATCH(layernorm_fwd_kernel, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b,
                     (bf16_t*)y, rows, cols, eps);
  return vita_check_launch();
}
