// Logit post-processing of GPTVLModel.forward (M/core/models/multimodal/gpt_vl_model.py:349-355):
//   logits = logits * output_multiplier_scale;  logits = tanh(logits / cap) * cap     (each step a bf16 tensor op in the reference:
// the bf16 rounding after every step is reproduced), in place on the [rows, vocab] bf16 logits of the selected rows.
// Backward: d/dx [cap * tanh(scale * x / cap)] = scale * (1 - (y / cap)^2), from the stored output y.
// HBM-bound elementwise work on a few rows x 152064 columns: 8 bf16 per lane.
#include "vita_common.h"

namespace {

__device__ __forceinline__ float post1(float x, float scale, float cap) {
  if (scale != 0.f) x = bf16_round(x * scale);
  if (cap != 0.f) {
    x = bf16_round(x / cap);
    x = bf16_round(tanhf(x));
    x = bf16_round(x * cap);
  }
  return x;
}

__global__ __launch_bounds__(256) void logit_post_kernel(bf16_t* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                         float scale, float cap) {
  const int64_t nv = cols >> 3, total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv, c = (i - r * nv) << 3;
    u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ld + c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = pack_bf16x2(post1(bf16lo_to_f32(v[j]), scale, cap), post1(bf16hi_to_f32(v[j]), scale, cap));
    *reinterpret_cast<u32x4*>(x + r * ld + c) = v;
  }
}

__global__ __launch_bounds__(256) void logit_post_bwd_kernel(const bf16_t* __restrict__ y, int64_t ldy, bf16_t* __restrict__ g,
                                                             int64_t ldg, int64_t rows, int64_t cols, float scale, float cap) {
  const int64_t nv = cols >> 3, total = rows * nv;
  const float s = scale != 0.f ? scale : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv, c = (i - r * nv) << 3;
    const u32x4 yv = *reinterpret_cast<const u32x4*>(y + r * ldy + c);
    u32x4 gv = *reinterpret_cast<const u32x4*>(g + r * ldg + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f0 = s, f1 = s;
      if (cap != 0.f) {
        const float t0 = bf16lo_to_f32(yv[j]) / cap, t1 = bf16hi_to_f32(yv[j]) / cap;
        f0 *= 1.f - t0 * t0;
        f1 *= 1.f - t1 * t1;
      }
      gv[j] = pack_bf16x2(bf16lo_to_f32(gv[j]) * f0, bf16hi_to_f32(gv[j]) * f1);
    }
    *reinterpret_cast<u32x4*>(g + r * ldg + c) = gv;
  }
}

unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); }

}  // namespace

extern "C" int vita_logit_postprocess(void* logits, int64_t ld, int64_t rows, int64_t cols, float multiplier_scale,
                                      float softcapping, void* stream) {
  if (!logits || rows < 0 || cols <= 0 || ld < cols) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || (ld & 7)) return VITA_ERR_UNSUPPORTED;
  if (rows == 0 || (multiplier_scale == 0.f && softcapping == 0.f)) return VITA_OK;
  hipLaunchKernelGGL(logit_post_kernel, dim3(grid_for(rows * (cols >> 3))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)logits,
                     rows, cols, ld, multiplier_scale, softcapping);
  return vita_check_launch();
}

extern "C" int vita_logit_postprocess_bwd(const void* y, int64_t ldy, void* grad, int64_t ldg, int64_t rows, int64_t cols,
                                          float multiplier_scale, float softcapping, void* stream) {
  if (!y || !grad || rows < 0 || cols <= 0 || ldy < cols || ldg < cols) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || (ldy & 7) || (ldg & 7)) return VITA_ERR_UNSUPPORTED;
  if (rows == 0 || (multiplier_scale == 0.f && softcapping == 0.f)) return VITA_OK;
  hipLaunchKernelGGL(logit_post_bwd_kernel, dim3(grid_for(rows * (cols >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                     ldy, (bf16_t*)grad, ldg, rows, cols, multiplier_scale, softcapping);
  return vita_check_launch();
}
