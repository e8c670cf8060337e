// ViT front / back end data-movement kernels for gfx950 (HBM-bound, 16-byte vectors where the
// layout allows).
//   patchify14         im2col of Conv2d(3, hidden, kernel 14, stride 14)
//                      M/core/models/vision/intern_vit_model.py:139-145,203-205 (H twin
//                      H/models/long_vita_qwen2_intern/modeling_intern_vit.py:96-101)
//   vit_assemble       cat(cls, patches) + position embedding, intern_vit_model.py:207-216
//   pixel_shuffle_ln   drop cls + pixel_shuffle(0.5) + LayerNorm(4h)
//                      M/pretrain_long_vita.py:467-483,572-582,443-446
//                      (H twin resampler_projector.py:26-46)
#include "vita_common.h"

namespace {

__global__ __launch_bounds__(256) void patchify14_kernel(const bf16_t* __restrict__ img,
                                                         bf16_t* __restrict__ out, int64_t n,
                                                         int H, int W, int k_pad) {
  const int gh = H / 14, gw = W / 14;
  const int64_t total = n * gh * gw * (int64_t)k_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % k_pad);
    const int64_t row = i / k_pad;
    bf16_t v = 0;
    if (col < 588) {
      const int px = (int)(row % gw);
      const int64_t t = row / gw;
      const int py = (int)(t % gh);
      const int64_t im = t / gh;
      const int c = col / 196, rem = col % 196, dy = rem / 14, dx = rem % 14;
      v = img[((im * 3 + c) * H + (py * 14 + dy)) * (int64_t)W + px * 14 + dx];
    }
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void vit_assemble_kernel(const u32x4* __restrict__ pe,
                                                           const u32x4* __restrict__ cls,
                                                           const u32x4* __restrict__ pos,
                                                           u32x4* __restrict__ x, int64_t n,
                                                           int n_patches, int nvec, int has_cls) {
  const int seq = n_patches + (has_cls ? 1 : 0);
  const int64_t total = n * seq * (int64_t)nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t r = i / nvec;
    const int s = (int)(r % seq);
    const int64_t im = r / seq;
    u32x4 a;
    if (has_cls && s == 0) a = cls[v];
    else a = pe[(im * n_patches + (s - (has_cls ? 1 : 0))) * nvec + v];
    const u32x4 b = pos[(int64_t)s * nvec + v];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf16lo_to_f32(a[j]) + bf16lo_to_f32(b[j]),
                         bf16hi_to_f32(a[j]) + bf16hi_to_f32(b[j]));
    x[i] = o;
  }
}

// one wave per output token; out channel block q*2h + r*h + c  <-  x[(2a'+q)*g + 2b'+r][c]
template <int VPL>
__global__ __launch_bounds__(256) void pixel_shuffle_ln_kernel(const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ bia,
                                                               bf16_t* __restrict__ y, int64_t n,
                                                               int g, int hidden, int has_cls,
                                                               float eps) {
  const int lane = threadIdx.x & 63;
  const int g2 = g >> 1;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n * g2 * g2) return;
  const int bp = (int)(tok % g2);
  const int64_t t = tok / g2;
  const int ap = (int)(t % g2);
  const int64_t im = t / g2;
  const int seq = g * g + (has_cls ? 1 : 0);
  const int hv = hidden >> 3;  // vectors per source token
  const int nvec = hv * 4;
  const int cols = hidden * 4;
  u32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      const int blk = vi / hv, cv = vi - blk * hv;
      const int q = blk >> 1, r = blk & 1;
      const int64_t src_tok = (int64_t)(2 * ap + q) * g + (2 * bp + r) + (has_cls ? 1 : 0);
      v[i] = *reinterpret_cast<const u32x4*>(x + (im * seq + src_tok) * hidden + cv * 8);
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
  const u32x4* br = reinterpret_cast<const u32x4*>(bia);
  u32x4* yr = reinterpret_cast<u32x4*>(y + tok * cols);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      const u32x4 wv = wr[vi];
      u32x4 bv = {0u, 0u, 0u, 0u};
      if (bia) bv = br[vi];
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

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  const int64_t cap = 256 * 32;
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

}  // namespace

extern "C" int vita_patchify14(const void* images, void* patches, int64_t n, int H, int W,
                               int k_pad, void* stream) {
  if (!images || !patches || n < 0 || H <= 0 || W <= 0) return VITA_ERR_INVALID_ARG;
  if (H % 14 || W % 14 || k_pad < 588) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int64_t total = n * (H / 14) * (W / 14) * (int64_t)k_pad;
  hipLaunchKernelGGL(patchify14_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)images, (bf16_t*)patches, n, H, W, k_pad);
  return vita_check_launch();
}

extern "C" int vita_vit_assemble(const void* patch_embeds, const void* cls_token,
                                 const void* pos_emb, void* x, int64_t n, int n_patches,
                                 int hidden, int has_cls, void* stream) {
  if (!patch_embeds || !pos_emb || !x || n < 0 || n_patches <= 0 || hidden <= 0)
    return VITA_ERR_INVALID_ARG;
  if (has_cls && !cls_token) return VITA_ERR_INVALID_ARG;
  if (hidden & 7) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int nvec = hidden >> 3;
  const int64_t total = n * (n_patches + (has_cls ? 1 : 0)) * (int64_t)nvec;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const u32x4*)patch_embeds, (const u32x4*)cls_token,
                     (const u32x4*)pos_emb, (u32x4*)x, n, n_patches, nvec, has_cls);
  return vita_check_launch();
}

extern "C" int vita_pixel_shuffle_ln(const void* x, const void* w, const void* b, void* y,
                                     int64_t n, int grid, int hidden, int has_cls, float eps,
                                     void* stream) {
  if (!x || !w || !y || n < 0 || grid <= 0 || hidden <= 0) return VITA_ERR_INVALID_ARG;
  if ((grid & 1) || (hidden & 7) || hidden * 4 > 8192) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int64_t toks = n * (grid / 2) * (grid / 2);
  dim3 g((unsigned)((toks + 3) / 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (hidden * 4 <= 4096)
    hipLaunchKernelGGL(pixel_shuffle_ln_kernel<8>, g, blk, 0, st, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, n, grid, hidden, has_cls, eps);
  else
    hipLaunchKernelGGL(pixel_shuffle_ln_kernel<16>, g, blk, 0, st, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, n, grid, hidden, has_cls, eps);
  return vita_check_launch();
}
