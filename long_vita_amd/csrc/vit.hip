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

// sbh = 0: output row = (image, patch) image-major;  sbh = 1: row = patch * n + image (Megatron's [s, b, h] order, so that the
// GEMM output IS rows 1.. of the encoder input and the backward's gradient rows line up with these rows without a transpose)
__global__ __launch_bounds__(256) void patchify14_kernel(const bf16_t* __restrict__ img,
                                                         bf16_t* __restrict__ out, int64_t n,
                                                         int H, int W, int k_pad, int sbh) {
  const int gh = H / 14, gw = W / 14;
  const int64_t total = n * gh * gw * (int64_t)k_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % k_pad);
    const int64_t row = i / k_pad;
    bf16_t v = 0;
    if (col < 588) {
      const int64_t pr = sbh ? row / n : row % ((int64_t)gh * gw);       // patch index inside the image
      const int64_t im = sbh ? row % n : row / ((int64_t)gh * gw);
      const int px = (int)(pr % gw);
      const int py = (int)(pr / gw);
      const int c = col / 196, rem = col % 196, dy = rem / 14, dx = rem % 14;
      v = img[((im * 3 + c) * H + (py * 14 + dy)) * (int64_t)W + px * 14 + dx];
    }
    out[i] = v;
  }
}

// sbh = 1: pe rows and x rows are token-major (row = token * n + image); pos_row0 = first row of the position table that is used
// (InternViTModel without a class token looks up rows 1 .. seq: intern_vit_model.py:141-143)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const u32x4* __restrict__ pe,
                                                           const u32x4* __restrict__ cls,
                                                           const u32x4* __restrict__ pos,
                                                           u32x4* __restrict__ x, int64_t n,
                                                           int n_patches, int nvec, int has_cls, int pos_row0, int sbh) {
  const int seq = n_patches + (has_cls ? 1 : 0);
  const int64_t total = n * seq * (int64_t)nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t r = i / nvec;
    const int s = (int)(sbh ? r / n : r % seq);
    const int64_t im = sbh ? r % n : r / seq;
    const int ps = s - (has_cls ? 1 : 0);
    u32x4 a;
    if (has_cls && s == 0) a = cls[v];
    else a = pe[(sbh ? (int64_t)ps * n + im : im * n_patches + ps) * nvec + v];
    const u32x4 b = pos[(int64_t)(s + pos_row0) * nvec + v];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf16lo_to_f32(a[j]) + bf16lo_to_f32(b[j]),
                         bf16hi_to_f32(a[j]) + bf16hi_to_f32(b[j]));
    x[i] = o;
  }
}

// one wave per output token; out channel block q*2h + r*h + c  <-  x[(2a'+q)*g + 2b'+r][c]
// x is addressed as x[image * img_stride + token * tok_stride + c] (elements): [n, seq, h] contiguous or the [s, b, h] layout the
// encoder leaves (a permuted view, no copy).  NORM = false: the pure permutation (forward_downsample alone).
template <int VPL, bool NORM>
__global__ __launch_bounds__(256) void pixel_shuffle_ln_kernel(const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ bia,
                                                               bf16_t* __restrict__ y, int64_t n,
                                                               int g, int hidden, int has_cls,
                                                               float eps, int64_t img_stride, int64_t tok_stride) {
  const int lane = threadIdx.x & 63;
  const int g2 = g >> 1;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n * g2 * g2) return;
  const int bp = (int)(tok % g2);
  const int64_t t = tok / g2;
  const int ap = (int)(t % g2);
  const int64_t im = t / g2;
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
      v[i] = *reinterpret_cast<const u32x4*>(x + im * img_stride + src_tok * tok_stride + cv * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(v[i][j]) + bf16hi_to_f32(v[i][j]);
    }
  }
  u32x4* yr = reinterpret_cast<u32x4*>(y + tok * cols);
  if (!NORM) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) yr[vi] = v[i];
    }
    return;
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

// Backward of pixel_shuffle_ln_kernel.  One wave per output token (strided over tokens): the token's 4 h source values are gathered
// again, mean / rstd recomputed, and with g = dy * w:
//     dx[source positions] = bf16( rstd * (g - mean(g) - xhat * mean(g * xhat)) ),  dgamma += dy * xhat,  dbeta += dy
// (NORM = false: dx[source positions] = dy).  dx (may be NULL: frozen encoder, parameter gradients only) has x's strides; the class
// token's row receives zeros (vit_output[:, 1:, :] drops it, M/pretrain_long_vita.py:454-455).  Parameter gradients: per-lane
// registers, summed over the workgroup's waves in LDS, flushed with fp32 atomics on consecutive columns (caller zeroes them).
template <int VPL, bool NORM>
__global__ __launch_bounds__(256) void pixel_shuffle_ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ w, bf16_t* __restrict__ dx,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   int64_t n, int g, int hidden, int has_cls, float eps,
                                                                   int64_t img_stride, int64_t tok_stride) {
  const int lane = threadIdx.x & 63;
  const int g2 = g >> 1;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t toks = n * g2 * g2;
  const int hv = hidden >> 3, nvec = hv * 4, cols = hidden * 4;
  float dgl[NORM ? VPL : 1][8], dbl[NORM ? VPL : 1][8];
#pragma unroll
  for (int i = 0; i < (NORM ? VPL : 1); ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dgl[i][j] = 0.f; dbl[i][j] = 0.f; }
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  for (int64_t tok = wave_id; tok < toks; tok += n_waves) {
    const int bp = (int)(tok % g2);
    const int64_t t = tok / g2;
    const int ap = (int)(t % g2);
    const int64_t im = t / g2;
    const u32x4* gr = reinterpret_cast<const u32x4*>(dy + tok * (int64_t)cols);
    int64_t soff[VPL];
    u32x4 xv[VPL], gv[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      soff[i] = 0;
      if (vi < nvec) {
        const int blk = vi / hv, cv = vi - blk * hv;
        const int q = blk >> 1, r = blk & 1;
        const int64_t src_tok = (int64_t)(2 * ap + q) * g + (2 * bp + r) + (has_cls ? 1 : 0);
        soff[i] = im * img_stride + src_tok * tok_stride + cv * 8;
        gv[i] = gr[vi];
        if (NORM) {
          xv[i] = *reinterpret_cast<const u32x4*>(x + soff[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(xv[i][j]) + bf16hi_to_f32(xv[i][j]);
        }
      }
    }
    if (has_cls && dx && ap == 0 && bp == 0) {                     // the class token's row of this image: no gradient
      const u32x4 z = {0u, 0u, 0u, 0u};
      for (int c = lane; c < hv; c += 64) *reinterpret_cast<u32x4*>(dx + im * img_stride + c * 8) = z;
    }
    if (!NORM) {
      if (dx) {
#pragma unroll
        for (int i = 0; i < VPL; ++i)
          if (lane + i * 64 < nvec) *reinterpret_cast<u32x4*>(dx + soff[i]) = gv[i];
      }
      continue;
    }
    const float mean = wave_reduce_sum(s) / (float)cols;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16lo_to_f32(xv[i][j]) - mean, c = bf16hi_to_f32(xv[i][j]) - mean;
          ss += a * a + c * c;
        }
      }
    }
    const float rstd = rsqrtf(wave_reduce_sum(ss) / (float)cols + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const u32x4 wv = wr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = (bf16lo_to_f32(xv[i][j]) - mean) * rstd, x1 = (bf16hi_to_f32(xv[i][j]) - mean) * rstd;
          const float d0 = bf16lo_to_f32(gv[i][j]), d1 = bf16hi_to_f32(gv[i][j]);
          const float g0 = d0 * bf16lo_to_f32(wv[j]), g1 = d1 * bf16hi_to_f32(wv[j]);
          sg += g0 + g1;
          sgx += g0 * x0 + g1 * x1;
          dgl[NORM ? i : 0][2 * j] += d0 * x0; dgl[NORM ? i : 0][2 * j + 1] += d1 * x1;
          dbl[NORM ? i : 0][2 * j] += d0;      dbl[NORM ? i : 0][2 * j + 1] += d1;
        }
      }
    }
    if (dx) {
      const float c1 = wave_reduce_sum(sg) / (float)cols, c2 = wave_reduce_sum(sgx) / (float)cols;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
          const u32x4 wv = wr[vi];
          u32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x0 = (bf16lo_to_f32(xv[i][j]) - mean) * rstd, x1 = (bf16hi_to_f32(xv[i][j]) - mean) * rstd;
            const float g0 = bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(wv[j]), g1 = bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(wv[j]);
            o[j] = pack_bf16x2(rstd * (g0 - c1 - x0 * c2), rstd * (g1 - c1 - x1 * c2));
          }
          *reinterpret_cast<u32x4*>(dx + soff[i]) = o;
        }
      }
    }
  }
  if (!NORM) return;
  extern __shared__ float pg_lds[];                  // [2][8][nvec]: element e of vector vi of dgamma | dbeta at e * nvec + vi
  const int wv4 = threadIdx.x >> 6;
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wv4 == w4) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float* pgm = pg_lds + j * nvec + vi;
            float* pbt = pg_lds + cols + j * nvec + vi;
            *pgm = (w4 ? *pgm : 0.f) + dgl[NORM ? i : 0][j];
            *pbt = (w4 ? *pbt : 0.f) + dbl[NORM ? i : 0][j];
          }
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < cols; i += 256) {
    const int at = (i & 7) * nvec + (i >> 3);
    atomicAdd(dgamma + i, pg_lds[at]);
    atomicAdd(dbeta + i, pg_lds[cols + at]);
  }
}

// d_pos[s, :] = bf16( sum over images of dx[image, s, :] ) — the position table's gradient (and, row 0, the class token's):
// x = cat(cls, patches) + pos broadcast over the batch (intern_vit_model.py:207-216).  One thread per (token, 8 columns), fp32 sums.
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const u32x4* __restrict__ dx, u32x4* __restrict__ d_pos, int64_t n,
                                                               int seq, int nvec, int sbh) {
  const int64_t total = (int64_t)seq * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t s = i / nvec;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t im = 0; im < n; ++im) {
      const u32x4 a = dx[(sbh ? s * n + im : im * seq + s) * nvec + v];
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16lo_to_f32(a[j]); acc[2 * j + 1] += bf16hi_to_f32(a[j]); }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(acc[2 * j], acc[2 * j + 1]);
    d_pos[i] = o;
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  const int64_t cap = 256 * 32;
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

}  // namespace

extern "C" int vita_patchify14_ex(const void* images, void* patches, int64_t n, int H, int W,
                                  int k_pad, int token_major, void* stream) {
  if (!images || !patches || n < 0 || H <= 0 || W <= 0) return VITA_ERR_INVALID_ARG;
  if (H % 14 || W % 14 || k_pad < 588) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int64_t total = n * (H / 14) * (W / 14) * (int64_t)k_pad;
  hipLaunchKernelGGL(patchify14_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)images, (bf16_t*)patches, n, H, W, k_pad, token_major);
  return vita_check_launch();
}

extern "C" int vita_patchify14(const void* images, void* patches, int64_t n, int H, int W,
                               int k_pad, void* stream) {
  return vita_patchify14_ex(images, patches, n, H, W, k_pad, 0, stream);
}

extern "C" int vita_vit_assemble_ex(const void* patch_embeds, const void* cls_token,
                                    const void* pos_emb, void* x, int64_t n, int n_patches,
                                    int hidden, int has_cls, int pos_row0, int token_major, void* stream) {
  if (!patch_embeds || !pos_emb || !x || n < 0 || n_patches <= 0 || hidden <= 0 || pos_row0 < 0)
    return VITA_ERR_INVALID_ARG;
  if (has_cls && !cls_token) return VITA_ERR_INVALID_ARG;
  if (hidden & 7) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int nvec = hidden >> 3;
  const int64_t total = n * (n_patches + (has_cls ? 1 : 0)) * (int64_t)nvec;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const u32x4*)patch_embeds, (const u32x4*)cls_token,
                     (const u32x4*)pos_emb, (u32x4*)x, n, n_patches, nvec, has_cls, pos_row0, token_major);
  return vita_check_launch();
}

extern "C" int vita_vit_assemble(const void* patch_embeds, const void* cls_token,
                                 const void* pos_emb, void* x, int64_t n, int n_patches,
                                 int hidden, int has_cls, void* stream) {
  return vita_vit_assemble_ex(patch_embeds, cls_token, pos_emb, x, n, n_patches, hidden, has_cls, 0, 0, stream);
}

extern "C" int vita_vit_assemble_bwd(const void* dx, void* d_pos, int64_t n, int seq, int hidden, int token_major, void* stream) {
  if (!dx || !d_pos || n < 0 || seq <= 0 || hidden <= 0) return VITA_ERR_INVALID_ARG;
  if (hidden & 7) return VITA_ERR_UNSUPPORTED;
  const int nvec = hidden >> 3;
  hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(grid_for((int64_t)seq * nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)dx, (u32x4*)d_pos, n, seq, nvec, token_major);
  return vita_check_launch();
}

extern "C" int vita_pixel_shuffle_ln_ex(const void* x, const void* w, const void* b, void* y,
                                        int64_t n, int grid, int hidden, int has_cls, float eps, int norm,
                                        int64_t img_stride, int64_t tok_stride, void* stream) {
  if (!x || !y || (norm && !w) || n < 0 || grid <= 0 || hidden <= 0) return VITA_ERR_INVALID_ARG;
  if ((grid & 1) || (hidden & 7) || hidden * 4 > 8192 || ((img_stride | tok_stride) & 7)) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int64_t toks = n * (grid / 2) * (grid / 2);
  dim3 g((unsigned)((toks + 3) / 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define VITA_PS(V, N) hipLaunchKernelGGL((pixel_shuffle_ln_kernel<V, N>), g, blk, 0, st, (const bf16_t*)x, (const bf16_t*)w, \
                                         (const bf16_t*)b, (bf16_t*)y, n, grid, hidden, has_cls, eps, img_stride, tok_stride)
  if (hidden * 4 <= 4096) { if (norm) VITA_PS(8, true); else VITA_PS(8, false); }
  else { if (norm) VITA_PS(16, true); else VITA_PS(16, false); }
#undef VITA_PS
  return vita_check_launch();
}

extern "C" int vita_pixel_shuffle_ln(const void* x, const void* w, const void* b, void* y,
                                     int64_t n, int grid, int hidden, int has_cls, float eps,
                                     void* stream) {
  const int64_t seq = (int64_t)grid * grid + (has_cls ? 1 : 0);
  return vita_pixel_shuffle_ln_ex(x, w, b, y, n, grid, hidden, has_cls, eps, 1, seq * hidden, hidden, stream);
}

extern "C" int vita_pixel_shuffle_ln_bwd(const void* dy, const void* x, const void* w, void* dx, float* dgamma, float* dbeta,
                                         int64_t n, int grid, int hidden, int has_cls, float eps, int norm,
                                         int64_t img_stride, int64_t tok_stride, void* stream) {
  if (!dy || n < 0 || grid <= 0 || hidden <= 0) return VITA_ERR_INVALID_ARG;
  if (norm && (!x || !w || !dgamma || !dbeta)) return VITA_ERR_INVALID_ARG;
  if (!norm && !dx) return VITA_ERR_INVALID_ARG;
  if ((grid & 1) || (hidden & 7) || hidden * 4 > 4096 || ((img_stride | tok_stride) & 7)) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int64_t toks = n * (grid / 2) * (grid / 2);
  const int cols = hidden * 4;
  dim3 g((unsigned)((toks + 3) / 4 < 512 ? (toks + 3) / 4 : 512)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (norm)
    hipLaunchKernelGGL((pixel_shuffle_ln_bwd_kernel<8, true>), g, blk, (size_t)cols * 8, st, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)dx, dgamma, dbeta, n, grid, hidden, has_cls, eps, img_stride, tok_stride);
  else
    hipLaunchKernelGGL((pixel_shuffle_ln_bwd_kernel<8, false>), g, blk, 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)dx, dgamma, dbeta, n, grid, hidden, has_cls, eps, img_stride, tok_stride);
  return vita_check_launch();
}
