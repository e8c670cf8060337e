// RoPE for gfx950: cos/sin table build + in-place apply (generic strided view, and the fused
// variant over Megatron's mixed QKV activation that also packs K/V for the CP all-gather).
// HBM-bound elementwise work: every lane moves 16-byte vectors (8 bf16).
//
// Reference arithmetic restated (M/core/models/common/embeddings/rotary_pos_embedding.py):
//   :98-106  freqs = outer(pos, inv_freq) (fp32); emb = cat(freqs, freqs)
//   :114-117 optional gather by position_ids  -> here: the caller passes the positions
//   :36-47   zig-zag CP slice                  -> here: the caller passes the rank's positions
//   :200-203 cos_ = cos(freqs).to(bf16); sin_ likewise; t = t*cos_ + rotate_half(t)*sin_
//            with every bf16 op rounding: bf16(bf16(t*cos) + bf16(rot*sin))
//   :169-171 rotate_half (non-interleaved): cat(-x2, x1)
#include "vita_common.h"

namespace {

__global__ void rope_table_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                  bf16_t* __restrict__ cos_out, bf16_t* __restrict__ sin_out,
                                  int64_t n, int half_dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half_dim) return;
  const int64_t r = i / half_dim;
  const int c = (int)(i - r * half_dim);
  const float f = (float)pos[r] * inv_freq[c];
  cos_out[i] = f32_to_bf16(cosf(f));
  sin_out[i] = f32_to_bf16(sinf(f));
}

// Megatron hands apply_rotary_pos_emb the fp32 angles `freqs` [s, 1, 1, dim] (emb = cat(freqs, freqs)): cos/sin of the first half
__global__ void rope_cos_sin_kernel(const float* __restrict__ freqs, int64_t row_stride, bf16_t* __restrict__ cos_out,
                                    bf16_t* __restrict__ sin_out, int64_t n, int half_dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half_dim) return;
  const int64_t r = i / half_dim;
  const float f = freqs[r * row_stride + (i - r * half_dim)];
  cos_out[i] = f32_to_bf16(cosf(f));
  sin_out[i] = f32_to_bf16(sinf(f));
}

__global__ __launch_bounds__(256) void rope_apply_kernel(bf16_t* __restrict__ t, int64_t rows,
                                                         int heads, int head_dim,
                                                         int64_t row_stride, int64_t head_stride,
                                                         const bf16_t* __restrict__ cos_tab,
                                                         const bf16_t* __restrict__ sin_tab,
                                                         float sign) {
  const int half = head_dim >> 1, nv = half >> 3;
  const int64_t total = rows * heads * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int vi = (int)(i % nv);
    const int64_t rh = i / nv;
    const int h = (int)(rh % heads);
    const int64_t r = rh / heads;
    bf16_t* p = t + r * row_stride + (int64_t)h * head_stride + vi * 8;
    u32x4 x1 = *reinterpret_cast<const u32x4*>(p);
    u32x4 x2 = *reinterpret_cast<const u32x4*>(p + half);
    const u32x4 c = *reinterpret_cast<const u32x4*>(cos_tab + r * half + vi * 8);
    const u32x4 s = *reinterpret_cast<const u32x4*>(sin_tab + r * half + vi * 8);
    rope_rotate8(x1, x2, c, s, sign);
    *reinterpret_cast<u32x4*>(p) = x1;
    *reinterpret_cast<u32x4*>(p + half) = x2;
  }
}

// mixed_qkv [rows, groups, (qpg + 2), d]: heads 0..qpg-1 = Q, qpg = K, qpg+1 = V.
__global__ __launch_bounds__(256) void rope_qkv_kernel(bf16_t* __restrict__ mixed, int64_t rows,
                                                       int groups, int qpg, int head_dim,
                                                       const bf16_t* __restrict__ cos_tab,
                                                       const bf16_t* __restrict__ sin_tab,
                                                       bf16_t* __restrict__ kv_out, float sign,
                                                       int kv_split) {
  const int half = head_dim >> 1, nv = half >> 3, hpg = qpg + 2;
  const int64_t per_row = (int64_t)groups * hpg * nv;
  const int64_t total = rows * per_row;
  // kv_out = [kv_split][2][rows][groups / kv_split][d]: one contiguous all-gather message per split
  const int hg = groups / kv_split;
  const int64_t kv_plane = rows * (int64_t)hg * head_dim;       // elements in K (or V) of one split
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    int rem = (int)(i - r * per_row);
    const int vi = rem % nv;
    rem /= nv;
    const int h = rem % hpg;
    const int g = rem / hpg;
    bf16_t* p = mixed + (r * groups + g) * (int64_t)hpg * head_dim + (int64_t)h * head_dim + vi * 8;
    u32x4 x1 = *reinterpret_cast<const u32x4*>(p);
    u32x4 x2 = *reinterpret_cast<const u32x4*>(p + half);
    if (h <= qpg) {
      const u32x4 c = *reinterpret_cast<const u32x4*>(cos_tab + r * half + vi * 8);
      const u32x4 s = *reinterpret_cast<const u32x4*>(sin_tab + r * half + vi * 8);
      rope_rotate8(x1, x2, c, s, sign);
      *reinterpret_cast<u32x4*>(p) = x1;
      *reinterpret_cast<u32x4*>(p + half) = x2;
    }
    if (kv_out && h >= qpg) {
      bf16_t* o = kv_out + ((g / hg) * 2 + (h - qpg)) * kv_plane + (r * hg + g % hg) * (int64_t)head_dim + vi * 8;
      *reinterpret_cast<u32x4*>(o) = x1;
      *reinterpret_cast<u32x4*>(o + half) = x2;
    }
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

}  // namespace

extern "C" int vita_rope_table(const int64_t* pos, const float* inv_freq, void* cos_out,
                               void* sin_out, int64_t n, int half_dim, void* stream) {
  if (!pos || !inv_freq || !cos_out || !sin_out || n < 0 || half_dim <= 0) return VITA_ERR_INVALID_ARG;
  if (n == 0) return VITA_OK;
  const int64_t total = n * half_dim;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pos, inv_freq, (bf16_t*)cos_out, (bf16_t*)sin_out, n,
                     half_dim);
  return vita_check_launch();
}

extern "C" int vita_rope_cos_sin(const float* freqs, int64_t row_stride, void* cos_out, void* sin_out, int64_t n,
                                 int half_dim, void* stream) {
  if (!freqs || !cos_out || !sin_out || n < 0 || half_dim <= 0 || row_stride < half_dim) return VITA_ERR_INVALID_ARG;
  if (n == 0) return VITA_OK;
  const int64_t total = n * half_dim;
  hipLaunchKernelGGL(rope_cos_sin_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, freqs,
                     row_stride, (bf16_t*)cos_out, (bf16_t*)sin_out, n, half_dim);
  return vita_check_launch();
}

extern "C" int vita_rope_apply(void* t, int64_t rows, int heads, int head_dim, int64_t row_stride,
                               int64_t head_stride, const void* cos_tab, const void* sin_tab,
                               int sign, void* stream) {
  if (!t || !cos_tab || !sin_tab || rows < 0 || heads <= 0 || head_dim <= 0) return VITA_ERR_INVALID_ARG;
  if ((head_dim & 15) || (row_stride & 7) || (head_stride & 7)) return VITA_ERR_UNSUPPORTED;
  if (sign != 1 && sign != -1) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  const int64_t total = rows * heads * (head_dim / 16);
  hipLaunchKernelGGL(rope_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (bf16_t*)t, rows, heads, head_dim, row_stride,
                     head_stride, (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, (float)sign);
  return vita_check_launch();
}

extern "C" int vita_rope_qkv_fwd(void* mixed_qkv, int64_t rows, int groups, int q_per_group,
                                 int head_dim, const void* cos_tab, const void* sin_tab,
                                 void* kv_out, int kv_split, void* stream) {
  if (!mixed_qkv || !cos_tab || !sin_tab || rows < 0 || groups <= 0 || q_per_group <= 0 ||
      head_dim <= 0)
    return VITA_ERR_INVALID_ARG;
  if (head_dim & 15) return VITA_ERR_UNSUPPORTED;
  if (kv_split <= 0 || groups % kv_split) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  const int64_t total = rows * groups * (q_per_group + 2) * (head_dim / 16);
  hipLaunchKernelGGL(rope_qkv_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (bf16_t*)mixed_qkv, rows, groups, q_per_group, head_dim,
                     (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, (bf16_t*)kv_out, 1.0f, kv_split);
  return vita_check_launch();
}

extern "C" int vita_rope_qkv_bwd(void* d_mixed_qkv, int64_t rows, int groups, int q_per_group,
                                 int head_dim, const void* cos_tab, const void* sin_tab, void* stream) {
  if (!d_mixed_qkv || !cos_tab || !sin_tab || rows < 0 || groups <= 0 || q_per_group <= 0 || head_dim <= 0)
    return VITA_ERR_INVALID_ARG;
  if (head_dim & 15) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  const int64_t total = rows * groups * (q_per_group + 2) * (head_dim / 16);
  hipLaunchKernelGGL(rope_qkv_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)d_mixed_qkv, rows, groups, q_per_group, head_dim, (const bf16_t*)cos_tab,
                     (const bf16_t*)sin_tab, (bf16_t*)nullptr, -1.0f, 1);
  return vita_check_launch();
}
