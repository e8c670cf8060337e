// Row gather / scatter, ordered mask compaction and the zig-zag context-parallel index remap.
// Pure integer / byte movement: results are bit-exact by construction.  HBM-bound; rows move as
// 16-byte vectors, consecutive lanes touch consecutive vectors of one row (coalesced).
//
// Reference call sites restated:
//   embedding lookup      M/core/tensor_parallel/layers.py:216-232
//   visual-token scatter  M/core/models/common/embeddings/language_model_embedding.py:123,126,131
//   masked_select/scatter M/core/tensor_parallel/layers.py:348,407,451,455
//   zig-zag index remap   M/training/utils.py:279-325,347-350
#include "vita_common.h"

namespace {

// dst[i] = src[idx[i]]  (SCATTER=false)   dst[didx[i]] = src[sidx ? sidx[i] : i]  (SCATTER=true)
template <bool SCATTER>
__global__ __launch_bounds__(256) void row_move_kernel(const u32x4* __restrict__ src,
                                                       int64_t src_rows,
                                                       const int64_t* __restrict__ sidx,
                                                       u32x4* __restrict__ dst, int64_t dst_rows,
                                                       const int64_t* __restrict__ didx, int64_t n,
                                                       int nvec, int* __restrict__ err_flag) {
  const int64_t total = n * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nvec;
    const int v = (int)(i - r * nvec);
    const int64_t sr = sidx ? sidx[r] : r;
    const int64_t dr = SCATTER ? didx[r] : r;
    if (sr < 0 || sr >= src_rows || dr < 0 || dr >= dst_rows) {
      if (err_flag && v == 0) atomicExch(err_flag, 1);
      continue;
    }
    dst[dr * nvec + v] = src[sr * nvec + v];
  }
}

// Single-workgroup ordered compaction (n <= 2^31): thread t owns the contiguous slice
// [t*per, (t+1)*per); slice counts are scanned in LDS, then every thread writes its hits in order.
__global__ __launch_bounds__(1024) void mask_to_index_kernel(const uint8_t* __restrict__ mask,
                                                             int64_t n,
                                                             int64_t* __restrict__ idx_out,
                                                             int64_t* __restrict__ count_out) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = (int64_t)t * per, hi = lo + per < n ? lo + per : n;
  int64_t c = 0;
  for (int64_t i = lo; i < hi; ++i) c += mask[i] != 0;
  part[t] = c;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 entries
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t add = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int64_t w = part[t] - c;  // exclusive prefix
  for (int64_t i = lo; i < hi; ++i)
    if (mask[i]) idx_out[w++] = i;
  if (t == 1023) *count_out = part[1023];
}

__global__ void cp_index_remap_kernel(const int64_t* __restrict__ indices_s, int64_t n,
                                      int64_t chunk, int cp_size, int cp_rank,
                                      uint8_t* __restrict__ hit, int64_t* __restrict__ local_pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = indices_s[i];
  const int64_t c = p >= 0 ? p / chunk : -1;
  const int64_t o = p - c * chunk;
  int64_t lp = -1;
  if (c == cp_rank) lp = o;
  else if (c == 2 * cp_size - 1 - cp_rank) lp = chunk + o;
  hit[i] = lp >= 0;
  local_pos[i] = lp;
}

__global__ void rows_any_kernel(const uint8_t* __restrict__ m, int64_t rows, int cols,
                                uint8_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  uint8_t a = 0;
  for (int c = 0; c < cols; ++c) a |= m[r * cols + c] != 0;
  out[r] = a;
}

__global__ void index_inverse_kernel(const int64_t* __restrict__ idx, int64_t n,
                                     int64_t* __restrict__ inv) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) inv[idx[k]] = k;
}

__global__ void cp_src_tgt_kernel(const int64_t* __restrict__ hit_idx, int64_t n_hit, int tok,
                                  const int64_t* __restrict__ img_rank,
                                  const int64_t* __restrict__ indices_b,
                                  const int64_t* __restrict__ local_pos,
                                  int64_t* __restrict__ src_b, int64_t* __restrict__ src_s,
                                  int64_t* __restrict__ tgt_b, int64_t* __restrict__ tgt_s) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_hit) return;
  const int64_t f = hit_idx[k];
  const int64_t img = f / tok;
  src_b[k] = img_rank[img];
  src_s[k] = f - img * tok;
  tgt_b[k] = indices_b[f];
  tgt_s[k] = local_pos[f];
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  const int64_t cap = 256 * 16;
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

}  // namespace

extern "C" int vita_row_gather(const void* src, int64_t src_rows, const int64_t* idx, void* dst,
                               int64_t n, int cols, int elem_bytes, int* err_flag, void* stream) {
  if (!src || !idx || !dst || n < 0 || cols <= 0 || elem_bytes <= 0 || src_rows < 0)
    return VITA_ERR_INVALID_ARG;
  const int64_t row_bytes = (int64_t)cols * elem_bytes;
  if (row_bytes & 15) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int nvec = (int)(row_bytes >> 4);
  hipLaunchKernelGGL(row_move_kernel<false>, dim3(grid_for(n * nvec, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const u32x4*)src, src_rows, idx, (u32x4*)dst, n,
                     (const int64_t*)nullptr, n, nvec, err_flag);
  return vita_check_launch();
}

extern "C" int vita_row_scatter(const void* src, int64_t src_rows, const int64_t* src_idx,
                                void* dst, int64_t dst_rows, const int64_t* dst_idx, int64_t n,
                                int cols, int elem_bytes, int* err_flag, void* stream) {
  if (!src || !dst || !dst_idx || n < 0 || cols <= 0 || elem_bytes <= 0 || src_rows < 0 ||
      dst_rows < 0)
    return VITA_ERR_INVALID_ARG;
  const int64_t row_bytes = (int64_t)cols * elem_bytes;
  if (row_bytes & 15) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  const int nvec = (int)(row_bytes >> 4);
  hipLaunchKernelGGL(row_move_kernel<true>, dim3(grid_for(n * nvec, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const u32x4*)src, src_rows, src_idx, (u32x4*)dst,
                     dst_rows, dst_idx, n, nvec, err_flag);
  return vita_check_launch();
}

extern "C" int vita_mask_to_index(const uint8_t* mask, int64_t n, int64_t* idx_out,
                                  int64_t* count_out, void* stream) {
  if (!mask || !idx_out || !count_out || n < 0) return VITA_ERR_INVALID_ARG;
  if (n > ((int64_t)1 << 31)) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mask_to_index_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, n,
                     idx_out, count_out);
  return vita_check_launch();
}

extern "C" int vita_cp_index_remap(const int64_t* indices_s, int64_t n, int64_t seq_len,
                                   int cp_size, int cp_rank, uint8_t* hit, int64_t* local_pos,
                                   void* stream) {
  if (!indices_s || !hit || !local_pos || n < 0 || seq_len <= 0 || cp_size <= 0 || cp_rank < 0 ||
      cp_rank >= cp_size)
    return VITA_ERR_INVALID_ARG;
  if (seq_len % (2 * cp_size)) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(cp_index_remap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, indices_s, n, seq_len / (2 * cp_size), cp_size, cp_rank,
                     hit, local_pos);
  return vita_check_launch();
}

extern "C" int vita_rows_any(const uint8_t* mask, int64_t rows, int cols, uint8_t* out,
                             void* stream) {
  if (!mask || !out || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  hipLaunchKernelGGL(rows_any_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, mask, rows, cols, out);
  return vita_check_launch();
}

extern "C" int vita_index_inverse(const int64_t* idx, int64_t n, int64_t* inv, void* stream) {
  if (!idx || !inv || n < 0) return VITA_ERR_INVALID_ARG;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(index_inverse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, idx, n, inv);
  return vita_check_launch();
}

extern "C" int vita_cp_src_tgt(const int64_t* hit_idx, int64_t n_hit, int tok_per_img,
                               const int64_t* img_rank, const int64_t* indices_b,
                               const int64_t* local_pos, int64_t* src_b, int64_t* src_s,
                               int64_t* tgt_b, int64_t* tgt_s, void* stream) {
  if (!hit_idx || !img_rank || !indices_b || !local_pos || !src_b || !src_s || !tgt_b || !tgt_s ||
      n_hit < 0 || tok_per_img <= 0)
    return VITA_ERR_INVALID_ARG;
  if (n_hit == 0) return VITA_OK;
  hipLaunchKernelGGL(cp_src_tgt_kernel, dim3((unsigned)((n_hit + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, hit_idx, n_hit, tok_per_img, img_rank, indices_b,
                     local_pos, src_b, src_s, tgt_b, tgt_s);
  return vita_check_launch();
}
