// Frame preprocessing: the step before the ViT.
// Reference: ImageProcessor.process_images, H/data/processor/image_processor.py:180-223 — per frame
//   expand2square (pad to a square with int(mean * 255), :189-201) -> PIL BICUBIC resize to 448 x 448 (:206-208)
//   -> float32 (x * 1.0 / 255.0 - mean) / std (:210-215) -> CHW; M/tasks/inference/module.py:693 then casts to bf16.
// The reference does this per frame on the rank-0 CPU (4096 frames for a 1M-token video) and broadcasts 4.9 GB.
//
// Pillow's resize of 8-bit images (src/libImaging/Resample.c, Pillow 12.x) is integer arithmetic:
//   two separable passes (horizontal, then vertical), each  out = clip8((2^21 + sum_x pixel[x] * kk[x]) >> 22)
//   with 22-bit fixed-point coefficients and a uint8 intermediate.  The coefficient tables are built on the host
//   (long_vita_amd/image_processor.py, double arithmetic as in precompute_coeffs / normalize_coeffs_8bpc); the two
//   kernels below apply them, so the uint8 result is bit-exact and the bf16 output equals the reference's.
// HBM-bound byte work: algorithmic bytes per frame = H*W*3 (read) + P*448*3 (intermediate, write + read) + 3*448*448*2.
#include "vita_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass over the (virtually padded) Ph x Pw source: tmp[n][y][xx][c], y in [0, Ph), xx in [0, out_w)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ frames, int64_t frame_stride, int H,
                                                         int W, int Ph, int ox, int oy, int out_w, int pad0, int pad1,
                                                         int pad2, const int* __restrict__ bounds,
                                                         const int* __restrict__ kk, int ksize,
                                                         uint8_t* __restrict__ tmp) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, n = blockIdx.z;
  if (xx >= out_w) return;
  const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
  const int* k = kk + (int64_t)xx * ksize;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  const int iy = y - oy;
  const bool row_in = iy >= 0 && iy < H;
  const uint8_t* row = frames + n * frame_stride + (int64_t)(row_in ? iy : 0) * W * 3;
  for (int x = 0; x < cnt; ++x) {
    const int ix = xmin + x - ox;
    int p0 = pad0, p1 = pad1, p2 = pad2;
    if (row_in && ix >= 0 && ix < W) { p0 = row[ix * 3]; p1 = row[ix * 3 + 1]; p2 = row[ix * 3 + 2]; }
    const int c = k[x];
    s0 += p0 * c; s1 += p1 * c; s2 += p2 * c;
  }
  uint8_t* o = tmp + (((int64_t)n * Ph + y) * out_w + xx) * 3;
  o[0] = (uint8_t)clip8(s0 >> kPrecisionBits);
  o[1] = (uint8_t)clip8(s1 >> kPrecisionBits);
  o[2] = (uint8_t)clip8(s2 >> kPrecisionBits);
}

// vertical pass + normalisation.  The out_h x out_w result is cut into tile x tile images (row-major blocks, the crop
// order of dynamic_preprocess): images[(n * tiles + block)][c][yy % tile][xx % tile] bf16; u8_out keeps the uncut result.
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int Ph, int out_w, int out_h,
                                                              int tile, int canvas_w, int canvas_h, int off_x, int off_y,
                                                              const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize, float m0, float m1,
                                                              float m2, float sd0, float sd1, float sd2,
                                                              bf16_t* __restrict__ images, uint8_t* __restrict__ u8_out) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y, n = blockIdx.z;
  if (xx >= out_w) return;
  const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const int* k = kk + (int64_t)yy * ksize;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  const uint8_t* col = tmp + (((int64_t)n * Ph + ymin) * out_w + xx) * 3;
  for (int y = 0; y < cnt; ++y) {
    const int c = k[y];
    s0 += (int)col[0] * c; s1 += (int)col[1] * c; s2 += (int)col[2] * c;
    col += (int64_t)out_w * 3;
  }
  const int v0 = clip8(s0 >> kPrecisionBits), v1 = clip8(s1 >> kPrecisionBits), v2 = clip8(s2 >> kPrecisionBits);
  if (u8_out) {
    uint8_t* u = u8_out + (((int64_t)n * out_h + yy) * out_w + xx) * 3;
    u[0] = (uint8_t)v0; u[1] = (uint8_t)v1; u[2] = (uint8_t)v2;
  }
  // float32, same operation order as numpy: (x * 1.0 / 255.0 - mean) / std, then round-to-nearest-even to bf16
  // position on the canvas the resized image is pasted onto (resize_and_pad_image :351-360; canvas == image otherwise)
  const int cx = xx + off_x, cy = yy + off_y;
  const int tiles_x = canvas_w / tile, tiles = tiles_x * (canvas_h / tile);
  const int block = (cy / tile) * tiles_x + cx / tile;
  const int64_t plane = (int64_t)tile * tile;
  bf16_t* o = images + ((int64_t)n * tiles + block) * 3 * plane + (int64_t)(cy % tile) * tile + (cx % tile);
  o[0] = f32_to_bf16(__fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), sd0));
  o[plane] = f32_to_bf16(__fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), sd1));
  o[2 * plane] = f32_to_bf16(__fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), sd2));
}

}  // namespace

extern "C" int vita_frames_resize_norm(const void* frames, int64_t frame_stride, int n, int height, int width,
                                       int pad_to_square, const int* pad_rgb, int out_w, int out_h, int tile,
                                       int canvas_w, int canvas_h, int off_x, int off_y, const void* h_bounds, const void* h_coeffs, int h_ksize, const void* v_bounds,
                                       const void* v_coeffs, int v_ksize, const float* mean, const float* std_, void* tmp,
                                       void* images, void* u8_out, void* stream) {
  if (!frames || !h_bounds || !h_coeffs || !v_bounds || !v_coeffs || !tmp || !images || !pad_rgb || !mean || !std_)
    return VITA_ERR_INVALID_ARG;
  if (n < 0 || height <= 0 || width <= 0 || out_w <= 0 || out_h <= 0 || tile <= 0 || h_ksize <= 0 || v_ksize <= 0)
    return VITA_ERR_INVALID_ARG;
  if (canvas_w <= 0) { canvas_w = out_w; canvas_h = out_h; off_x = off_y = 0; }      // no canvas: the image itself
  if (canvas_w % tile || canvas_h % tile || off_x < 0 || off_y < 0 || off_x + out_w > canvas_w || off_y + out_h > canvas_h)
    return VITA_ERR_INVALID_ARG;
  if (n == 0) return VITA_OK;
  const int P = height > width ? height : width;
  const int Pw = pad_to_square ? P : width, Ph = pad_to_square ? P : height;     // expand2square (:189-201) or as is
  if (n > 65535 || out_h > 65535 || Ph > 65535) return VITA_ERR_UNSUPPORTED;
  const int ox = (Pw - width) / 2, oy = (Ph - height) / 2;                        // paste offsets (:195, :199)
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(256);
  const unsigned gx = (unsigned)((out_w + 255) / 256);
  hipLaunchKernelGGL(resample_h_kernel, dim3(gx, (unsigned)Ph, (unsigned)n), block, 0, st, (const uint8_t*)frames,
                     frame_stride, height, width, Ph, ox, oy, out_w, pad_rgb[0], pad_rgb[1], pad_rgb[2],
                     (const int*)h_bounds, (const int*)h_coeffs, h_ksize, (uint8_t*)tmp);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(gx, (unsigned)out_h, (unsigned)n), block, 0, st, (const uint8_t*)tmp, Ph,
                     out_w, out_h, tile, canvas_w, canvas_h, off_x, off_y, (const int*)v_bounds, (const int*)v_coeffs, v_ksize, mean[0], mean[1], mean[2],
                     std_[0], std_[1], std_[2], (bf16_t*)images, (uint8_t*)u8_out);
  return vita_check_launch();
}
