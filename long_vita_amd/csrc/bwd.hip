// Backward-pass kernels (training step, SURVEY.md §8a rows a9/a10/a12/a14/a15 backward halves) that
// are HBM-bound: transpose, RMSNorm backward, SwiGLU forward/backward, GELU backward, LayerNorm
// parameter gradients, vocabulary cross-entropy (+ its gradient), fp32 row scatter-add, and the
// attention-backward pre-pass D = rowsum(dO * O).
//
// The reference obtains all of these from torch autograd over the modules of the forward pass
// (no first-party backward code except M/core/tensor_parallel/layers.py:416-534); each kernel
// states the forward expression it differentiates, with the reference's bf16 rounding points.
#include "vita_common.h"

namespace {

inline unsigned grid_for(int64_t total, int block, int64_t cap = 256 * 16) {
  int64_t g = (total + block - 1) / block;
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

// ---------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c]   (bf16).  64x64 tiles through LDS (+1 padding), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, int64_t lds_,
                                                        bf16_t* __restrict__ dst, int64_t ldd,
                                                        int64_t R, int64_t C) {
  __shared__ bf16_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    const int64_t gr = r0 + r, gc = c0 + tx;
    tile[r][tx] = (gr < R && gc < C) ? src[gr * lds_ + gc] : (bf16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    const int64_t gc = c0 + c, gr = r0 + tx;
    if (gc < C && gr < R) dst[gc * ldd + gr] = tile[tx][c];
  }
}

// ---------------------------------------------------------------------------------------------
// RMSNorm backward.  forward (transformer_engine.py:74-79):  n = x_f * rstd (fp32) ;
//   o = bf16(n) ; y = bf16(o * w).      autograd:  do = bf16(dy * w) ; dn = float(do) ;
//   dx = bf16( rstd * (dn - n * mean(dn * n)) ) ;  dw = sum_rows bf16(dy * o).
// r03 kernel (cols <= 6144): a WORKGROUP per row (strided over rows), 256 lanes x VPL 16-byte vectors; sum(x^2) and sum(do * x) come
// out of ONE pass and one barrier (mean(do * n) = rstd * sum(do * x) / cols), the weight row lives in registers, every lane owns the
// same 8 VPL columns for all rows and sums their dw in registers; 512 workgroups = 2 per CU.  At the end dw goes through LDS so that a
// wave's atomic instruction covers 64 consecutive columns: all workgroups' atomics meet on the same cols / 32 cache lines, and it is
// line operations they cost (lane-strided: 0.15 ms per call; consecutive: 0.02 ms).  Measured at 5120 columns with dw and the
// residual branch, 16384 / 131072 rows: 0.17 / 1.13 ms = 3.9 / 4.7 TB/s (r02 kernel: 0.46 / 3.58 ms; profiles/r03_rmsnorm_bwd_ab.jsonl).
// rmsnorm_bwd_wave_kernel (r02: a wave per row, dw in 8 VPL registers per lane) keeps the wider rows.
// ---------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w, const bf16_t* __restrict__ res,
                                                             bf16_t* __restrict__ dx, float* __restrict__ dw_acc, int64_t rows,
                                                             int cols, float eps) {
  __shared__ float red[2 * 4 * 2];                   // [parity][wave][ss, dr]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nvec = cols >> 3;
  float dwl[VPL][8];                                 // this lane's columns 8 (tid + 256 i) .. + 7, summed over the workgroup's rows
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwl[i][j] = 0.f;
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  u32x4 wv4[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) wv4[i] = wr[vi];
  }
  int par = 0;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
    const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * (int64_t)cols);
    const u32x4* rr = reinterpret_cast<const u32x4*>(res + row * (int64_t)cols);
    u32x4 xv[VPL], gv[VPL], rv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = tid + i * 256;
      if (vi < nvec) {
        xv[i] = xr[vi]; gv[i] = gr[vi];
        if (res) rv[i] = rr[vi];
      }
    }
    float ss = 0.f, dr = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      if (tid + i * 256 < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16lo_to_f32(xv[i][j]), b = bf16hi_to_f32(xv[i][j]);
          const float d0 = bf16_round(bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(wv4[i][j]));
          const float d1 = bf16_round(bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(wv4[i][j]));
          ss += a * a + b * b;
          dr += d0 * a + d1 * b;
        }
      }
    }
    ss = wave_reduce_sum(ss); dr = wave_reduce_sum(dr);
    if (lane == 0) { red[(par * 4 + wv) * 2] = ss; red[(par * 4 + wv) * 2 + 1] = dr; }
    __syncthreads();
    float ss_t = 0.f, dr_t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ss_t += red[(par * 4 + k) * 2]; dr_t += red[(par * 4 + k) * 2 + 1]; }
    const float rstd = rsqrtf(ss_t / (float)cols + eps);
    const float c = dr_t * rstd / (float)cols;       // mean(do * n), n = x * rstd
    u32x4* dxr = reinterpret_cast<u32x4*>(dx + row * (int64_t)cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = tid + i * 256;
      if (vi < nvec) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = bf16lo_to_f32(xv[i][j]) * rstd, x1 = bf16hi_to_f32(xv[i][j]) * rstd;
          const float g0 = bf16lo_to_f32(gv[i][j]), g1 = bf16hi_to_f32(gv[i][j]);
          const float d0 = bf16_round(g0 * bf16lo_to_f32(wv4[i][j])), d1 = bf16_round(g1 * bf16hi_to_f32(wv4[i][j]));
          dwl[i][2 * j] += bf16_round(g0 * bf16_round(x0));
          dwl[i][2 * j + 1] += bf16_round(g1 * bf16_round(x1));
          float r0 = rstd * (d0 - x0 * c), r1 = rstd * (d1 - x1 * c);
          if (res) {
            r0 = bf16_round(r0) + bf16lo_to_f32(rv[i][j]);
            r1 = bf16_round(r1) + bf16hi_to_f32(rv[i][j]);
          }
          o[j] = pack_bf16x2(r0, r1);
        }
        dxr[vi] = o;
      }
    }
  }
  if (dw_acc) {
    // flush through LDS so that a wave's atomic instruction covers 64 CONSECUTIVE columns (two cache lines) instead of 64 columns
    // 32 bytes apart (sixteen lines): the atomics of all workgroups meet on the same cols / 32 lines, and line operations are what they cost
    extern __shared__ float dw_lds[];                // [8][nvec]: element e of vector vi at e * nvec + vi (bank = lane)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = tid + i * 256;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dw_lds[j * nvec + vi] = dwl[i][j];
      }
    }
    __syncthreads();
    for (int i = tid; i < cols; i += 256) atomicAdd(dw_acc + i, dw_lds[(i & 7) * nvec + (i >> 3)]);
  }
}


template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_bwd_wave_kernel(const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_acc, int64_t rows,
                                                          int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int nvec = cols >> 3;
  float dwl[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwl[i][j] = 0.f;
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  for (int64_t row = wave_id; row < rows; row += n_waves) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
    const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * (int64_t)cols);
    u32x4 xv[VPL], gv[VPL];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        xv[i] = xr[vi];
        gv[i] = gr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16lo_to_f32(xv[i][j]), b = bf16hi_to_f32(xv[i][j]);
          ss += a * a + b * b;
        }
      }
    }
    const float rstd = rsqrtf(wave_reduce_sum(ss) / (float)cols + eps);
    // pass 2: do = bf16(dy*w) ; c = mean(do * n) ; dw += bf16(dy * bf16(n))
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const u32x4 wv = wr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = bf16lo_to_f32(xv[i][j]) * rstd, x1 = bf16hi_to_f32(xv[i][j]) * rstd;
          const float g0 = bf16lo_to_f32(gv[i][j]), g1 = bf16hi_to_f32(gv[i][j]);
          const float d0 = bf16_round(g0 * bf16lo_to_f32(wv[j])), d1 = bf16_round(g1 * bf16hi_to_f32(wv[j]));
          dot += d0 * x0 + d1 * x1;
          dwl[i][2 * j] += bf16_round(g0 * bf16_round(x0));
          dwl[i][2 * j + 1] += bf16_round(g1 * bf16_round(x1));
        }
      }
    }
    const float c = wave_reduce_sum(dot) / (float)cols;
    u32x4* dxr = reinterpret_cast<u32x4*>(dx + row * (int64_t)cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const u32x4 wv = wr[vi];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = bf16lo_to_f32(xv[i][j]) * rstd, x1 = bf16hi_to_f32(xv[i][j]) * rstd;
          const float d0 = bf16_round(bf16lo_to_f32(gv[i][j]) * bf16lo_to_f32(wv[j]));
          const float d1 = bf16_round(bf16hi_to_f32(gv[i][j]) * bf16hi_to_f32(wv[j]));
          float r0 = rstd * (d0 - x0 * c), r1 = rstd * (d1 - x1 * c);
          if (res) {                                   // residual branch: dx = bf16(res + bf16(dx_norm))
            const unsigned rw = reinterpret_cast<const unsigned*>(res + row * (int64_t)cols)[vi * 4 + j];
            r0 = bf16_round(r0) + bf16lo_to_f32(rw);
            r1 = bf16_round(r1) + bf16hi_to_f32(rw);
          }
          o[j] = pack_bf16x2(r0, r1);
        }
        dxr[vi] = o;
      }
    }
  }
  if (dw_acc) {
    // reduce the workgroup's 4 waves through LDS first: one fp32 atomic per column and workgroup
    // (8192 waves x 5120 atomics on 5120 addresses made this kernel 20x slower than its HBM time)
    __shared__ float red[4][64][8];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wv][lane][j] = dwl[i][j];
      __syncthreads();
      if (wv == 0 && vi < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          atomicAdd(dw_acc + vi * 8 + j, (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]));
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// SwiGLU on the unfused fc1 output y = [gate | up]  ([rows, 2F] bf16):
//   fwd  a  = bf16( bf16(silu(g)) * u )
//   bwd  du = bf16(da * s), ds = bf16(da * u), dg = bf16(ds * silu'(g)),  s = bf16(silu(g))
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }
__device__ __forceinline__ float silu_grad_f(float g) {
  const float sg = 1.0f / (1.0f + __expf(-g));
  return sg * (1.0f + g * (1.0f - sg));
}

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ y,
                                                         bf16_t* __restrict__ a, int64_t rows, int F) {
  const int nv = F >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int v = (int)(i - r * nv);
    const u32x4 g = *reinterpret_cast<const u32x4*>(y + r * 2 * F + v * 8);
    const u32x4 u = *reinterpret_cast<const u32x4*>(y + r * 2 * F + F + v * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf16_round(silu_f(bf16lo_to_f32(g[j]))) * bf16lo_to_f32(u[j]),
                         bf16_round(silu_f(bf16hi_to_f32(g[j]))) * bf16hi_to_f32(u[j]));
    *reinterpret_cast<u32x4*>(a + r * F + v * 8) = o;
  }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ y,
                                                         const bf16_t* __restrict__ da,
                                                         bf16_t* __restrict__ dy, int64_t rows, int F) {
  const int nv = F >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int v = (int)(i - r * nv);
    const u32x4 g = *reinterpret_cast<const u32x4*>(y + r * 2 * F + v * 8);
    const u32x4 u = *reinterpret_cast<const u32x4*>(y + r * 2 * F + F + v * 8);
    const u32x4 d = *reinterpret_cast<const u32x4*>(da + r * F + v * 8);
    u32x4 og, ou;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = bf16lo_to_f32(g[j]), g1 = bf16hi_to_f32(g[j]);
      const float u0 = bf16lo_to_f32(u[j]), u1 = bf16hi_to_f32(u[j]);
      const float d0 = bf16lo_to_f32(d[j]), d1 = bf16hi_to_f32(d[j]);
      ou[j] = pack_bf16x2(d0 * bf16_round(silu_f(g0)), d1 * bf16_round(silu_f(g1)));
      og[j] = pack_bf16x2(bf16_round(d0 * u0) * silu_grad_f(g0), bf16_round(d1 * u1) * silu_grad_f(g1));
    }
    *reinterpret_cast<u32x4*>(dy + r * 2 * F + v * 8) = og;
    *reinterpret_cast<u32x4*>(dy + r * 2 * F + F + v * 8) = ou;
  }
}

// GELU (erf) backward: dx = bf16(dy * gelu'(x))
// tanh_form: d/dx [0.5 x (1 + tanh u)], u = c (x + 0.044715 x^3)  =  0.5 (1 + tanh u) + 0.5 x (1 - tanh^2 u) c (1 + 0.134145 x^2)   (SigLIP)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ dy,
                                                       bf16_t* __restrict__ dx, int64_t n8, int tanh_form) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
       i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 xv = reinterpret_cast<const u32x4*>(x)[i];
    const u32x4 gv = reinterpret_cast<const u32x4*>(dy)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float xx = h ? bf16hi_to_f32(xv[j]) : bf16lo_to_f32(xv[j]);
        const float gg = h ? bf16hi_to_f32(gv[j]) : bf16lo_to_f32(gv[j]);
        if (tanh_form) {
          const float th = tanhf(0.79788456080286535588f * (xx + 0.044715f * xx * xx * xx));
          r[h] = gg * (0.5f * (1.0f + th) + 0.5f * xx * (1.0f - th * th) * 0.79788456080286535588f * (1.0f + 0.134145f * xx * xx));
        } else {
          const float cdf = 0.5f * (1.0f + erff(xx * 0.70710678118654752440f));
          const float pdf = 0.39894228040143267794f * __expf(-0.5f * xx * xx);
          r[h] = gg * (cdf + xx * pdf);
        }
      }
      o[j] = pack_bf16x2(r[0], r[1]);
    }
    reinterpret_cast<u32x4*>(dx)[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// ViT training (reference stage 2 does not freeze the encoder): LayerNorm backward with parameter gradients, GELU forward
// as a kernel of its own (the backward needs the pre-activation), and the LayerScale residual of InternViTTransformerLayer.
// ---------------------------------------------------------------------------------------------
// LayerNorm backward (torch.nn.LayerNorm / TENorm; forward y = bf16(xhat * w + b), xhat = (x - mean) * rstd in fp32):
//   g = dy * w ; dx = bf16( rstd * (g - mean(g) - xhat * mean(g * xhat)) ) ; dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy.
// One wave per row (strided over rows); parameter gradients per lane in registers, flushed with fp32 atomics (caller zeroes).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int nvec = cols >> 3;
  float dgl[VPL][8], dbl[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dgl[i][j] = 0.f; dbl[i][j] = 0.f; }
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
  for (int64_t row = wave_id; row < rows; row += n_waves) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
    const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * (int64_t)cols);
    u32x4 xv[VPL], gv[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        xv[i] = xr[vi];
        gv[i] = gr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(xv[i][j]) + bf16hi_to_f32(xv[i][j]);
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
          dgl[i][2 * j] += d0 * x0; dgl[i][2 * j + 1] += d1 * x1;
          dbl[i][2 * j] += d0;      dbl[i][2 * j + 1] += d1;
        }
      }
    }
    const float c1 = wave_reduce_sum(sg) / (float)cols, c2 = wave_reduce_sum(sgx) / (float)cols;
    u32x4* dxr = reinterpret_cast<u32x4*>(dx + row * (int64_t)cols);
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
        dxr[vi] = o;
      }
    }
  }
  // flush (r03, as rmsnorm_bwd_kernel): the workgroup's four waves are summed in LDS and the atomics go out with consecutive lanes on
  // consecutive columns — one instruction touches two cache lines instead of sixteen, and every workgroup adds into the same lines
  extern __shared__ float pg_lds[];                  // [2][8][nvec]: element e of vector vi of dgamma | dbeta at e * nvec + vi
  const int wv = threadIdx.x >> 6;
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wv == w4) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float* pgm = pg_lds + j * nvec + vi;
            float* pbt = pg_lds + cols + j * nvec + vi;
            *pgm = (w4 ? *pgm : 0.f) + dgl[i][j];
            *pbt = (w4 ? *pbt : 0.f) + dbl[i][j];
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

// GELU forward as its own kernel: a = bf16(gelu(x)), erf form (tanh != 0: the tanh approximation of SigLIP)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ a, int64_t n8, int tanh_form) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 xv = reinterpret_cast<const u32x4*>(x)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float xx = h ? bf16hi_to_f32(xv[j]) : bf16lo_to_f32(xv[j]);
        r[h] = tanh_form ? 0.5f * xx * (1.0f + tanhf(0.79788456080286535588f * (xx + 0.044715f * xx * xx * xx)))
                         : 0.5f * xx * (1.0f + erff(xx * 0.70710678118654752440f));
      }
      o[j] = pack_bf16x2(r[0], r[1]);
    }
    reinterpret_cast<u32x4*>(a)[i] = o;
  }
}

// InternViTTransformerLayer's residual (M/core/models/vision/intern_vit_model.py:60-66,79-82), the module-by-module rounding chain:
//   t = bf16(x + bias) ; u = bf16(t * scale) ; out = bf16(residual + u)        (bias / scale may be null: SigLIP has no LayerScale)
__global__ __launch_bounds__(256) void bias_scale_res_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ bias,
                                                                 const bf16_t* __restrict__ scale, const bf16_t* __restrict__ res,
                                                                 bf16_t* __restrict__ out, int64_t rows, int cols) {
  const int nvec = cols >> 3;
  const int64_t total = rows * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int vi = (int)(i % nvec);
    const u32x4 xv = reinterpret_cast<const u32x4*>(x)[i], rv = reinterpret_cast<const u32x4*>(res)[i];
    u32x4 bv = {0u, 0u, 0u, 0u}, sv = {0u, 0u, 0u, 0u};
    if (bias) bv = reinterpret_cast<const u32x4*>(bias)[vi];
    if (scale) sv = reinterpret_cast<const u32x4*>(scale)[vi];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t0 = bf16lo_to_f32(xv[j]), t1 = bf16hi_to_f32(xv[j]);
      if (bias) { t0 = bf16_round(t0 + bf16lo_to_f32(bv[j])); t1 = bf16_round(t1 + bf16hi_to_f32(bv[j])); }
      if (scale) { t0 = bf16_round(t0 * bf16lo_to_f32(sv[j])); t1 = bf16_round(t1 * bf16hi_to_f32(sv[j])); }
      o[j] = pack_bf16x2(bf16lo_to_f32(rv[j]) + t0, bf16hi_to_f32(rv[j]) + t1);
    }
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
}

// its backward, g = d(out):  d_residual = g (the caller aliases it) ; dx = bf16(g * scale) ; d_bias += sum_rows dx ;
//   d_scale += sum_rows g * bf16(x + bias).   One workgroup = a block of rows x all columns; column sums per thread, then fp32 atomics.
__global__ __launch_bounds__(256) void bias_scale_res_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ scale,
                                                                 bf16_t* __restrict__ dx, float* __restrict__ d_bias,
                                                                 float* __restrict__ d_scale, int64_t rows, int cols, int rows_per_block) {
  extern __shared__ float bs_lds[];                  // the flush image [2][8][nvec]: element e of vector vi of d_bias | d_scale at e * nvec + vi
  const int nvec = cols >> 3;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int vi = threadIdx.x; vi < nvec; vi += blockDim.x) {
    u32x4 bv = {0u, 0u, 0u, 0u}, sv = {0u, 0u, 0u, 0u};
    if (bias) bv = reinterpret_cast<const u32x4*>(bias)[vi];
    if (scale) sv = reinterpret_cast<const u32x4*>(scale)[vi];
    float ab[8], as[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ab[j] = 0.f; as[j] = 0.f; }
    for (int64_t r = r0; r < r1; ++r) {
      const u32x4 gv = reinterpret_cast<const u32x4*>(g + r * (int64_t)cols)[vi];
      const u32x4 xv = reinterpret_cast<const u32x4*>(x + r * (int64_t)cols)[vi];
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g0 = bf16lo_to_f32(gv[j]), g1 = bf16hi_to_f32(gv[j]);
        float t0 = bf16lo_to_f32(xv[j]), t1 = bf16hi_to_f32(xv[j]);
        if (bias) { t0 = bf16_round(t0 + bf16lo_to_f32(bv[j])); t1 = bf16_round(t1 + bf16hi_to_f32(bv[j])); }
        const float d0 = scale ? bf16_round(g0 * bf16lo_to_f32(sv[j])) : g0, d1 = scale ? bf16_round(g1 * bf16hi_to_f32(sv[j])) : g1;
        ab[2 * j] += d0; ab[2 * j + 1] += d1;
        as[2 * j] += g0 * t0; as[2 * j + 1] += g1 * t1;
        o[j] = pack_bf16x2(d0, d1);
      }
      if (dx) reinterpret_cast<u32x4*>(dx + r * (int64_t)cols)[vi] = o;
    }
    // flush through LDS: consecutive lanes on consecutive columns (see rmsnorm_bwd_kernel)
#pragma unroll
    for (int j = 0; j < 8; ++j) { bs_lds[j * nvec + vi] = ab[j]; bs_lds[cols + j * nvec + vi] = as[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cols; i += blockDim.x) {
    const int at = (i & 7) * nvec + (i >> 3);
    if (d_bias) atomicAdd(d_bias + i, bs_lds[at]);
    if (d_scale) atomicAdd(d_scale + i, bs_lds[cols + at]);
  }
}

// LayerNorm parameter gradients (the projector pre-norm; its input comes from the frozen ViT, so dx
// is not needed): dgamma += sum_rows dy * xhat, dbeta += sum_rows dy   (fp32 atomics).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_param_grad_kernel(const bf16_t* __restrict__ dy,
                                                                   const bf16_t* __restrict__ x,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta,
                                                                   int64_t rows, int cols, float eps,
                                                                   int prenorm) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int nvec = cols >> 3;
  float dg[VPL][8], db[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
  for (int64_t row = wave_id; row < rows; row += n_waves) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
    const u32x4* gr = reinterpret_cast<const u32x4*>(dy + row * (int64_t)cols);
    u32x4 xv[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        xv[i] = xr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(xv[i][j]) + bf16hi_to_f32(xv[i][j]);
      }
    }
    float mean = wave_reduce_sum(s) / (float)cols;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16lo_to_f32(xv[i][j]) - mean, c = bf16hi_to_f32(xv[i][j]) - mean;
          ss += a * a + c * c;
        }
      }
    }
    float rstd = rsqrtf(wave_reduce_sum(ss) / (float)cols + eps);
    if (prenorm) { mean = 0.f; rstd = 1.f; }            // x already holds xhat
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const u32x4 gv = gr[vi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g0 = bf16lo_to_f32(gv[j]), g1 = bf16hi_to_f32(gv[j]);
          dg[i][2 * j] += g0 * (bf16lo_to_f32(xv[i][j]) - mean) * rstd;
          dg[i][2 * j + 1] += g1 * (bf16hi_to_f32(xv[i][j]) - mean) * rstd;
          db[i][2 * j] += g0;
          db[i][2 * j + 1] += g1;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(dgamma + vi * 8 + j, dg[i][j]);
        atomicAdd(dbeta + vi * 8 + j, db[i][j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Vocabulary cross-entropy on the selected rows (TP = 1 form of Megatron's
// vocab_parallel_cross_entropy, called at M/core/models/multimodal/gpt_vl_model.py:414):
//   loss[i] = logsumexp(float(logits[i, :])) - float(logits[i, label[i]])
//   dlogits[i, v] = bf16( (softmax_v - [v == label]) * grad_scale[i] )     (optional)
// One workgroup per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ce_ld(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ float ce_ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ void ce_st(bf16_t* p, int64_t i, float v) { p[i] = f32_to_bf16(v); }
__device__ __forceinline__ void ce_st(float* p, int64_t i, float v) { p[i] = v; }

// T = bf16_t (the stand-alone step: logits as the head GEMM left them) or float (Megatron hands vocab_parallel_cross_entropy
// `logits.float()`, language_module.compute_language_model_loss; the gradient then has to be fp32 too)
template <typename T>
__global__ __launch_bounds__(256) void ce_loss_kernel(const T* __restrict__ logits, int64_t ld,
                                                      const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss,
                                                      T* __restrict__ dlogits, int64_t ldd,
                                                      const float* __restrict__ grad_scale,
                                                      int V, int* __restrict__ err_flag) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, ce_ld(lr, v));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) se += __expf(ce_ld(lr, v) - mx);
  se = block_reduce_sum(se, red);
  const int64_t lab = labels[row];
  // A label outside [0, V) (the datasets pad with IGNORE_TOKEN_ID = -100, M/pretrain_long_vita.py:751): Megatron's
  // vocab_parallel_cross_entropy masks the target — predicted logit 0 AFTER the max subtraction, no one-hot term in the gradient —
  // and leaves the row to the caller's loss mask.  The same here; err_flag (when given) still reports that such a row was seen.
  const bool in_range = lab >= 0 && lab < V;
  if (!in_range && threadIdx.x == 0 && err_flag) atomicExch(err_flag, 1);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0 && loss) loss[row] = in_range ? lse - ce_ld(lr, lab) : __logf(se);
  if (dlogits) {
    const float gs = grad_scale ? grad_scale[row] : 1.0f;
    const float inv = 1.0f / se;
    T* dr = dlogits + row * ldd;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      float pr = __expf(ce_ld(lr, v) - mx) * inv;
      if (v == lab) pr -= 1.0f;
      ce_st(dr, v, pr * gs);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The vocabulary-PARALLEL form (TP > 1; megatron/core/tensor_parallel/cross_entropy.py): every rank holds [rows, V / TP] and never
// sees the other shards.  Three row passes, ONE collective between the first two:
//   stats   per row of the local shard: {max, sum exp(l - max), predicted raw logit (0 if the label is not in this shard), in-shard 0/1}
//           -> the caller all-gathers the [rows, 4] records of the TP group
//   finish  per row: global max, global sum-exp (each shard's sum rescaled), loss = log(sumexp) - (predicted - max) — or log(sumexp)
//           when no shard holds the label (Megatron's masked target) — and {max, sumexp} kept for the backward
//   grad    per row of the local shard: dlogits = (exp(l - max) / sumexp - [v == label - vocab_start]) * grad_scale
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_vp_stats_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                          int64_t vocab_start, float* __restrict__ stats, int V) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, ce_ld(lr, v));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) se += __expf(ce_ld(lr, v) - mx);
  se = block_reduce_sum(se, red);
  if (threadIdx.x == 0) {
    const int64_t lab = labels[row] - vocab_start;
    const bool mine = lab >= 0 && lab < V;
    float* st = stats + row * 4;
    st[0] = mx;
    st[1] = se;
    st[2] = mine ? ce_ld(lr, lab) : 0.f;
    st[3] = mine ? 1.f : 0.f;
  }
}

// stats_all [tp, rows, 4] (rank-major, as all_gather_into_tensor leaves it) -> loss [rows], row_stat [rows, 2] = {max, sumexp}
__global__ __launch_bounds__(256) void ce_vp_finish_kernel(const float* __restrict__ stats_all, int tp, int64_t rows,
                                                           float* __restrict__ loss, float* __restrict__ row_stat) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  float mx = -INFINITY;
  for (int r = 0; r < tp; ++r) mx = fmaxf(mx, stats_all[((int64_t)r * rows + row) * 4]);
  float se = 0.f, pred = 0.f, held = 0.f;
  for (int r = 0; r < tp; ++r) {
    const float* st = stats_all + ((int64_t)r * rows + row) * 4;
    se += st[1] * __expf(st[0] - mx);
    pred += st[2];
    held += st[3];
  }
  if (loss) loss[row] = __logf(se) - (held > 0.f ? pred - mx : 0.f);
  row_stat[row * 2] = mx;
  row_stat[row * 2 + 1] = se;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_vp_grad_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                         int64_t vocab_start, const float* __restrict__ row_stat,
                                                         const float* __restrict__ grad_scale, T* __restrict__ dlogits, int64_t ldd, int V) {
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  T* dr = dlogits + row * ldd;
  const float mx = row_stat[row * 2], inv = 1.0f / row_stat[row * 2 + 1];
  const float gs = grad_scale ? grad_scale[row] : 1.0f;
  const int64_t lab = labels[row] - vocab_start;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float pr = __expf(ce_ld(lr, v) - mx) * inv;
    if (v == lab) pr -= 1.0f;
    ce_st(dr, v, pr * gs);
  }
}

// dst_f32[idx[i], :] += float(src_bf16[i, :])   (embedding-weight gradient; atomics)
__global__ __launch_bounds__(256) void row_scatter_add_kernel(const bf16_t* __restrict__ src,
                                                              const int64_t* __restrict__ idx,
                                                              float* __restrict__ dst, int64_t dst_rows,
                                                              int64_t n, int cols, int* __restrict__ err_flag) {
  const int nv = cols >> 1;
  const int64_t total = n * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nv;
    const int v = (int)(i - r * nv);
    const int64_t d = idx[r];
    if (d < 0) continue;                                   // negative index = "skip this row"
    if (d >= dst_rows) {
      if (err_flag && v == 0) atomicExch(err_flag, 1);
      continue;
    }
    const unsigned w = *reinterpret_cast<const unsigned*>(src + r * cols + v * 2);
    atomicAdd(dst + d * cols + v * 2, bf16lo_to_f32(w));
    atomicAdd(dst + d * cols + v * 2 + 1, bf16hi_to_f32(w));
  }
}

// D[b, h, row] = sum_d float(dO[b,row,h,d]) * float(O[b,row,h,d])  — attention backward pre-pass.
// one wave per (row, head): 64 lanes x 2 (d=128) or x1 (d=64) elements
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o,
                                                         const bf16_t* __restrict__ d_o,
                                                         float* __restrict__ delta, int64_t rows,
                                                         int heads, int D, int64_t o_rs, int64_t o_hs,
                                                         int64_t do_rs, int64_t do_hs) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * heads) return;
  const int64_t row = item / heads;
  const int h = (int)(item - row * heads);
  const bf16_t* op = o + row * o_rs + (int64_t)h * o_hs;
  const bf16_t* dp = d_o + row * do_rs + (int64_t)h * do_hs;
  float s = 0.f;
  for (int e = lane * 2; e < D; e += 128) {
    const unsigned a = *reinterpret_cast<const unsigned*>(op + e);
    const unsigned b = *reinterpret_cast<const unsigned*>(dp + e);
    s += bf16lo_to_f32(a) * bf16lo_to_f32(b) + bf16hi_to_f32(a) * bf16hi_to_f32(b);
  }
  s = wave_reduce_sum(s);
  if (lane == 0) delta[(int64_t)h * rows + row] = s;
}

// the same for head sizes that are multiples of 8 with 16-byte aligned rows (64, 128: every kernel on the path): LPI = D / 8 lanes per
// (row, head) item, one 16-byte load per lane and tensor, 64 / LPI items per wave (r03: 2 x the rate of the 4-byte version)
template <int LPI>
__global__ __launch_bounds__(256) void attn_delta16_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o,
                                                           float* __restrict__ delta, int64_t rows, int heads, int64_t o_rs, int64_t o_hs,
                                                           int64_t do_rs, int64_t do_hs) {
  constexpr int IPW = 64 / LPI;
  const int lane = threadIdx.x & 63, sub = lane % LPI;
  const int64_t item = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * IPW + lane / LPI;
  float s = 0.f;
  int64_t row = 0;
  int h = 0;
  const bool live = item < rows * heads;
  if (live) {
    row = item / heads;
    h = (int)(item - row * heads);
    const u32x4 a = *reinterpret_cast<const u32x4*>(o + row * o_rs + (int64_t)h * o_hs + sub * 8);
    const u32x4 b = *reinterpret_cast<const u32x4*>(d_o + row * do_rs + (int64_t)h * do_hs + sub * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) s += bf16lo_to_f32(a[j]) * bf16lo_to_f32(b[j]) + bf16hi_to_f32(a[j]) * bf16hi_to_f32(b[j]);
  }
#pragma unroll
  for (int m = LPI / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (live && sub == 0) delta[(int64_t)h * rows + row] = s;
}

// out = bf16(a + b): the residual add behind a tensor-parallel all-reduce (bias_dropout_add with bias None, p = 0)
__global__ __launch_bounds__(256) void add_bf16_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                       u32x4* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 x = a[i], y = b[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf16lo_to_f32(x[j]) + bf16lo_to_f32(y[j]), bf16hi_to_f32(x[j]) + bf16hi_to_f32(y[j]));
    out[i] = o;
  }
}

}  // namespace

extern "C" int vita_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst,
                                   int64_t rows, int64_t cols, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0 || ld_src < cols || ld_dst < rows) return VITA_ERR_INVALID_ARG;
  if (rows == 0 || cols == 0) return VITA_OK;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  if (grid.y > 65535) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     ld_src, (bf16_t*)dst, ld_dst, rows, cols);
  return vita_check_launch();
}

extern "C" int vita_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* res, void* dx,
                                float* dw_acc, int64_t rows, int cols, float eps, void* stream) {
  if (!dy || !x || !w || !dx || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || cols > 8192) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  hipStream_t st = (hipStream_t)stream;
  const char* e = vita_dev_getenv("VITA_RMSNORM_BWD");           // developer A/B switch: "o" = the r02 wave-per-row kernel everywhere
  if (cols <= 6144 && !(e && e[0] == 'o')) {
    dim3 grid((unsigned)(rows < 512 ? rows : 512)), block(256);  // 2 workgroups per CU
#define VITA_RW(V) hipLaunchKernelGGL(rmsnorm_bwd_kernel<V>, grid, block, (size_t)cols * 4, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)res, (bf16_t*)dx, dw_acc, rows, cols, eps)
    if (cols <= 2048) VITA_RW(1); else if (cols <= 4096) VITA_RW(2); else VITA_RW(3);
#undef VITA_RW
    return vita_check_launch();
  }
  const int vpl = (cols + 511) / 512;
  dim3 grid((unsigned)((rows + 3) / 4 < 512 ? (rows + 3) / 4 : 512)), block(256);
#define VITA_RB(V) hipLaunchKernelGGL(rmsnorm_bwd_wave_kernel<V>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)res, (bf16_t*)dx, dw_acc, rows, cols, eps)
  if (vpl <= 2) VITA_RB(2); else if (vpl <= 4) VITA_RB(4); else if (vpl <= 8) VITA_RB(8); else if (vpl <= 10) VITA_RB(10); else VITA_RB(16);
#undef VITA_RB
  return vita_check_launch();
}

extern "C" int vita_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (!a || !b || !out || n < 0) return VITA_ERR_INVALID_ARG;
  if (n & 7) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a,
                     (const u32x4*)b, (u32x4*)out, n / 8);
  return vita_check_launch();
}

extern "C" int vita_swiglu_fwd(const void* y, void* a, int64_t rows, int ffn, void* stream) {
  if (!y || !a || rows < 0 || ffn <= 0) return VITA_ERR_INVALID_ARG;
  if (ffn & 7) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (ffn / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)a, rows, ffn);
  return vita_check_launch();
}

extern "C" int vita_swiglu_bwd(const void* y, const void* da, void* dy, int64_t rows, int ffn,
                               void* stream) {
  if (!y || !da || !dy || rows < 0 || ffn <= 0) return VITA_ERR_INVALID_ARG;
  if (ffn & 7) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (ffn / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)da, (bf16_t*)dy, rows, ffn);
  return vita_check_launch();
}

static int gelu_bwd_launch(const void* x, const void* dy, void* dx, int64_t n, int tanh_form, void* stream) {
  if (!x || !dy || !dx || n < 0) return VITA_ERR_INVALID_ARG;
  if (n & 7) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 8, tanh_form);
  return vita_check_launch();
}

extern "C" int vita_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  return gelu_bwd_launch(x, dy, dx, n, 0, stream);
}

extern "C" int vita_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  return gelu_bwd_launch(x, dy, dx, n, 1, stream);
}

extern "C" int vita_layernorm_bwd(const void* dy, const void* x, const void* w, void* dx, float* dgamma, float* dbeta,
                                  int64_t rows, int cols, float eps, void* stream) {
  if (!dy || !x || !w || !dx || !dgamma || !dbeta || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || cols > 8192) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  dim3 grid((unsigned)((rows + 3) / 4 < 512 ? (rows + 3) / 4 : 512)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define VITA_LB(V) hipLaunchKernelGGL(layernorm_bwd_kernel<V>, grid, block, (size_t)cols * 8, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)dx, dgamma, dbeta, rows, cols, eps)
  const int vpl = (cols + 511) / 512;
  static std::atomic<unsigned long long> attr_set{0};           // cols = 8192 needs 64 KiB of dynamic LDS for the flush image
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layernorm_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  });
  if (vpl <= 2) VITA_LB(2); else if (vpl <= 4) VITA_LB(4); else if (vpl <= 8) VITA_LB(8); else VITA_LB(16);
#undef VITA_LB
  return vita_check_launch();
}

extern "C" int vita_gelu_fwd(const void* x, void* a, int64_t n, int tanh_form, void* stream) {
  if (!x || !a || n < 0) return VITA_ERR_INVALID_ARG;
  if (n & 7) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)a,
                     n / 8, tanh_form);
  return vita_check_launch();
}

extern "C" int vita_bias_scale_res_fwd(const void* x, const void* bias, const void* scale, const void* residual, void* out,
                                       int64_t rows, int cols, void* stream) {
  if (!x || !residual || !out || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if (cols & 7) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  hipLaunchKernelGGL(bias_scale_res_fwd_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)bias, (const bf16_t*)scale, (const bf16_t*)residual, (bf16_t*)out, rows, cols);
  return vita_check_launch();
}

extern "C" int vita_bias_scale_res_bwd(const void* g, const void* x, const void* bias, const void* scale, void* dx, float* d_bias,
                                       float* d_scale, int64_t rows, int cols, void* stream) {
  if (!g || !x || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if (cols & 7) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  if (cols > 8192) return VITA_ERR_UNSUPPORTED;       // the flush image: 2 x cols floats of LDS
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bias_scale_res_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  });
  const int rpb = 64;
  hipLaunchKernelGGL(bias_scale_res_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(cols / 8 < 256 ? (cols / 8 + 63) / 64 * 64 : 256),
                     (size_t)cols * 8, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)bias, (const bf16_t*)scale,
                     (bf16_t*)dx, d_bias, d_scale, rows, cols, rpb);
  return vita_check_launch();
}

extern "C" int vita_layernorm_param_grad(const void* dy, const void* x, float* dgamma, float* dbeta,
                                         int64_t rows, int cols, float eps, int prenormalized,
                                         void* stream) {
  if (!dy || !x || !dgamma || !dbeta || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 7) || cols > 8192) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  dim3 grid((unsigned)((rows + 3) / 4 < 1024 ? (rows + 3) / 4 : 1024)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (cols <= 4096)
    hipLaunchKernelGGL(layernorm_param_grad_kernel<8>, grid, block, 0, st, (const bf16_t*)dy,
                       (const bf16_t*)x, dgamma, dbeta, rows, cols, eps, prenormalized);
  else
    hipLaunchKernelGGL(layernorm_param_grad_kernel<16>, grid, block, 0, st, (const bf16_t*)dy,
                       (const bf16_t*)x, dgamma, dbeta, rows, cols, eps, prenormalized);
  return vita_check_launch();
}

extern "C" int vita_ce_loss(const void* logits, int64_t ld, const int64_t* labels, float* loss,
                            void* dlogits, int64_t ld_d, const float* grad_scale, int64_t rows,
                            int vocab, int* err_flag, void* stream) {
  if (!logits || !labels || !loss || rows < 0 || vocab <= 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  if (rows > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ce_loss_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)logits, ld, labels, loss, (bf16_t*)dlogits, ld_d, grad_scale, vocab,
                     err_flag);
  return vita_check_launch();
}

// the same on fp32 logits / fp32 dlogits; `loss` may be NULL (a backward call that only wants dlogits)
extern "C" int vita_ce_loss_f32(const float* logits, int64_t ld, const int64_t* labels, float* loss,
                                float* dlogits, int64_t ld_d, const float* grad_scale, int64_t rows,
                                int vocab, int* err_flag, void* stream) {
  if (!logits || !labels || (!loss && !dlogits) || rows < 0 || vocab <= 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  if (rows > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ce_loss_kernel<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     logits, ld, labels, loss, dlogits, ld_d, grad_scale, vocab, err_flag);
  return vita_check_launch();
}

// ---- vocabulary-parallel cross entropy (ABI 17) ----
extern "C" int vita_ce_vp_stats(const void* logits, int is_f32, int64_t ld, const int64_t* labels, int64_t vocab_start, float* stats,
                                int64_t rows, int vocab_local, void* stream) {
  if (!logits || !labels || !stats || rows < 0 || vocab_local <= 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  if (rows > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  if (is_f32)
    hipLaunchKernelGGL(ce_vp_stats_kernel<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ld, labels,
                       vocab_start, stats, vocab_local);
  else
    hipLaunchKernelGGL(ce_vp_stats_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld,
                       labels, vocab_start, stats, vocab_local);
  return vita_check_launch();
}

extern "C" int vita_ce_vp_finish(const float* stats_all, int tp, int64_t rows, float* loss, float* row_stat, void* stream) {
  if (!stats_all || !row_stat || tp <= 0 || rows < 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  hipLaunchKernelGGL(ce_vp_finish_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, stats_all, tp, rows, loss, row_stat);
  return vita_check_launch();
}

extern "C" int vita_ce_vp_grad(const void* logits, int is_f32, int64_t ld, const int64_t* labels, int64_t vocab_start, const float* row_stat,
                               const float* grad_scale, void* dlogits, int64_t ld_d, int64_t rows, int vocab_local, void* stream) {
  if (!logits || !labels || !row_stat || !dlogits || rows < 0 || vocab_local <= 0) return VITA_ERR_INVALID_ARG;
  if (rows == 0) return VITA_OK;
  if (rows > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  if (is_f32)
    hipLaunchKernelGGL(ce_vp_grad_kernel<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ld, labels,
                       vocab_start, row_stat, grad_scale, (float*)dlogits, ld_d, vocab_local);
  else
    hipLaunchKernelGGL(ce_vp_grad_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels,
                       vocab_start, row_stat, grad_scale, (bf16_t*)dlogits, ld_d, vocab_local);
  return vita_check_launch();
}

extern "C" int vita_row_scatter_add_f32(const void* src, const int64_t* idx, float* dst,
                                        int64_t dst_rows, int64_t n, int cols, int* err_flag,
                                        void* stream) {
  if (!src || !idx || !dst || n < 0 || cols <= 0 || dst_rows < 0) return VITA_ERR_INVALID_ARG;
  if (cols & 1) return VITA_ERR_UNSUPPORTED;
  if (n == 0) return VITA_OK;
  hipLaunchKernelGGL(row_scatter_add_kernel, dim3(grid_for(n * (cols / 2), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, idx, dst, dst_rows, n, cols, err_flag);
  return vita_check_launch();
}

extern "C" int vita_attn_delta(const void* o, const void* d_o, float* delta, int64_t rows, int heads,
                               int head_dim, int64_t o_row_stride, int64_t o_head_stride,
                               int64_t do_row_stride, int64_t do_head_stride, void* stream) {
  if (!o || !d_o || !delta || rows < 0 || heads <= 0 || head_dim <= 0) return VITA_ERR_INVALID_ARG;
  if (head_dim & 1) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  const int64_t items = rows * heads;
  const bool wide = !((o_row_stride | o_head_stride | do_row_stride | do_head_stride) & 7) && !(((uintptr_t)o | (uintptr_t)d_o) & 15);
  if (wide && (head_dim == 128 || head_dim == 64)) {
    const int ipw = head_dim == 128 ? 4 : 8;                      // items per wave
    const dim3 grid((unsigned)((items + 4 * ipw - 1) / (4 * ipw)));
    if (head_dim == 128)
      hipLaunchKernelGGL(attn_delta16_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, (const bf16_t*)d_o, delta, rows,
                         heads, o_row_stride, o_head_stride, do_row_stride, do_head_stride);
    else
      hipLaunchKernelGGL(attn_delta16_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, (const bf16_t*)d_o, delta, rows,
                         heads, o_row_stride, o_head_stride, do_row_stride, do_head_stride);
    return vita_check_launch();
  }
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)o, (const bf16_t*)d_o, delta, rows, heads,
                     head_dim, o_row_stride, o_head_stride, do_row_stride, do_head_stride);
  return vita_check_launch();
}
