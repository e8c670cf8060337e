// Single-token decode against a sequence-sharded KV cache: the caller of
// the prefill path, M/inference/text_generation/generation.py:123-205 with --use-kv-cache, which the
// reference switches OFF under context parallelism (server_cp .sh:184) and re-prefills instead.
//
// Every kernel here is HBM-bound:
//   gemv            : N*K*2 bytes of weights per token (x stays in L2)
//   decode attention: len * 2 * G * d * 2 bytes of cache per token and layer, split over
//                     ceil(len/256) x G workgroups; partial (max, sum, O) per workgroup are merged by
//                     a second kernel, which also merges the per-rank partials after the CP gather.
#include "vita_common.h"

#include <math.h>

namespace {

__device__ __forceinline__ float dot8(const u32x4 a, const u32x4 b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    s += bf16lo_to_f32(a[j]) * bf16lo_to_f32(b[j]) + bf16hi_to_f32(a[j]) * bf16hi_to_f32(b[j]);
  return s;
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {      // a.lo*b.lo + a.hi*b.hi + c, fp32
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// ---- y[N] = epilogue(W[N,K] . x[K]) : one wave per output, 16-byte pieces of the W row per lane ----
template <int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                   int64_t ldw, bf16_t* __restrict__ y, int64_t N, int64_t K,
                                                   const bf16_t* __restrict__ bias,
                                                   const bf16_t* __restrict__ R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= N) return;
  const int nvec = (int)(K >> 3);
  const u32x4* xr = reinterpret_cast<const u32x4*>(x);
  const u32x4* w0 = reinterpret_cast<const u32x4*>(W + n * ldw);
  const u32x4* w1 = reinterpret_cast<const u32x4*>(W + (n + N) * ldw);      // SWIGLU: the "up" row
  float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
  for (int v = lane; v < nvec; v += 64) {
    const u32x4 xv = xr[v];
    a0 += dot8(xv, w0[v]);
    if (EPI == VITA_EPI_SWIGLU) a1 += dot8(xv, w1[v]);
  }
  a0 = wave_reduce_sum(a0);
  if (EPI == VITA_EPI_SWIGLU) a1 = wave_reduce_sum(a1);
  if (lane != 0) return;
  float o = a0;
  if (EPI == VITA_EPI_BIAS) o += bf16_to_f32(bias[n]);
  if (EPI == VITA_EPI_RESIDUAL) o = bf16_round(o) + bf16_to_f32(R[n]);
  if (EPI == VITA_EPI_SWIGLU) {                     // same rounding chain as the GEMM epilogue
    const float g = bf16_round(a0), u = bf16_round(a1);
    o = bf16_round(g / (1.0f + __expf(-g))) * u;
  }
  y[n] = f32_to_bf16(o);
}

// ---- decode attention, stage 1 ---------------------------------------------------------------
// One workgroup = a contiguous range of cached keys of one kv group; its 4 waves run independently
// (no barrier, no LDS in the loop) over interleaved 32-key (16 for qpg >= 7) tiles with their own online softmax:
//   lane = (kq = lane/16, sub = lane%16): the 16 lanes of a row-group read one 256-byte K/V row per
//   instruction (4 rows per wave instruction), 8 K and 8 V row loads in flight per lane;
//   q.k = 4 x v_dot2_f32_bf16 per head and row + a 4-step DPP rotate-add that leaves the score in all
//   16 lanes — exactly the lanes that need p for their 8 output dims in P.V, so scores never leave
//   registers.  The waves' (m, l, O) are merged through LDS once at the end.
constexpr int kDecKeys = 256;        // key granularity of a split (host side: ops.DECODE_KEYS_PER_TILE)

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float row16_allreduce_sum(float v) {   // row_ror:8,4,2,1 inside each 16-lane row
  v = dpp_add<0x128>(v);
  v = dpp_add<0x124>(v);
  v = dpp_add<0x122>(v);
  v = dpp_add<0x121>(v);
  return v;
}

template <int QPG>
__global__ __launch_bounds__(256, 2) void decode_attn_partial_kernel(
    const bf16_t* __restrict__ q, int64_t q_group_stride, int64_t q_head_stride,
    const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc, int64_t kv_row_stride,
    int64_t kv_group_stride, int len, const int* __restrict__ len_dev, float scale_log2,
    float* __restrict__ pm, float* __restrict__ pl, float* __restrict__ po) {
  constexpr int D = 128;
  constexpr int ITS = QPG >= 7 ? 4 : 8;                   // K/V row loads in flight per lane (register budget)
  constexpr int kWaveKeys = ITS * 4;
  __shared__ float wm[4][QPG], wl[4][QPG];
  __shared__ float wo[4][QPG][D];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int sub = lane & 15, kq = lane >> 4;
  const int blk = blockIdx.x, g = blockIdx.y;
  const int H = gridDim.y * QPG;
  if (len_dev) len = min(len, *len_dev);                  // row count kept on the device (graph replay)
  const int ntiles = (len + kDecKeys - 1) / kDecKeys;
  const int tiles_per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int key_begin = blk * tiles_per_wg * kDecKeys;
  const int key_end = min(len, key_begin + tiles_per_wg * kDecKeys);
  if (key_begin >= len) {                                 // empty split: neutral element of the merge
    if (t < QPG) { pm[(int64_t)blk * H + g * QPG + t] = -INFINITY; pl[(int64_t)blk * H + g * QPG + t] = 0.f; }
    if (t < D)
      for (int h = 0; h < QPG; ++h) po[((int64_t)blk * H + g * QPG + h) * D + t] = 0.f;
    return;
  }

  unsigned qp[QPG][4];                                    // packed bf16 pairs, consumed by v_dot2_f32_bf16
#pragma unroll
  for (int h = 0; h < QPG; ++h) {
    const u32x4 qv = *reinterpret_cast<const u32x4*>(q + g * q_group_stride + h * q_head_stride + sub * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) qp[h][j] = qv[j];
  }
  float mrun[QPG], lrun[QPG], acc[QPG][8];
#pragma unroll
  for (int h = 0; h < QPG; ++h) {
    mrun[h] = -INFINITY;
    lrun[h] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[h][i] = 0.f;
  }
  const bf16_t* kbase = kc + g * kv_group_stride + sub * 8;
  const bf16_t* vbase = vc + g * kv_group_stride + sub * 8;

  for (int key0 = key_begin + wave * kWaveKeys; key0 < key_end; key0 += 4 * kWaveKeys) {
    u32x4 kr[ITS], vr[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int key = key0 + it * 4 + kq;
      kr[it] = u32x4{0u, 0u, 0u, 0u};
      vr[it] = u32x4{0u, 0u, 0u, 0u};
      if (key < key_end) {
        kr[it] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)key * kv_row_stride);
        vr[it] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)key * kv_row_stride);
      }
    }
    float p[QPG][ITS];
#pragma unroll
    for (int h = 0; h < QPG; ++h) {
      float tm = -INFINITY;
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        float sv = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sv = dot2_bf16(qp[h][j], kr[it][j], sv);
        sv = row16_allreduce_sum(sv) * scale_log2;
        sv = (key0 + it * 4 + kq < key_end) ? sv : -INFINITY;
        p[h][it] = sv;
        tm = fmaxf(tm, sv);
      }
      tm = fmaxf(tm, __shfl_xor(tm, 16));
      tm = fmaxf(tm, __shfl_xor(tm, 32));                  // finite: key0 < key_end
      const float mnew = fmaxf(mrun[h], tm);
      const float alpha = __builtin_amdgcn_exp2f(mrun[h] - mnew);        // exp2(-inf) = 0 on the first tile
      mrun[h] = mnew;
      float ls = 0.f;
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        p[h][it] = __builtin_amdgcn_exp2f(p[h][it] - mnew);
        ls += p[h][it];
      }
      lrun[h] = lrun[h] * alpha + ls;                      // this row-group's keys only; summed over kq at the end
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[h][i] *= alpha;
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      float vf[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { vf[2 * i] = bf16lo_to_f32(vr[it][i]); vf[2 * i + 1] = bf16hi_to_f32(vr[it][i]); }
#pragma unroll
      for (int h = 0; h < QPG; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[h][i] += p[h][it] * vf[i];
    }
  }
  // wave result: sum the 4 row-groups (lanes sub, sub+16, sub+32, sub+48 hold the same dims)
#pragma unroll
  for (int h = 0; h < QPG; ++h) {
    float l = lrun[h];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (lane == 0) { wm[wave][h] = mrun[h]; wl[wave][h] = l; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = acc[h][i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) wo[wave][h][sub * 8 + i] = v;
    }
  }
  __syncthreads();
  if (t < D) {
#pragma unroll
    for (int h = 0; h < QPG; ++h) {
      const float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));   // finite: wave 0 had a tile
      float L = 0.f, O = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float f = (wm[w][h] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(wm[w][h] - M);
        L += wl[w][h] * f;
        O += wo[w][h][t] * f;
      }
      po[((int64_t)blk * H + g * QPG + h) * D + t] = O;
      if (t == 0) { pm[(int64_t)blk * H + g * QPG + h] = M; pl[(int64_t)blk * H + g * QPG + h] = L; }
    }
  }
}

// ---- stage 2: merge nparts partials of one head; final -> bf16 context, else merged partial -----
__global__ __launch_bounds__(128) void decode_attn_merge_kernel(const float* __restrict__ pm,
                                                                const float* __restrict__ pl,
                                                                const float* __restrict__ po, int nparts,
                                                                int64_t sml, int64_t so,
                                                                float* __restrict__ om, float* __restrict__ ol,
                                                                float* __restrict__ oo,
                                                                bf16_t* __restrict__ out) {
  constexpr int D = 128;
  const int head = blockIdx.x, d = threadIdx.x;
  float M = -INFINITY;
  for (int p = 0; p < nparts; ++p) M = fmaxf(M, pm[p * sml + head]);
  float L = 0.f, O = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float m = pm[p * sml + head];
    const float w = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
    L += pl[p * sml + head] * w;
    O += po[p * so + head * D + d] * w;
  }
  if (out) {
    out[head * D + d] = f32_to_bf16(L > 0.f ? O / L : 0.f);
  } else {
    if (d == 0) { om[head] = M; ol[head] = L; }
    oo[head * D + d] = O;
  }
}

}  // namespace

extern "C" int vita_gemv_bf16(const void* x, const void* W, int64_t ldw, void* y, int64_t N, int64_t K,
                              int epilogue, const void* bias, const void* R, void* stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if ((K & 7) || (ldw & 7)) return VITA_ERR_UNSUPPORTED;
  if ((epilogue == VITA_EPI_BIAS && !bias) || (epilogue == VITA_EPI_RESIDUAL && !R)) return VITA_ERR_INVALID_ARG;
  dim3 grid((unsigned)((N + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define VITA_GEMV(E)                                                                                  \
  hipLaunchKernelGGL(gemv_kernel<E>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)W, ldw,     \
                     (bf16_t*)y, N, K, (const bf16_t*)bias, (const bf16_t*)R)
  switch (epilogue) {
    case VITA_EPI_NONE: VITA_GEMV(VITA_EPI_NONE); break;
    case VITA_EPI_BIAS: VITA_GEMV(VITA_EPI_BIAS); break;
    case VITA_EPI_RESIDUAL: VITA_GEMV(VITA_EPI_RESIDUAL); break;
    case VITA_EPI_SWIGLU: VITA_GEMV(VITA_EPI_SWIGLU); break;
    default: return VITA_ERR_UNSUPPORTED;
  }
#undef VITA_GEMV
  return vita_check_launch();
}

extern "C" int vita_decode_attn_partial(const void* q, int64_t q_group_stride, int64_t q_head_stride,
                                        const void* k_cache, const void* v_cache, int64_t kv_row_stride,
                                        int64_t kv_group_stride, int len, const void* len_dev, int n_splits,
                                        int groups, int qpg, int head_dim, float softmax_scale, void* part_m,
                                        void* part_l, void* part_o, void* stream) {
  if (!q || !k_cache || !v_cache || !part_m || !part_l || !part_o || len <= 0 || groups <= 0 || n_splits <= 0)
    return VITA_ERR_INVALID_ARG;
  if (head_dim != 128 || qpg < 1 || qpg > 8 || (q_group_stride & 7) || (q_head_stride & 7) ||
      (kv_row_stride & 7) || (kv_group_stride & 7))
    return VITA_ERR_UNSUPPORTED;
  dim3 grid((unsigned)n_splits, (unsigned)groups), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = softmax_scale * 1.4426950408889634f;
#define VITA_DEC(Q)                                                                                   \
  case Q:                                                                                             \
    hipLaunchKernelGGL(decode_attn_partial_kernel<Q>, grid, block, 0, st, (const bf16_t*)q,           \
                       q_group_stride, q_head_stride, (const bf16_t*)k_cache, (const bf16_t*)v_cache, \
                       kv_row_stride, kv_group_stride, len, (const int*)len_dev, sl2, (float*)part_m, (float*)part_l,      \
                       (float*)part_o);                                                               \
    break;
  switch (qpg) {
    VITA_DEC(1) VITA_DEC(2) VITA_DEC(3) VITA_DEC(4) VITA_DEC(5) VITA_DEC(6) VITA_DEC(7) VITA_DEC(8)
  }
#undef VITA_DEC
  return vita_check_launch();
}

extern "C" int vita_decode_attn_merge(const void* part_m, const void* part_l, const void* part_o, int nparts,
                                      int64_t part_ml_stride, int64_t part_o_stride, int heads, int head_dim,
                                      void* out_m, void* out_l, void* out_o,
                                      void* out_bf16, void* stream) {
  if (nparts < 0 || heads <= 0) return VITA_ERR_INVALID_ARG;
  if (nparts > 0 && (!part_m || !part_l || !part_o)) return VITA_ERR_INVALID_ARG;
  if (!out_bf16 && (!out_m || !out_l || !out_o)) return VITA_ERR_INVALID_ARG;
  if (head_dim != 128) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_attn_merge_kernel, dim3((unsigned)heads), dim3(128), 0, (hipStream_t)stream,
                     (const float*)part_m, (const float*)part_l, (const float*)part_o, nparts, part_ml_stride, part_o_stride,
                     (float*)out_m,
                     (float*)out_l, (float*)out_o, (bf16_t*)out_bf16);
  return vita_check_launch();
}
