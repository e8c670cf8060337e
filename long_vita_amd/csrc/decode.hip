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

constexpr int kGemvRows = 1;

// ---- y[N] = epilogue(W[N,K] . norm(x)[K]) : one wave per output, 16-byte pieces of the W row per lane ----
// NORM: x is RMS-normalised on the fly, bf16(bf16(x * rstd) * gamma) as vita_rmsnorm_fwd rounds it; every wave
// recomputes the 10 KB sum of squares (L2-resident) instead of a separate launch + round trip.
template <int EPI, bool NORM>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                   int64_t ldw, bf16_t* __restrict__ y, int64_t N, int64_t K,
                                                   const bf16_t* __restrict__ bias, const bf16_t* __restrict__ R,
                                                   const bf16_t* __restrict__ gamma, float eps) {
  constexpr int ROWS = kGemvRows;                  // outputs per wave: x is read (and normalised) once for all of them
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * ROWS;
  if (n0 >= N) return;
  const int nvec = (int)(K >> 3);
  const u32x4* xr = reinterpret_cast<const u32x4*>(x);
  const u32x4* gr = reinterpret_cast<const u32x4*>(gamma);
  float rstd = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int v = lane; v < nvec; v += 64) ss += dot8(xr[v], xr[v]);
    rstd = rsqrtf(wave_reduce_sum(ss) / (float)K + eps);
  }
  const u32x4* w0[ROWS];
  const u32x4* w1[ROWS];
  float a0[ROWS], a1[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int64_t n = (n0 + r < N) ? n0 + r : N - 1;                          // clamp: tail rows recompute row N-1
    w0[r] = reinterpret_cast<const u32x4*>(W + n * ldw);
    w1[r] = reinterpret_cast<const u32x4*>(W + (n + N) * ldw);                // SWIGLU: the "up" row
    a0[r] = 0.f; a1[r] = 0.f;
  }
#pragma unroll 4
  for (int v = lane; v < nvec; v += 64) {
    u32x4 xv = xr[v];
    if (NORM) {
      const u32x4 gv = gr[v];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xv[j] = pack_bf16x2(bf16_round(bf16lo_to_f32(xv[j]) * rstd) * bf16lo_to_f32(gv[j]),
                            bf16_round(bf16hi_to_f32(xv[j]) * rstd) * bf16hi_to_f32(gv[j]));
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      a0[r] += dot8(xv, w0[r][v]);
      if (EPI == VITA_EPI_SWIGLU) a1[r] += dot8(xv, w1[r][v]);
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float s0 = wave_reduce_sum(a0[r]);
    const float s1 = (EPI == VITA_EPI_SWIGLU) ? wave_reduce_sum(a1[r]) : 0.f;
    const int64_t n = n0 + r;
    if (lane != 0 || n >= N) continue;
    float o = s0;
    if (EPI == VITA_EPI_BIAS) o += bf16_to_f32(bias[n]);
    if (EPI == VITA_EPI_RESIDUAL) o = bf16_round(o) + bf16_to_f32(R[n]);
    if (EPI == VITA_EPI_SWIGLU) {                     // same rounding chain as the GEMM epilogue
      const float g = bf16_round(s0), u = bf16_round(s1);
      o = bf16_round(g / (1.0f + __expf(-g))) * u;
    }
    y[n] = f32_to_bf16(o);
  }
}

template <int EPI, bool NORM>
void launch_gemv(const void* x, const void* W, int64_t ldw, void* y, int64_t N, int64_t K, const void* bias,
                 const void* R, const void* gamma, float eps, hipStream_t st) {
  const int64_t waves = (N + kGemvRows - 1) / kGemvRows;
  hipLaunchKernelGGL((gemv_kernel<EPI, NORM>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x,
                     (const bf16_t*)W, ldw, (bf16_t*)y, N, K, (const bf16_t*)bias, (const bf16_t*)R,
                     (const bf16_t*)gamma, eps);
}

// ---- RoPE of the new token's q and k heads (in place in the mixed qkv row) + append of its K / V to the cache ----
__global__ __launch_bounds__(256) void rope_append_kernel(bf16_t* __restrict__ mixed, int groups, int qpg, int head_dim,
                                                          const bf16_t* __restrict__ cos_tab,
                                                          const bf16_t* __restrict__ sin_tab, bf16_t* __restrict__ kc,
                                                          bf16_t* __restrict__ vc, int64_t kv_row_stride,
                                                          int64_t kv_group_stride, int append_row) {
  const int half = head_dim >> 1, nv = half >> 3, hpg = qpg + 2;
  const int total = groups * hpg * nv;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int vi = i % nv, h = (i / nv) % hpg, g = i / (nv * hpg);
    bf16_t* p = mixed + ((int64_t)g * hpg + h) * head_dim + vi * 8;
    u32x4 x1 = *reinterpret_cast<const u32x4*>(p);
    u32x4 x2 = *reinterpret_cast<const u32x4*>(p + half);
    if (h <= qpg) {
      const u32x4 c = *reinterpret_cast<const u32x4*>(cos_tab + vi * 8);
      const u32x4 s = *reinterpret_cast<const u32x4*>(sin_tab + vi * 8);
      rope_rotate8(x1, x2, c, s, 1.0f);
      *reinterpret_cast<u32x4*>(p) = x1;
      *reinterpret_cast<u32x4*>(p + half) = x2;
    }
    if (h >= qpg && append_row >= 0) {
      bf16_t* o = (h == qpg ? kc : vc) + (int64_t)append_row * kv_row_stride + (int64_t)g * kv_group_stride + vi * 8;
      *reinterpret_cast<u32x4*>(o) = x1;
      *reinterpret_cast<u32x4*>(o + half) = x2;
    }
  }
}

// ---- decode attention, stage 1 ---------------------------------------------------------------
// One workgroup = a contiguous range of cached keys of one kv group; its 4 waves run independently
// (no barrier, no LDS in the loop) over interleaved 32-key (16 for qpg >= 7) tiles with their own online softmax:
//   lane = (kq = lane/16, sub = lane%16): the 16 lanes of a row-group read one 256-byte K/V row per
//   instruction (4 rows per wave instruction), 8 K and 8 V row loads in flight per lane;
//   q.k = 4 x v_dot2_f32_bf16 per head and row + a 4-step DPP rotate-add that leaves the score in all
//   16 lanes — exactly the lanes that need p for their 8 output dims in P.V, so scores never leave
//   registers.  The waves' (m, l, O) are merged through LDS once at the end.
constexpr int kDecKeys = 128;        // key granularity of a split (host side: ops.DECODE_KEYS_PER_TILE)

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
// pass 1: threads over parts -> block max M, weights w_p = exp2(m_p - M) in LDS, L = sum l_p w_p;
// pass 2: thread d accumulates sum_p w_p O_p[d] with independent loads (the serial version was latency-bound).
constexpr int kMergeMaxParts = 1024;
__global__ __launch_bounds__(128) void decode_attn_merge_kernel(const float* __restrict__ pm,
                                                                const float* __restrict__ pl,
                                                                const float* __restrict__ po, int nparts,
                                                                int64_t sml, int64_t so,
                                                                float* __restrict__ om, float* __restrict__ ol,
                                                                float* __restrict__ oo,
                                                                bf16_t* __restrict__ out) {
  constexpr int D = 128;
  __shared__ float w[kMergeMaxParts];
  __shared__ float red[4];
  const int head = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float m = -INFINITY;
  for (int p = t; p < nparts; p += 128) m = fmaxf(m, pm[p * sml + head]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  const float M = fmaxf(red[0], red[1]);
  float l = 0.f;
  for (int p = t; p < nparts; p += 128) {
    const float mp = pm[p * sml + head];
    const float wp = (mp == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mp - M);
    w[p] = wp;
    l += pl[p * sml + head] * wp;
  }
  l = wave_reduce_sum(l);
  if (lane == 0) red[2 + wave] = l;
  __syncthreads();
  const float L = red[2] + red[3];
  float O0 = 0.f, O1 = 0.f, O2 = 0.f, O3 = 0.f;
  const float* src = po + head * D + t;
  int p = 0;
  for (; p + 4 <= nparts; p += 4) {
    O0 += src[(p + 0) * so] * w[p + 0];
    O1 += src[(p + 1) * so] * w[p + 1];
    O2 += src[(p + 2) * so] * w[p + 2];
    O3 += src[(p + 3) * so] * w[p + 3];
  }
  for (; p < nparts; ++p) O0 += src[p * so] * w[p];
  const float O = (O0 + O1) + (O2 + O3);
  if (out) {
    out[head * D + t] = f32_to_bf16(L > 0.f ? O / L : 0.f);
  } else {
    if (t == 0) { om[head] = M; ol[head] = L; }
    oo[head * D + t] = O;
  }
}

}  // namespace

extern "C" int vita_gemv_bf16(const void* x, const void* W, int64_t ldw, void* y, int64_t N, int64_t K,
                              int epilogue, const void* bias, const void* R, void* stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if ((K & 7) || (ldw & 7)) return VITA_ERR_UNSUPPORTED;
  if ((epilogue == VITA_EPI_BIAS && !bias) || (epilogue == VITA_EPI_RESIDUAL && !R)) return VITA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VITA_EPI_NONE: launch_gemv<VITA_EPI_NONE, false>(x, W, ldw, y, N, K, bias, R, nullptr, 0.f, st); break;
    case VITA_EPI_BIAS: launch_gemv<VITA_EPI_BIAS, false>(x, W, ldw, y, N, K, bias, R, nullptr, 0.f, st); break;
    case VITA_EPI_RESIDUAL: launch_gemv<VITA_EPI_RESIDUAL, false>(x, W, ldw, y, N, K, bias, R, nullptr, 0.f, st); break;
    case VITA_EPI_SWIGLU: launch_gemv<VITA_EPI_SWIGLU, false>(x, W, ldw, y, N, K, bias, R, nullptr, 0.f, st); break;
    default: return VITA_ERR_UNSUPPORTED;
  }
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
  if (nparts > kMergeMaxParts) return VITA_ERR_UNSUPPORTED;
  if (nparts > 0 && (!part_m || !part_l || !part_o)) return VITA_ERR_INVALID_ARG;
  if (!out_bf16 && (!out_m || !out_l || !out_o)) return VITA_ERR_INVALID_ARG;
  if (head_dim != 128) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_attn_merge_kernel, dim3((unsigned)heads), dim3(128), 0, (hipStream_t)stream,
                     (const float*)part_m, (const float*)part_l, (const float*)part_o, nparts, part_ml_stride, part_o_stride,
                     (float*)out_m,
                     (float*)out_l, (float*)out_o, (bf16_t*)out_bf16);
  return vita_check_launch();
}

// ---- one decoder layer for one token, launched from C (7 kernels instead of ~13 launches from Python) ----------
namespace {
int launch_decode_partial(const vita_decode_layer_params* p, hipStream_t st) {
  const int qpg = p->heads / p->kv_groups, hpg = qpg + 2;
  const bf16_t* q = (const bf16_t*)p->qkv;
  return vita_decode_attn_partial(q, (int64_t)hpg * p->head_dim, p->head_dim, p->k_cache, p->v_cache, p->kv_row_stride,
                                  p->kv_group_stride, p->len, nullptr, p->n_splits, p->kv_groups, qpg, p->head_dim,
                                  p->softmax_scale, p->part_m, p->part_l, p->part_o, st);
}
bool decode_layer_args_ok(const vita_decode_layer_params* p) {
  if (!p || !p->h || !p->qkv || !p->ctx || !p->act) return false;
  if (p->hidden <= 0 || p->heads <= 0 || p->kv_groups <= 0 || p->heads % p->kv_groups || p->ffn <= 0) return false;
  return true;
}
}  // namespace

extern "C" int vita_decode_layer_attn(const vita_decode_layer_params* p, void* stream) {
  if (!decode_layer_args_ok(p) || !p->ln1 || !p->qkv_w || !p->qkv_b || !p->cos || !p->sin || !p->k_cache || !p->v_cache ||
      !p->part_m || !p->part_l || !p->part_o)
    return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 128 || (p->hidden & 7) || p->len < 0 || p->len > p->capacity || p->append_row >= p->capacity ||
      p->n_splits <= 0)
    return VITA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int qpg = p->heads / p->kv_groups;
  const int64_t qkv_out = (int64_t)(p->heads + 2 * p->kv_groups) * p->head_dim;
  launch_gemv<VITA_EPI_BIAS, true>(p->h, p->qkv_w, p->hidden, p->qkv, qkv_out, p->hidden, p->qkv_b, nullptr, p->ln1, p->eps, st);
  hipLaunchKernelGGL(rope_append_kernel, dim3(4), dim3(256), 0, st, (bf16_t*)p->qkv, p->kv_groups, qpg, p->head_dim,
                     (const bf16_t*)p->cos, (const bf16_t*)p->sin, (bf16_t*)p->k_cache, (bf16_t*)p->v_cache,
                     p->kv_row_stride, p->kv_group_stride, p->append_row);
  const int H = p->heads, D = p->head_dim;
  if (p->len > 0) {
    const int rc = launch_decode_partial(p, st);
    if (rc != VITA_OK) return rc;
  }
  const int nparts = p->len > 0 ? p->n_splits : 0;
  if (p->msg) {                       // CP > 1: packed (o, m, l) partial for the all-gather
    float* msg = (float*)p->msg;
    return vita_decode_attn_merge(p->part_m, p->part_l, p->part_o, nparts, H, (int64_t)H * D, H, D, msg + (int64_t)H * D,
                                  msg + (int64_t)H * D + H, msg, nullptr, st);
  }
  return vita_decode_attn_merge(p->part_m, p->part_l, p->part_o, nparts, H, (int64_t)H * D, H, D, nullptr, nullptr, nullptr,
                                p->ctx, st);
}

extern "C" int vita_decode_layer_mlp(const vita_decode_layer_params* p, void* stream) {
  if (!decode_layer_args_ok(p) || !p->o_w || !p->ln2 || !p->fc1_w || !p->fc2_w) return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 128 || (p->hidden & 7) || (p->ffn & 7)) return VITA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int H = p->heads, D = p->head_dim;
  if (p->gathered) {                  // CP > 1: merge the ranks' partials into the context
    if (p->n_ranks <= 0) return VITA_ERR_INVALID_ARG;
    const float* g = (const float*)p->gathered;
    const int64_t msg_len = (int64_t)H * D + 2 * H;
    const int rc = vita_decode_attn_merge(g + (int64_t)H * D, g + (int64_t)H * D + H, g, p->n_ranks, msg_len, msg_len, H, D,
                                          nullptr, nullptr, nullptr, p->ctx, st);
    if (rc != VITA_OK) return rc;
  }
  const int64_t hd = (int64_t)H * D;
  launch_gemv<VITA_EPI_RESIDUAL, false>(p->ctx, p->o_w, hd, p->h, p->hidden, hd, nullptr, p->h, nullptr, 0.f, st);
  launch_gemv<VITA_EPI_SWIGLU, true>(p->h, p->fc1_w, p->hidden, p->act, p->ffn, p->hidden, nullptr, nullptr, p->ln2, p->eps, st);
  launch_gemv<VITA_EPI_RESIDUAL, false>(p->act, p->fc2_w, p->ffn, p->h, p->hidden, p->ffn, nullptr, p->h, nullptr, 0.f, st);
  return vita_check_launch();
}
