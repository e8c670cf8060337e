// Stand-alone candidate for the next flash-attention forward (NOT part of libvita_hip.so; written at the end of round 1 after
// attn_ladder.hip put the ceiling of the shipped structure — 8 waves x 32 query rows, 32x32x16 MFMAs — at 1.30 PFLOP/s).
// Plain causal GQA attention, d = 128:  Q [S][Hq][128], K / V [S][Hkv][128] bf16 -> O [S][Hq][128] bf16, S % 256 == 0.
// The two levers in one kernel: v_mfma_f32_16x16x32_bf16 and 64 query rows per wave (every K / V fragment feeds 4 MFMAs).
//   * one workgroup = 4 waves (one per SIMD, up to 512 registers) = 256 query rows of one query head; 64-key tiles
//   * S^T = K Q^T in 16 x 16 blocks: s[kb][qb], a lane holds S^T[key 16kb + 4g + r][query 16qb + c], g = lane >> 4, c = lane & 15;
//     the Q fragments (B operand, 64 registers) are loaded once, straight from global memory
//   * softmax in registers: row maximum = in-lane over 16 values, then v_permlane16_swap + v_permlane32_swap across the four lanes
//     that share a query; the row sum stays a per-lane partial until the end; exp2 with the scale folded in
//   * O^T = V^T P^T: the B operand is the packed S^T accumulator pair (blocks 2k, 2k+1 of a 32-key step: k-slot (g, i) <-> key
//     16(2k) + 4g + i for i < 4 and 16(2k+1) + 4g + i - 4 for i >= 4); the A operand gathers exactly those keys from the row-major V
//     tile with two ds_read_b64_tr_b16 (16-lane group g reads the [4 keys][16 d] block of keys 4g..4g+3)
//   * K / V tiles HBM/L2 -> VGPR -> LDS, two stages, one barrier per tile; K layout = the library's (slot ^ (row & 15)), V layout
//     row * 256 + ((chunk ^ (((row & 3) << 1) | ((row >> 2) & 1))) << 5): conflict-free for the 16-row fragment reads
// First version: program order, no hand scheduling.  It checks itself against a naive kernel and times S = 16K and 128K.
//   hipcc --offload-arch=gfx950 -O3 -fno-honor-nans attn16.hip -o attn16 && ./attn16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef unsigned short bf16_t;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int D = 128, KVT = 64, ROWB = D * 2, TILEB = KVT * ROWB, SLOTB = 2 * TILEB;    // 16 KiB K | 16 KiB V per stage

__device__ __forceinline__ int k_off(int row, int slot) { return row * ROWB + ((slot ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk, int b) {
  return row * ROWB + ((chunk ^ (((row & 3) << 1) | ((row >> 2) & 1))) << 5) + b;
}
__device__ __forceinline__ unsigned pack2(float a, float b) {         // RNE fp32 -> bf16 pair
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ float group_max(float x) {                 // over the four lanes l, l^16, l^32, l^48
  unsigned xi = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  xi = __float_as_uint(x);
  auto b = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float x) {
  unsigned xi = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  xi = __float_as_uint(x);
  auto b = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ASM = false: compiler builtins (hipcc chooses the register classes: with > 256 live registers every MFMA result lands in AGPRs
//   and is shuffled through v_accvgpr moves — the correctness baseline).
// ASM = true : the same MFMAs as inline asm with explicit classes — S^T accumulators in VGPRs ("+v": the softmax reads them), O^T
//   accumulators in AGPRs ("+a": with the lazy rescale below they are touched by MFMAs only) — and s_nop padding where a VALU
//   reads an MFMA result, because the hazard recognizer does not look into inline asm.  NOT yet run: validate against ASM = false.
template <bool ACC_AGPR, bool ASM>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8 a, const bf16x8 b) {
  if (ASM) {
    if (ACC_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
}

template <bool ASM>
__global__ __launch_bounds__(256, 1) void attn16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                        const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int S, int Hq,
                                                        int Hkv, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = S / 256;
  const int head = blockIdx.x % Hq, qt = nq - 1 - blockIdx.x / Hq;          // heaviest query tiles first
  const int kvh = head / (Hq / Hkv);
  const int q0 = qt * 256, qw = q0 + wave * 64;                              // first query row of the workgroup / of this wave
  const int my_diag = qw / KVT;                                              // the one tile this wave has to mask
  const int n_tiles = q0 / KVT + 4;                                          // tiles 0 .. (q0 + 255) / 64

  // Q fragments (B operand of S^T): query 16qb + c, d = 32ks + 8g .. + 7
  bf16x8 qf[4][4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[qb][ks] = *(const bf16x8*)(Q + ((int64_t)(qw + 16 * qb + c) * Hq + head) * D + 32 * ks + 8 * g);

  // staging: thread -> 4 pieces of K and 4 of V per tile: tile row tid / 16 + 16 i, 16-byte slot tid & 15
  const int srow = tid >> 4, sslot = tid & 15;
  const bf16_t* kp = K + ((int64_t)srow * Hkv + kvh) * D + sslot * 8;
  const bf16_t* vp = V + ((int64_t)srow * Hkv + kvh) * D + sslot * 8;
  const int64_t row16 = (int64_t)16 * Hkv * D, tile_stride = (int64_t)KVT * Hkv * D;
  unsigned kst[4], vst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    kst[i] = k_off(srow + 16 * i, sslot);
    vst[i] = TILEB + v_off(srow + 16 * i, sslot >> 1, (sslot & 1) << 4);
  }
  u32x4 kg[4], vg[4];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kg[i] = *(const u32x4*)(kp + t * tile_stride + i * row16);
      vg[i] = *(const u32x4*)(vp + t * tile_stride + i * row16);
    }
  };
  auto store_tile = [&](unsigned stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(lds_u32x4*)(uintptr_t)(stage + kst[i]) = kg[i];
      *(lds_u32x4*)(uintptr_t)(stage + vst[i]) = vg[i];
    }
  };
  // fragment read offsets
  unsigned kfo[4], vfo[8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kfo[ks] = k_off(c, 4 * ks + g);           // key row c (+16 kb), d = 32ks + 8g: the swizzle only
                                                                            // sees row & 15, so + 16 kb rows is an immediate
#pragma unroll
  for (int db = 0; db < 8; ++db)                                            // 16-lane group g, lane c: key 4g + (c >> 2), d = 16db + 4(c & 3)
    vfo[db] = TILEB + v_off(4 * g + (c >> 2), db, 8 * (c & 3));            // + 16-key steps: (row & 3), (row >> 2) & 1 unchanged -> immediate

  f32x4 o[8][4];                                                            // O^T[d 16db + 4g + r][query 16qb + c]
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) o[db][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) { m_run[qb] = -1.0e30f; l_run[qb] = 0.f; }

  load_tile(0);
  store_tile(lds0);
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    const unsigned cur = lds0 + (t & 1) * SLOTB, nxt = lds0 + ((t + 1) & 1) * SLOTB;
    if (t > my_diag && t + 1 < n_tiles) load_tile(t + 1);                   // (a wave without work on this tile only stages)
    if (t <= my_diag) {                                                     // tiles above this wave's rows are skipped
      // ---- S^T = K Q^T -----------------------------------------------------------------------------------------------
      f32x4 s[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) s[kb][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // fragments are read one step ahead of the four MFMAs that use them; sched_barrier keeps hipcc from hoisting all 16 reads
      auto k_frag = [&](int i) __attribute__((always_inline)) {             // i = 4 ks + kb
        return *(lds_bf16x8*)(uintptr_t)(cur + kfo[i >> 2] + (i & 3) * 16 * ROWB);
      };
      bf16x8 kf = k_frag(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        bf16x8 kn = kf;
        if (i + 1 < 16) kn = k_frag(i + 1);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) mfma16<false, ASM>(s[i & 3][qb], kf, qf[qb][i >> 2]);
        __builtin_amdgcn_sched_barrier(0);
        kf = kn;
      }
      if (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");         // MFMA -> VALU access to s (the mask writes it, the softmax reads it)
      if (t + 1 < n_tiles) load_tile(t + 1);                                // in flight under the softmax and P V
      // ---- causal mask on the diagonal tile: key 64t + 16kb + 4g + r visible to query qw + 16qb + c iff key <= query ------
      if (t == my_diag) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * kb + 4 * g + r > 16 * qb + c) s[kb][qb][r] = -1.0e30f;
      }
      // ---- online softmax (base 2, scale folded in), lazy rescale: the running maximum only moves when some query of this wave
      //      exceeds it by more than 8 / scale_log2e (P <= 2^8 until then), so O^T is normally not touched between the MFMAs -------
      float mx[4];
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        float m = s[0][qb][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kb][qb][r]);
        mx[qb] = group_max(m);
        grow = grow || (mx[qb] * scale_log2e > m_run[qb] * scale_log2e + 8.0f);
      }
      if (__any(grow)) {                                                   // wave-uniform
        if (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // MFMA -> read of o
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const float m_new = fmaxf(m_run[qb], mx[qb]);
          const float alpha = exp2f((m_run[qb] - m_new) * scale_log2e);
          m_run[qb] = m_new;
          l_run[qb] *= alpha;
#pragma unroll
          for (int db = 0; db < 8; ++db) o[db][qb] *= alpha;
        }
      }
      unsigned pk[4][4][2];                                                 // P^T packed: [kb][qb] -> 4 bf16
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        const float mb = m_run[qb] * scale_log2e;
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          float p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = exp2f(s[kb][qb][r] * scale_log2e - mb);
            ls += p[r];
          }
          pk[kb][qb][0] = pack2(p[0], p[1]);
          pk[kb][qb][1] = pack2(p[2], p[3]);
        }
        l_run[qb] += ls;
      }
      auto v_frag = [&](int i) __attribute__((always_inline)) {             // i = 8 kk + db
        const unsigned va = cur + vfo[i & 7] + (32 * (i >> 3)) * ROWB;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va));                  // keys 32kk + 4g ..
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va + 16 * ROWB));      // keys 32kk + 16 + 4g ..
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      bf16x8 vf = v_frag(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = i >> 3, db = i & 7;
        bf16x8 vn = vf;
        if (i + 1 < 16) vn = v_frag(i + 1);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const u32x4 w = (u32x4){pk[2 * kk][qb][0], pk[2 * kk][qb][1], pk[2 * kk + 1][qb][0], pk[2 * kk + 1][qb][1]};
          mfma16<true, ASM>(o[db][qb], vf, __builtin_bit_cast(bf16x8, w));
        }
        __builtin_amdgcn_sched_barrier(0);
        vf = vn;
      }
    }
    if (t + 1 < n_tiles) store_tile(nxt);
    __syncthreads();
  }
  // ---- epilogue: O[query][head][16db + 4g + r] = O^T / l ----------------------------------------------------------------------
  if (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const float inv = 1.0f / group_sum(l_run[qb]);
    bf16_t* op = O + ((int64_t)(qw + 16 * qb + c) * Hq + head) * D + 4 * g;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      const f32x4 v = o[db][qb];
      *(u32x2*)(op + 16 * db) = (u32x2){pack2(v[0] * inv, v[1] * inv), pack2(v[2] * inv, v[3] * inv)};
    }
  }
}

__global__ void naive_attn(const bf16_t* Q, const bf16_t* K, const bf16_t* V, float* out, const int* rows, int nrows, int S, int Hq,
                           int Hkv, float scale) {
  const int ri = blockIdx.x, head = blockIdx.y, d = threadIdx.x;           // one block per (sampled row, head), 128 threads = d
  const int q = rows[ri], kvh = head / (Hq / Hkv);
  __shared__ float red[128];
  auto bf = [](bf16_t h) { return __uint_as_float((unsigned)h << 16); };
  const bf16_t* qp = Q + ((int64_t)q * Hq + head) * D;
  float m = -1e30f, l = 0.f, acc = 0.f;
  for (int k = 0; k <= q; ++k) {
    red[d] = bf(qp[d]) * bf(K[((int64_t)k * Hkv + kvh) * D + d]);
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) { if (d < st) red[d] += red[d + st]; __syncthreads(); }
    const float sc = red[0] * scale;
    __syncthreads();
    const float mn = fmaxf(m, sc), a = expf(m - mn), p = expf(sc - mn);
    l = l * a + p;
    acc = acc * a + p * bf(V[((int64_t)k * Hkv + kvh) * D + d]);
    m = mn;
  }
  out[((int64_t)ri * Hq + head) * D + d] = acc / l;
}

static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f(bf16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <bool ASM>
static int run(int S, int Hq, int Hkv, int nrows_check) {
  const size_t nq = (size_t)S * Hq * D, nkv = (size_t)S * Hkv * D;
  std::vector<bf16_t> hQ(nq), hK(nkv), hV(nkv);
  unsigned s = 777u + S;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& x : hQ) x = f2bf(rnd() * 2.f);
  for (auto& x : hK) x = f2bf(rnd() * 2.f);
  for (auto& x : hV) x = f2bf(rnd() * 2.f);
  bf16_t *dQ, *dK, *dV, *dO; float* dRef; int* dRows;
  (void)hipMalloc(&dQ, nq * 2); (void)hipMalloc(&dK, nkv * 2); (void)hipMalloc(&dV, nkv * 2); (void)hipMalloc(&dO, nq * 2);
  (void)hipMemcpy(dQ, hQ.data(), nq * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dK, hK.data(), nkv * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dV, hV.data(), nkv * 2, hipMemcpyHostToDevice);
  (void)hipMemset(dO, 0xff, nq * 2);
  const float scale = 1.0f / sqrtf((float)D);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_kernel<ASM>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOTB);
  const int grid = (S / 256) * Hq;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f, ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(attn16_kernel<ASM>, dim3(grid), dim3(256), 2 * SLOTB, 0, dQ, dK, dV, dO, S, Hq, Hkv, scale * 1.4426950408889634f);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  std::vector<int> rows;
  for (int i = 0; i < nrows_check; ++i) rows.push_back((int)(((int64_t)i * 2654435761u) % (S < 4096 ? S : 4096)));   // early rows: cheap to check
  rows[0] = 0; rows[1] = 63; rows[2] = 64; rows[3] = 255; rows[4] = 256; rows[5] = (S < 4096 ? S : 4096) - 1;
  const int nr = (int)rows.size();
  (void)hipMalloc(&dRef, (size_t)nr * Hq * D * 4); (void)hipMalloc(&dRows, nr * 4);
  (void)hipMemcpy(dRows, rows.data(), nr * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(naive_attn, dim3(nr, Hq), dim3(128), 0, 0, dQ, dK, dV, dRef, dRows, nr, S, Hq, Hkv, scale);
  std::vector<float> ref((size_t)nr * Hq * D); std::vector<bf16_t> hO(nq);
  (void)hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hO.data(), dO, nq * 2, hipMemcpyDeviceToHost);
  double max_err = 0; long bad = 0;
  for (int i = 0; i < nr; ++i)
    for (int h = 0; h < Hq; ++h)
      for (int d = 0; d < D; ++d) {
        const float want = ref[((size_t)i * Hq + h) * D + d], got = bf2f(hO[((size_t)rows[i] * Hq + h) * D + d]);
        const double err = fabs(got - want);
        if (err > max_err) max_err = err;
        if (!(err < 2e-2)) ++bad;                                          // |V| <= 1, P rounded to bf16: abs error ~ 4e-3
      }
  const double pairs = (double)S * (S + 1) / 2;
  printf("%s S=%6d Hq=%d Hkv=%d  %9.3f ms  %7.1f TFLOP/s   checked %d rows x %d heads: max abs err %.2e, %ld bad\n", ASM ? "asm     " : "builtins", S, Hq, Hkv, best,
         4.0 * D * Hq * pairs / (best * 1e-3) / 1e12, nr, Hq, max_err, bad);
  (void)hipFree(dQ); (void)hipFree(dK); (void)hipFree(dV); (void)hipFree(dO); (void)hipFree(dRef); (void)hipFree(dRows);
  return bad != 0;
}

int main() {
  int rc = run<false>(256, 5, 1, 64);                 // one query tile: every structure once
  rc |= run<false>(1024, 10, 2, 96);
  rc |= run<false>(16384, 40, 8, 48);
  rc |= run<true>(256, 5, 1, 64);
  rc |= run<true>(1024, 10, 2, 96);
  rc |= run<true>(16384, 40, 8, 48);
  rc |= run<true>(131072, 40, 8, 16);
  printf(rc ? "FAILED\n" : "all checks passed\n");
  return rc;
}
