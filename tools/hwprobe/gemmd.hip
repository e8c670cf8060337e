// Stand-alone candidate for the GEMM main loop, round 2 (NOT part of libvita_hip.so).
//   C[M,N] = A[M,K] @ W[N,K]^T, bf16 in, fp32 accumulate, bf16 out;  M, N multiples of 256, K a multiple of 64.
// What is different from gemm.hip's kernels and from gemm16.hip:
//   * 4 waves x (128 x 128) with v_mfma_f32_16x16x32_bf16, the 64 accumulator blocks (256 registers) pinned in AGPRs by
//     inline-asm MFMAs ("+a"), the 32 operand fragments of a K tile in VGPRs: nothing moves between the two halves of the file;
//   * operands go HBM/L2 -> LDS with the LDS-DMA (buffer_load_dwordx4 ... lds through a buffer descriptor: one per-lane offset
//     register for all 16 pieces of a wave, SGPR row offsets);
//   * LDS layout "interleaved rows, padded lines": line (h, r16) = the eight tile rows 128 h + 16 rb + r16 (rb = 0..7), 128 B
//     each, + 16 B of padding.  One DMA instruction fills exactly one line (lane -> rb = lane / 8, 16-B chunk lane % 8: the
//     global side reads whole 128-B row segments), and the 16 x 16 x 32 fragment read (lane -> row lane % 16, k chunk lane / 16)
//     strides 1040 B = 260 dwords between lanes: 4 r mod 64 banks, conflict-free, with row block and k half as immediates;
//   * two stages, prefetch distance two: the DMA of tile t+2 goes into the stage tile t is being read from, as soon as every
//     wave has its second-half fragments in registers (barrier in the first half of the iteration), so a piece has about one and a
//     half iterations to land; the fragments of tile t+1's first half are read at the end of iteration t;
//   * the whole loop is a fixed instruction order (asm volatile statements + sched_barrier): MFMA slots 0..127, a fragment read
//     behind every second MFMA of slots 0..30, lgkmcnt(0) + s_barrier at 36, a DMA piece every fourth slot 40..100,
//     vmcnt + s_barrier at 103, the next tile's 16 first-half reads behind slots 104..119;
//   * optional staggered K start by XCD (STAG).
// It checks itself against a naive kernel and times the decoder shapes.   hipcc --offload-arch=gfx950 -O3 gemmd.hip -o gemmd
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
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int BK = 64, BM = 256, BN = 256;
constexpr int LINE = 1040, HALF = 16 * LINE, OPB = 2 * HALF, STAGE = 2 * OPB, LDS_BYTES = 2 * STAGE;   // 133120 B

__device__ __forceinline__ unsigned pack2(float a, float b) {         // RNE fp32 -> bf16 pairs
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

template <int STAG, int RGAP = 2, int BAR1 = 36, int DMA0 = 40, int DGAP = 4, int NX0 = 104, int NGAP = 1, int GM = 4>
__global__ __launch_bounds__(256, 1) void gemmd_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                       bf16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware grouped tile order (as gemm.hip)
  const int tiles_m = M / BM, tiles_n = N / BN, nwg = tiles_m * tiles_n;
  int pid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = GM;
  const int per_group = GROUP_M * tiles_n, group = pid / per_group, first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M), in_group = pid - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int nk = K / BK;

  // ---- DMA geometry: wave w fills lines (h = w >> 1, r16 = (w & 1) * 8 + i), i = 0..7, of both operands ----------------------
  const unsigned voff = (unsigned)(((lane >> 3) * 16 * K + (lane & 7) * 8) * 2);            // bytes; the only per-lane part
  const int h = wave >> 1, r0 = (wave & 1) * 8;
  const unsigned s_row0 = (unsigned)((128 * h + r0) * K * 2), s_step = (unsigned)(K * 2);  // SGPR row offsets
  const unsigned d_line0 = (unsigned)(h * HALF + r0 * LINE);
  const char* ap = (const char*)(A + m0 * K);
  const char* wp = (const char*)(W + n0 * K);
  int koff = 0;                                                                             // byte offset of the K tile fetched next
  if (STAG == 1) koff = (int)((((int64_t)(blockIdx.x & 7) * nk) >> 3) * BK * 2);
  koff = __builtin_amdgcn_readfirstlane(koff);
  auto dma_piece = [&](unsigned stage, int j) __attribute__((always_inline)) {              // j = 0..7: A lines, 8..15: W lines
    const char* base = (j < 8 ? ap : wp) + koff;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const int i = j & 7;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lvoid*)(uintptr_t)(stage + (j < 8 ? 0 : OPB) + d_line0 + i * LINE), 16, voff,
                                             s_row0 + i * s_step, 0, 0);
  };
  auto next_tile = [&]() __attribute__((always_inline)) { koff += BK * 2; if (koff == K * 2) koff = 0; };

  // ---- fragment reads: lane -> (row lane % 16) * LINE + (k chunk lane / 16) * 16, + row block * 128 + k half * 64 ------------------
  const unsigned rd_a = (unsigned)(wm * HALF + (lane & 15) * LINE + (lane >> 4) * 16);
  const unsigned rd_w = (unsigned)(OPB + wn * HALF + (lane & 15) * LINE + (lane >> 4) * 16);
  bf16x8 af[2][8], wf[2][8];
  // read number q (0..15) of k half ks: W block 0 first, then the eight A blocks, then W blocks 1..7 (the order the MFMAs need them)
  auto frag_read = [&](unsigned stage, int ks, int q) __attribute__((always_inline)) {
    if (q == 0 || q > 8) {
      const int nb = q == 0 ? 0 : q - 8;
      const unsigned ad = stage + rd_w + nb * 128 + ks * 64;
      asm volatile("ds_read_b128 %0, %1" : "=v"(wf[ks][nb]) : "v"(ad));
    } else {
      const int mb = q - 1;
      const unsigned ad = stage + rd_a + mb * 128 + ks * 64;
      asm volatile("ds_read_b128 %0, %1" : "=v"(af[ks][mb]) : "v"(ad));
    }
  };

  f32x4 acc[8][8];                                                       // [n block][m block], AGPRs
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tiles 0 and 1 in flight, first-half fragments of tile 0 ---------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 16; ++j) dma_piece(lds0, j);
  next_tile();
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_piece(lds0 + STAGE, j);
    next_tile();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) frag_read(lds0, 0, q);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // one K tile: DMA = tile t+2 exists (goes into `cur`), NEXT = tile t+1 exists (its first-half fragments come from `nxt`)
  auto tile = [&](const bool DMA, const bool NEXT, unsigned cur, unsigned nxt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 128; ++s) {
      const int ks = s >> 6, nb = (s >> 3) & 7, mb = s & 7;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nb][mb]) : "v"(wf[ks][nb]), "v"(af[ks][mb]));
      static_assert(15 * RGAP < BAR1 && BAR1 < DMA0 && DMA0 + 15 * DGAP < NX0 - 1 && NX0 + 15 * NGAP < 128, "slot plan");
      if (s < 16 * RGAP && s % RGAP == 0) frag_read(cur, 1, s / RGAP);
      if (s == BAR1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (DMA && s >= DMA0 && s < DMA0 + 16 * DGAP && (s - DMA0) % DGAP == 0) dma_piece(cur, (s - DMA0) / DGAP);
      if (NEXT && s == NX0 - 1) {
        if (DMA) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (NEXT && s >= NX0 && s < NX0 + 16 * NGAP && (s - NX0) % NGAP == 0) frag_read(nxt, 0, (s - NX0) / NGAP);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DMA) next_tile();
    if (NEXT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  int t = 0;
  for (; t + 2 < nk; ++t) {
    tile(true, true, lds0 + (t & 1) * STAGE, lds0 + ((t + 1) & 1) * STAGE);
  }
  if (t + 1 < nk) {
    tile(false, true, lds0 + (t & 1) * STAGE, lds0 + ((t + 1) & 1) * STAGE);
    ++t;
  }
  tile(false, false, lds0 + (t & 1) * STAGE, 0u);

  // ---- epilogue.  The inline-asm MFMAs are invisible to the compiler's hazard tracking: wait for the matrix pipe with the
  // accumulators as operands of the wait, so that no accumulator read can be placed above it ------------------------------------------
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[nb][0]), "+a"(acc[nb][1]), "+a"(acc[nb][2]), "+a"(acc[nb][3]), "+a"(acc[nb][4]),
                 "+a"(acc[nb][5]), "+a"(acc[nb][6]), "+a"(acc[nb][7]));
  // block (nb, mb): lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4)
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int64_t m = m0 + wm * 128 + mb * 16 + (lane & 15);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int64_t n = n0 + wn * 128 + nb * 16 + 4 * (lane >> 4);
      const f32x4 v = acc[nb][mb];
      *(u32x2*)(C + m * N + n) = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
    }
  }
}

__global__ void naive_rows(const bf16_t* A, const bf16_t* W, float* out, const int* rows, int nrows, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
  if (n >= N || ri >= nrows) return;
  const bf16_t* a = A + (int64_t)rows[ri] * K;
  const bf16_t* w = W + (int64_t)n * K;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += __uint_as_float((unsigned)a[k] << 16) * __uint_as_float((unsigned)w[k] << 16);
  out[(int64_t)ri * N + n] = s;
}

static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f(bf16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(const bf16_t*, const bf16_t*, bf16_t*, int, int, int);
static int run(kern_t kern, const char* name, int M, int N, int K, bool check_all) {
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345u + M + N + K;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& x : hA) x = f2bf(rnd());
  for (auto& x : hW) x = f2bf(rnd() * 0.1f);
  bf16_t *dA, *dW, *dC; float* dRef; int* dRows;
  (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dW, hW.size() * 2); (void)hipMalloc(&dC, (size_t)M * N * 2);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemset(dC, 0xff, (size_t)M * N * 2);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f, ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, dA, dW, dC, M, N, K);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  std::vector<int> rows;
  if (check_all) for (int i = 0; i < M; ++i) rows.push_back(i);
  else for (int i = 0; i < 64; ++i) rows.push_back((int)(((int64_t)i * 2654435761u) % M));
  if (!check_all) { rows[0] = 0; rows[1] = M - 1; rows[2] = 255; rows[3] = 256; }
  const int nr = (int)rows.size();
  (void)hipMalloc(&dRef, (size_t)nr * N * 4); (void)hipMalloc(&dRows, nr * 4);
  (void)hipMemcpy(dRows, rows.data(), nr * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, nr), dim3(256), 0, 0, dA, dW, dRef, dRows, nr, N, K);
  std::vector<float> ref((size_t)nr * N); std::vector<bf16_t> hC((size_t)M * N);
  (void)hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double max_rel = 0; long bad = 0;
  for (int i = 0; i < nr; ++i)
    for (int n = 0; n < N; ++n) {
      const float want = ref[(size_t)i * N + n], got = bf2f(hC[(size_t)rows[i] * N + n]);
      const double err = fabs(got - want) / (fabs(want) + 1.0);          // bf16 output: <= 2^-8 relative
      if (err > max_rel) max_rel = err;
      if (!(err < 8e-3)) ++bad;
    }
  printf("%-28s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s   checked %d rows: max rel err %.2e, %ld bad\n", name, M, N, K, best,
         2.0 * M * N * K / (best * 1e-3) / 1e12, nr, max_rel, bad);
  (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dRef); (void)hipFree(dRows);
  return bad != 0;
}

#define RUN(name, M, N, K, all, ...) rc |= run(gemmd_kernel<__VA_ARGS__>, name, M, N, K, all)
int main() {
  int rc = 0;
  RUN("base", 512, 512, 256, true, 0);
  RUN("base", 256, 256, 64, true, 0);                      // a single K tile
  RUN("base", 256, 512, 128, true, 0);                     // two K tiles
  RUN("stag", 2048, 2048, 1024, true, 1);
  RUN("next96/2", 2048, 2048, 1024, true, 0, 2, 36, 40, 3, 96, 2);
  RUN("early", 2048, 2048, 1024, true, 0, 1, 20, 24, 4, 96, 2);
  const int shapes[4][3] = {{131072, 5120, 5120}, {131072, 5120, 13824}, {16384, 7168, 5120}, {8192, 8192, 8192}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    RUN("base", M, N, K, false, 0);
    RUN("next96/2 dgap3", M, N, K, false, 0, 2, 36, 40, 3, 96, 2);
    RUN("early rgap1 bar20", M, N, K, false, 0, 1, 20, 24, 4, 96, 2);
    RUN("early rgap1 bar20 nx104", M, N, K, false, 0, 1, 20, 24, 5, 104, 1);
    RUN("dgap2", M, N, K, false, 0, 2, 36, 40, 2, 104, 1);
    RUN("group_m 8", M, N, K, false, 0, 2, 36, 40, 4, 104, 1, 8);
    RUN("group_m 2", M, N, K, false, 0, 2, 36, 40, 4, 104, 1, 2);
    RUN("base again", M, N, K, false, 0);
  }
  printf(rc ? "FAILED\n" : "all checks passed\n");
  return rc;
}
