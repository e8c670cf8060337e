// Stand-alone candidate for the next GEMM main loop (NOT part of libvita_hip.so; written at the end of round 1 after
// staging.hip showed v_mfma_f32_16x16x32_bf16 sustaining 1.77 PFLOP/s under full operand staging where 32x32x16 sustains 1.52).
//   C[M,N] = A[M,K] @ W[N,K]^T, bf16 in, fp32 accumulate, bf16 out;  M, N multiples of 256, K a multiple of 64.
// One workgroup = 4 waves = one 256 x 256 tile, BK = 64, each wave 128 x 128 = 8 x 8 blocks of 16 x 16 (256 accumulator
// registers), operands HBM/L2 -> VGPR -> LDS (two 64 KiB stages, the library's [rows][64] layout and swizzle, which is also
// conflict-free for the 16-row fragments), program order = issue order (sched_barrier after every slot).
// Per k-step of 64 MFMAs, one memory instruction behind every second MFMA (even slot 2e, e = 0..31):
//   e % 4 == 0 : W fragment of n-block B+2 into a ring of four (the MFMAs walk n-blocks outermost, so a W fragment serves eight
//                consecutive MFMAs and only the A fragments need two full sets)
//   k-step 0   : the other 24: 8 A-fragment reads of k-step 1, then the 16 ds_write_b128 of tile t+1
//   k-step 1   : 12 of the other 24 + four odd slots of the first half: the 16 global_load_dwordx4 of tile t+2 (52 MFMAs before
//                their first store), [lgkmcnt(0) + s_barrier at slot 42, behind the last read of tile t], 8 A-fragment reads of tile t+1's k-step 0
// It checks itself against a naive kernel (every element of a 512 x 512 x 256 problem, sampled rows of the big ones) and
// times the decoder shapes.   hipcc --offload-arch=gfx950 -O3 gemm16.hip -o gemm16 && ./gemm16
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
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int BK = 64, BM = 256, BN = 256, A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;

__device__ __forceinline__ int tile_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ unsigned pack2(float a, float b) {         // RNE fp32 -> bf16 pairs
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

__global__ __launch_bounds__(256, 1) void gemm16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
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
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n, group = pid / per_group, first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M), in_group = pid - group * per_group;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  // staging piece j (0..7 per operand) of a wave: tile rows (wave*8 + j)*8 + lane/8, 16-byte slot lane & 7.  Address =
  // uniform base (SGPRs: tile, wave, piece, K tile) + one per-lane 32-bit offset, so the 16 loads share two VGPRs
  const int r0 = wave * 64 + (lane >> 3), sl = lane & 7;
  const char* a_base = (const char*)(A + (m0 + wave * 64) * (int64_t)K);
  const char* w_base = (const char*)(W + (n0 + wave * 64) * (int64_t)K);
  const unsigned lane_off = (unsigned)(((lane >> 3) * K + sl * 8) * 2);
  const int64_t row8 = (int64_t)8 * K * 2;                               // bytes between consecutive pieces
  int64_t kbyte = 0;                                                     // byte offset of the K tile being fetched
  // LDS destination: row r0 + 8j -> the swizzle term ((row >> 1) & 7) alternates with j & 1
  const unsigned d_even = tile_off(r0, sl), d_odd = tile_off(r0 + 8, sl) - 8 * 128;
  // fragment read offsets: rows (lane & 15) of a 16-row block, 16-byte slot 4*ks + (lane >> 4)
  unsigned fa[2], fw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    fa[ks] = tile_off(wm * 128 + (lane & 15), ks * 4 + (lane >> 4));
    fw[ks] = A_BYTES + tile_off(wn * 128 + (lane & 15), ks * 4 + (lane >> 4));
  }
  f32x4 acc[8][8];                                                       // [n block][m block]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 g[16];
  bf16x8 af[2][8], wr[4];
  const int nk = K / BK;
  auto load_piece = [&](int j) __attribute__((always_inline)) {
    const char* base = j < 8 ? a_base + j * row8 : w_base + (j - 8) * row8;
    g[j] = *(const u32x4*)(base + kbyte + lane_off);
  };
  auto store_piece = [&](int j, unsigned stage) __attribute__((always_inline)) {
    const int q = j & 7;
    const unsigned off = (j < 8 ? 0 : A_BYTES) + ((q & 1) ? d_odd : d_even) + q * 8 * 128;
    *(lds_u32x4*)(uintptr_t)(stage + off) = g[j];
  };
  auto read_a = [&](unsigned base, int ks, int i) __attribute__((always_inline)) {       // A fragment of m-block i, k-step ks
    af[ks][i] = *(lds_bf16x8*)(uintptr_t)(base + fa[ks] + i * 16 * 128);
  };
  auto read_w = [&](unsigned cur, unsigned nxt, int B) __attribute__((always_inline)) {  // W fragment of block B (16.. = next tile)
    const unsigned base = B < 16 ? cur : nxt;
    wr[B & 3] = *(lds_bf16x8*)(uintptr_t)(base + fw[(B >> 3) & 1] + (B & 7) * 16 * 128);
  };
  // prologue: tile 0 -> stage 0, tile 1 -> registers, A fragments of (tile 0, k-step 0), W fragments of blocks 0 and 1
#pragma unroll
  for (int j = 0; j < 16; ++j) load_piece(j);
  if (nk > 1) kbyte += BK * 2;
#pragma unroll
  for (int j = 0; j < 16; ++j) store_piece(j, lds0);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) load_piece(j);
  if (nk > 2) kbyte += BK * 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) read_a(lds0, 0, i);
  read_w(lds0, lds0, 0);
  read_w(lds0, lds0, 1);

  for (int t = 0; t < nk; ++t) {
    const unsigned cur = lds0 + (t & 1) * STAGE, nxt = lds0 + ((t + 1) & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
          const int s = nb * 8 + mb, B = ks * 8 + nb;                    // slot 0..63 of this k-step; n-block 0..15 of the tile
          if (ks == 1 && s == 42) {       // behind the LAST read of this tile (the W-ring read of n-block 15 at slot 40): the
                                          // next iteration stores into this stage without another barrier
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[B & 3], af[ks][mb], acc[nb][mb], 0, 0, 0);
          if ((s & 1) == 0) {
            const int e = s >> 1;                                        // 0..31
            if ((e & 3) == 0) {
              read_w(cur, nxt, B + 2);                                   // e = 4 nb: two n-blocks ahead
            } else {
              const int o = e - (e >> 2) - 1;                            // 0..23: the other even slots
              if (ks == 0 && o < 8) read_a(cur, 1, o);
              if (ks == 0 && o >= 8) store_piece(o - 8, nxt);            // tile t+1 (loaded during the previous iteration)
              if (ks == 1 && o < 12) load_piece(o);                      // tile t+2
              if (ks == 1 && o >= 16) read_a(nxt, 0, o - 16);            // behind the barrier (o = 16 is slot 44)
            }
          } else if (ks == 1 && s < 32 && (s & 7) == 1) {
            load_piece(12 + (s >> 3));                                   // the last four loads, also in the first half: hipcc
          }                                                              // waits with vmcnt(0) before the first store of the next
                                                                         // iteration, so the LAST load sets the latency budget
          __builtin_amdgcn_sched_barrier(0);
        }
    if (t + 3 < nk) kbyte += BK * 2;                                    // the fetch stops at the last K tile
  }
  // epilogue: block (nb, mb): lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4)
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

static int run(int M, int N, int K, bool check_all) {
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
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f, ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(gemm16_kernel, dim3(grid), dim3(256), 2 * STAGE, 0, dA, dW, dC, M, N, K);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // check: every row (small problem) or 64 sampled rows incl. first / last of tiles
  std::vector<int> rows;
  if (check_all) for (int i = 0; i < M; ++i) rows.push_back(i);
  else for (int i = 0; i < 64; ++i) rows.push_back((int)(((int64_t)i * 2654435761u) % M));
  rows[0] = 0; rows[1] = M - 1; rows[2] = 255; rows[3] = 256;
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
  printf("M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s   checked %d rows: max rel err %.2e, %ld bad\n", M, N, K, best,
         2.0 * M * N * K / (best * 1e-3) / 1e12, nr, max_rel, bad);
  (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dRef); (void)hipFree(dRows);
  return bad != 0;
}

int main() {
  int rc = run(512, 512, 256, true);
  rc |= run(256, 256, 64, true);                      // a single K tile
  rc |= run(8192, 8192, 8192, false);
  rc |= run(131072, 5120, 5120, false);
  rc |= run(131072, 5120, 13824, false);
  rc |= run(16384, 7168, 5120, false);
  printf(rc ? "FAILED\n" : "all checks passed\n");
  return rc;
}
