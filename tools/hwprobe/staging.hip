// Probe (gfx950): what does operand staging cost a GEMM main loop?  One workgroup per CU, WAVES waves, no barrier.  Per
// iteration every wave issues SLOTS MFMAs (32x32x16 bf16, or pairs of 16x16x32 with M16 = 1) and, spread evenly behind them,
//   R  ds_read_b128   fragment reads (conflict-free addresses)
//   W  ds_write_b128  staging stores of registers (CHAIN = 1: the registers the G loads of the PREVIOUS iteration
//                     returned, each store behind a counted vmcnt)
//   G  global_load_dwordx4 from a 64 KiB-per-CU window (L1 / L2 hits)
//   D  global_load_lds_dwordx4 LDS-DMA pieces from the same window
// and the run is long enough (>= 5 ms) for the sustained clock.  Reports the MFMA rate as PFLOP/s over 256 CUs.
// The decoder GEMM's 256x256x64 tile is, per K tile and wave: 4 waves: SLOTS 64, R 32, and W 16 + G 16 (register staging)
// or D 16 (LDS-DMA); 8 waves: SLOTS 32, R 24, D 8.   hipcc --offload-arch=gfx950 -O3 staging.hip -o staging && ./staging
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

template <int WAVES, int SLOTS, int R, int W, int G, int D, int CHAIN, int M16>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe(const char* __restrict__ win, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) ((float*)smem)[i] = 1e-4f * i;
  __syncthreads();
  f32x16 acc[4];                          // 4 independent accumulation chains (a chain is revisited every 128 cycles)
  f32x4 acc16[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc16[i][j] = 0.f;
  bf16x8 fr[16];                          // a fragment is read 9 MFMAs before its first use (the GEMMs read one k-step ahead)
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 8; ++j) fr[i][j] = (__bf16)(1e-3f * (lane + i));
  u32x4 g[16];
  for (int i = 0; i < 16; ++i) g[i] = (u32x4){(unsigned)lane, 1u, 2u, (unsigned)i};
  // fragment read: row = lane & 31 of a [rows][64] bf16 tile with the GEMM's swizzle; staging store / DMA: lane-linear 1 KiB
  const unsigned rd = lds0 + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4) + wave * 4096;
  const unsigned wr = lds0 + 65536 + wave * (65536 / WAVES) + lane * 16;     // 64 KiB of staging area per CU
  const char* gp = win + (size_t)blockIdx.x * 65536 + wave * (65536 / WAVES) + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (M16) {
        acc16[(2 * s) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s & 15], fr[(s + 3) & 15], acc16[(2 * s) & 7], 0, 0, 0);
        acc16[(2 * s + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s & 15], fr[(s + 3) & 15], acc16[(2 * s + 1) & 7], 0, 0, 0);
      } else {
        acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s & 15], fr[(s + 3) & 15], acc[s & 3], 0, 0, 0);
      }
      if (R && (s * R) / SLOTS != ((s + 1) * R) / SLOTS)             // R of the SLOTS slots, evenly
        fr[(s + 12) & 15] = *(lds_bf16x8*)(uintptr_t)(rd + (((s * R) / SLOTS) & 3) * 4096);
      if (W && (s * W) / SLOTS != ((s + 1) * W) / SLOTS) {
        const int j = (s * W) / SLOTS;
        // the store is issued before this slot's load, so G loads are in flight and the oldest one (issued a whole
        // iteration = SLOTS MFMAs ago) is the one this store needs
        if (CHAIN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G - 1) : "memory");
        *(lds_u32x4*)(uintptr_t)(wr + (j & 15) * 1024) = g[j & 15];
      }
      if (G && (s * G) / SLOTS != ((s + 1) * G) / SLOTS) {
        const int j = (s * G) / SLOTS;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g[j & 15]) : "v"(gp + (j & 15) * 1024) : "memory");
      }
      if (D && (s * D) / SLOTS != ((s + 1) * D) / SLOTS) {
        const int j = (s * D) / SLOTS;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(gp + (j & 15) * 1024), (lds_void*)(uintptr_t)(wr - lane * 16 + (j & 15) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (D || (G && !CHAIN)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += acc16[i][0];
  for (int i = 0; i < 16; ++i) r += (float)g[i][0];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int WAVES, int SLOTS, int R, int W, int G, int D, int CHAIN, int M16>
void run(const char* what, const char* win, float* sink) {
  auto k = probe<WAVES, SLOTS, R, W, G, D, CHAIN, M16>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400000 / SLOTS;                                   // ~12.8 M MFMA cycles per SIMD: >= 5 ms
  float best = 1e30f, ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 160 * 1024 - 1024, 0, win, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flop = 256.0 * WAVES * (double)iters * SLOTS * 2.0 * 32 * 32 * 16;
  printf("%-66s %8.3f ms  %6.3f PFLOP/s\n", what, best, flop / (best * 1e-3) / 1e15);
}

int main() {
  char* win; float* sink;
  (void)hipMalloc(&win, 256 * 65536 + 65536); (void)hipMemset(win, 1, 256 * 65536 + 65536);
  (void)hipMalloc(&sink, 4096);
  run<4, 64, 0, 0, 0, 0, 0, 0>("4 waves: MFMA only", win, sink);
  run<4, 64, 0, 0, 0, 0, 0, 1>("4 waves: MFMA only, 16x16x32 pairs", win, sink);
  run<4, 64, 32, 0, 0, 0, 0, 0>("4 waves: + 32 fragment reads", win, sink);
  run<4, 64, 32, 16, 0, 0, 0, 0>("4 waves: + 32 reads + 16 ds_write_b128", win, sink);
  run<4, 64, 32, 0, 16, 0, 0, 0>("4 waves: + 32 reads + 16 global loads (waited per iteration)", win, sink);
  run<4, 64, 32, 16, 16, 0, 1, 0>("4 waves: + 32 reads + 16 loads -> 16 stores, counted vmcnt", win, sink);
  run<4, 64, 32, 16, 16, 0, 1, 1>("4 waves: same with 16x16x32 pairs", win, sink);
  run<4, 64, 32, 0, 0, 16, 0, 0>("4 waves: + 32 reads + 16 LDS-DMA pieces", win, sink);
  run<8, 32, 0, 0, 0, 0, 0, 0>("8 waves: MFMA only", win, sink);
  run<8, 32, 24, 0, 0, 0, 0, 0>("8 waves: + 24 fragment reads", win, sink);
  run<8, 32, 24, 0, 0, 8, 0, 0>("8 waves: + 24 reads + 8 LDS-DMA pieces", win, sink);
  run<8, 32, 24, 8, 8, 0, 1, 0>("8 waves: + 24 reads + 8 loads -> 8 stores, counted vmcnt", win, sink);
  return 0;
}
