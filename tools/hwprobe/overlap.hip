// Probe (gfx950): do MFMA work of one wave and VALU / transcendental work of ANOTHER wave on the same SIMD overlap?
// And inside one wave, with the VALU ops pinned between the MFMAs?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 = role A (MFMA stream), waves 4-7 = role B (VALU stream);
// wave w and w+4 share a SIMD.  Each role reports its own duration (100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// A: 0 idle, 1 = 16 MFMA per iter, 2 = 16 x (MFMA + NV fma pinned behind it)
// B: 0 idle, 1 = 64 independent fma per iter, 2 = 64 independent v_exp per iter
template <int A, int B, int NV, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, float* sink, int iters) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  const unsigned long long t0 = wall_clock64();
  if (wave < 4) {
    if (A) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f); b[j] = (__bf16)1.0f; }
      float v[8];
      for (int k = 0; k < 8; ++k) v[k] = threadIdx.x + k;
      float c1 = 1.0001f + threadIdx.x * 1e-9f, c2 = 0.5f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
          if (A == 2) {
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(c1), "v"(c2));
          }
        }
        if (A == 2) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
          }
        }
      }
      for (int i = 0; i < 4; ++i) r += acc[i][0];
      for (int k = 0; k < 8; ++k) r += v[k];
    }
  } else if (B) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
    float c1 = 1.0001f + threadIdx.x * 1e-9f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (B == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c1), "v"(c2));
          if (B == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
          if (B == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(c2));
          if (B == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[k]) : "v"(c2));
          if (B == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c1), "v"(c2));
        }
      }
    }
    for (int k = 0; k < 8; ++k) r += v[k];
  }
  const unsigned long long t1 = wall_clock64();
  if (r == 123.456f) sink[threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int A, int B, int NV, int PRIO = 0>
void run(const char* what, unsigned long long* out, float* sink, int iters) {
  unsigned long long h[8];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<A, B, NV, PRIO>), dim3(256), dim3(512), 0, 0, out, sink, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-46s role A %7.1f ns/iter   role B %7.1f ns/iter\n", what, h[0] * 10.0 / iters, h[4] * 10.0 / iters);
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096);
  const int iters = 20000;
  run<1, 0, 0>("warm-up", out, sink, iters * 4);
  run<1, 0, 0>("A: 16 MFMA 32x32x16 bf16", out, sink, iters);
  run<0, 1, 0>("B: 64 v_fma_f32", out, sink, iters);
  run<0, 2, 0>("B: 64 v_exp_f32", out, sink, iters);
  run<0, 3, 0>("B: 64 v_add_f32", out, sink, iters);
  run<0, 4, 0>("B: 64 v_cvt_pk_bf16_f32", out, sink, iters);
  run<0, 5, 0>("B: 64 v_max3_f32", out, sink, iters);
  run<1, 1, 0>("A: 16 MFMA || B: 64 v_fma_f32", out, sink, iters);
  run<1, 2, 0>("A: 16 MFMA || B: 64 v_exp_f32", out, sink, iters);
  run<1, 1, 0, 1>("A: 16 MFMA || B: 64 v_fma_f32, B prio 3", out, sink, iters);
  run<2, 0, 2>("A: 16 x (MFMA + 2 v_fma_f32) one wave", out, sink, iters);
  run<2, 0, 4>("A: 16 x (MFMA + 4 v_fma_f32) one wave", out, sink, iters);
  run<2, 0, 6>("A: 16 x (MFMA + 6 v_fma_f32) one wave", out, sink, iters);
  run<2, 0, 8>("A: 16 x (MFMA + 8 v_fma_f32) one wave", out, sink, iters);
  run<2, 0, 12>("A: 16 x (MFMA + 12 v_fma_f32) one wave", out, sink, iters);
  run<2, 1, 4>("A: 16 x (MFMA + 4 fma) || B: 64 v_fma_f32", out, sink, iters);
  run<2, 2, 4>("A: 16 x (MFMA + 4 fma) || B: 64 v_exp_f32", out, sink, iters);
  return 0;
}
