// Probe (gfx950): one wave per SIMD, 32 MFMA per iteration with the softmax instruction mix of a 32x64 attention
// sub-tile (32 v_fma, 32 v_exp, 32 v_add, 16 v_cvt_pk, 16 v_max3) interleaved evenly between them, plus LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;

template <int VALU, int LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float*)smem)[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f); b[j] = (__bf16)1.0f; }
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
  float c1 = 1.0001f + threadIdx.x * 1e-9f, c2 = 0.5f;
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      bf16x8 aa = a;
      if (LDS && (u % LDS) == 0) aa = *(lds_bf16x8*)(uintptr_t)(lbase + (u & 15) * 1024);
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, b, acc[u & 3], 0, 0, 0);
      if (VALU) {
        // per MFMA: 1 fma, 1 exp, 1 add, and alternately 1 cvt_pk or 1 max3  (= 128 VALU per 32 MFMA)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(c1), "v"(c2));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[(u + 3) & 7]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(u + 5) & 7]) : "v"(c2));
        if (u & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[(u + 6) & 7]) : "v"(c2));
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[(u + 6) & 7]) : "v"(c1), "v"(c2));
        if (VALU > 1) {   // + address / bookkeeping ops: 1 more simple op per MFMA
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(u + 1) & 7]) : "v"(c1));
        }
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int k = 0; k < 8; ++k) r += v[k];
  if (r == 123.456f) sink[threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int VALU, int LDS, int WAVES>
void run(const char* what, unsigned long long* out, float* sink, int iters) {
  unsigned long long h;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<VALU, LDS, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  const double flop = 256.0 * WAVES * 32.0 * iters * 32768.0;
  printf("%-64s %7.1f ns / 32 MFMA per wave; kernel %.3f ms -> %.0f TFLOP/s aggregate\n", what, h * 10.0 / iters, ms, flop / ms / 1e9);
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096);
  const int iters = 10000;
  run<0, 0, 4>("warm-up", out, sink, iters * 8);
  run<0, 0, 4>("4 waves/CU: MFMA only", out, sink, iters);
  run<1, 0, 4>("4 waves/CU: + softmax mix (4 VALU incl. exp per MFMA)", out, sink, iters);
  run<2, 0, 4>("4 waves/CU: + softmax mix + 1 more VALU per MFMA", out, sink, iters);
  run<0, 1, 4>("4 waves/CU: MFMA + 1 ds_read_b128 per MFMA", out, sink, iters);
  run<1, 1, 4>("4 waves/CU: softmax mix + 1 ds_read_b128 per MFMA", out, sink, iters);
  run<1, 2, 4>("4 waves/CU: softmax mix + 1 ds_read_b128 per 2 MFMA", out, sink, iters);
  run<2, 2, 4>("4 waves/CU: softmax mix + 1 VALU + 1 ds_read per 2 MFMA", out, sink, iters);
  run<0, 0, 8>("8 waves/CU: MFMA only (per wave: 2x)", out, sink, iters);
  run<1, 1, 8>("8 waves/CU: softmax mix + 1 ds_read_b128 per MFMA", out, sink, iters);
  run<1, 2, 8>("8 waves/CU: softmax mix + 1 ds_read_b128 per 2 MFMA", out, sink, iters);
  return 0;
}
