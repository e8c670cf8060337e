// comm_overlap.hip — does a communication kernel get CUs while the attention kernel owns the chip?  (VERDICT r2 "next round" 4)
//
// Every hot kernel of this library is ONE 512-register workgroup per CU; RCCL's all-gather is itself a CU kernel (a few
// "channels" = workgroups of 256 threads with few registers that stay resident until the collective is done).  The
// "gather j + 1 runs under attention j" claim of dot_product_attention.forward_cp only holds if such a kernel is given CUs
// while 640 attention workgroups are queued.  This file is the stand-in for RCCL's kernel on a 1-GPU box:
//
//   probe_channel_copy   n_channels workgroups x 256 threads copy `bytes` (16 B per lane, grid-stride), stamp the device-wide
//                        constant-rate clock (s_memrealtime, 100 MHz) at their first and last instruction, and — to model a
//                        kernel that is paced by an xGMI link rather than by HBM — stay resident (s_sleep) until
//                        `min_ticks` have passed since their start.
//   probe_stamp          one lane writes the clock: brackets the attention launch on ITS stream with the same clock.
//
// Built as a small shared library (hipcc --offload-arch=gfx950 -O3 -shared -fPIC) and driven from tools/probe_comm_overlap.py
// through ctypes, next to libvita_hip.so's own vita_flash_attn_fwd on a second stream.  Test infrastructure, not product.
#include <hip/hip_runtime.h>
#include <stdint.h>

// HEAVY = RCCL's real footprint.  Read from this image's librccl.so (gfx950 code object, llvm-readelf --notes):
// rcclGenericKernel<1|2|4, ...> — the one kernel behind every collective — is 256 threads with .vgpr_count 261 / 278 / 280
// (VGPR + AGPR; allocation granule 8) and 19744 B of LDS, i.e. ONE wave per SIMD: a channel cannot share a CU with one of this
// library's 448-to-512-register workgroups and has to wait for a whole CU.  The HEAVY variant pins the same 280 registers
// (v255 + a23 clobbered) and the same LDS; the light one (a dozen registers) can slip into the registers an attention workgroup
// leaves free.
template <bool HEAVY>
__global__ __launch_bounds__(256) void channel_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16,
                                                           unsigned long long* __restrict__ stamps, unsigned long long min_ticks) {
  const unsigned long long t0 = wall_clock64();
  if (HEAVY) {
    __shared__ unsigned lds[19744 / 4];
    asm volatile("" ::: "v255", "a23");
    lds[threadIdx.x] = (unsigned)t0;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 0xdeadbeefu && min_ticks == 12345) stamps[0] = 0;     // keeps the LDS allocation alive
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  __syncthreads();
  unsigned long long t1 = wall_clock64();
  while (t1 - t0 < min_ticks) {          // link-paced residency: the channel keeps its CU slot until the "transfer" is over
    __builtin_amdgcn_s_sleep(64);
    t1 = wall_clock64();
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

__global__ void stamp_kernel(unsigned long long* out) { *out = wall_clock64(); }

extern "C" int probe_channel_copy(void* dst, const void* src, int64_t bytes, int n_channels, void* stamps, unsigned long long min_ticks,
                                  int heavy, void* stream) {
  if (heavy)
    hipLaunchKernelGGL(channel_copy_kernel<true>, dim3((unsigned)n_channels), dim3(256), 0, (hipStream_t)stream, (uint4*)dst,
                       (const uint4*)src, bytes / 16, (unsigned long long*)stamps, min_ticks);
  else
    hipLaunchKernelGGL(channel_copy_kernel<false>, dim3((unsigned)n_channels), dim3(256), 0, (hipStream_t)stream, (uint4*)dst,
                       (const uint4*)src, bytes / 16, (unsigned long long*)stamps, min_ticks);
  return (int)hipGetLastError();
}

extern "C" int probe_stamp(void* out, void* stream) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
  return (int)hipGetLastError();
}

extern "C" int probe_memcpy_async(void* dst, const void* src, int64_t bytes, int kind, void* stream) {
  // kind 0 = device to device (the runtime's choice of engine), 1 = host to device, 2 = device to host (pinned host memory: SDMA)
  const hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToDevice : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  return (int)hipMemcpyAsync(dst, src, (size_t)bytes, k, (hipStream_t)stream);
}
