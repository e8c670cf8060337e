// Probe (gfx950), round 4: can the attention backward emit dQ as fp32 atomics?  
//
// The fused dK + dQ kernel under design is KV-outer: a workgroup owns 256 keys of one kv head, walks the
// 32-row query halves that see them and, per half, adds a [32 q x 128 d] fp32 tile (the sum over ITS 256 keys) into dQ — 16 KB of
// atomics per workgroup per ~64 MFMAs per wave.  Every wave owns one 32-column block of the tile: one `global_atomic_add_f32` (no
// return) per accumulator register, 64 lanes = 2 rows x 32 consecutive floats = two whole 128-byte lines per instruction.
// This program reproduces exactly that stream — 256 workgroups x 4 waves, one per SIMD (100 KB of LDS requested), per iteration
// NMFMA MFMAs (32 x 32 x 16 bf16, pseudo-random operands) with NATOM atomic instructions spread evenly behind them, tiles rotating
// through a dQ buffer of the 16K geometry (5 query heads x 512 halves x 16 KB per kv group) — and varies
//   scope:      agent (`sc1`: the only form that is correct when two XCDs touch a line)  |  none (executes in the XCD's own L2:
//               correct only if EVERY contributor to a line runs on the same XCD — kv group g <-> XCD g via blockIdx % 8)
//   placement:  XCD-local (workgroup b adds into the buffer of kv group b % 8)  |  shared (group (b / 8) % 8: every line is hit from all XCDs)
// and reports ns per iteration, the atomic byte rate, and whether the final sums are right (every add is 1.0f; HW_REG_XCC_ID of
// every workgroup is recorded to check the b % 8 placement the XCD-local form relies on).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int HEADS = 5, HALVES = 512, TILE_FLOATS = 32 * 128;          // per kv group: 5 x 512 x 16 KB = 40 MB
constexpr size_t GROUP_FLOATS = (size_t)HEADS * HALVES * TILE_FLOATS;

template <bool SC1>
__device__ __forceinline__ void atom(unsigned voff, float v, const float* base) {
  if (SC1) asm volatile("global_atomic_add_f32 %0, %1, %2 sc1" :: "v"(voff), "v"(v), "s"(base) : "memory");
  else asm volatile("global_atomic_add_f32 %0, %1, %2" :: "v"(voff), "v"(v), "s"(base) : "memory");
}

// NMFMA MFMAs and NATOM atomic instructions per iteration and wave; SC1: agent scope; LOCAL: XCD-local placement
template <int NMFMA, int NATOM, bool SC1, bool LOCAL>
__global__ __launch_bounds__(256, 1) void probe(float* dq, unsigned long long* out, int* xcc, float* sink, int iters) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  if (tid == 0) { xcc[bid] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }   // HW_REG_XCC_ID, bits 0..3
  const int grp = LOCAL ? (bid & 7) : ((bid >> 3) & 7);
  const int j = bid >> 3;                                              // 0..31: the key block inside the group (staggered starts)
  const float* gbase = dq + (size_t)grp * GROUP_FLOATS;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  bf16x8 a, b;
  unsigned rng = tid * 2654435761u + bid * 40503u + 12345u;
  for (int k = 0; k < 8; ++k) {
    rng = rng * 1664525u + 1013904223u; a[k] = (__bf16)(((int)(rng >> 16) & 255) * (1.0f / 128.f) - 1.0f);
    rng = rng * 1664525u + 1013904223u; b[k] = (__bf16)(((int)(rng >> 16) & 255) * (1.0f / 128.f) - 1.0f);
  }
  // lane -> float offset inside a tile for accumulator register r: row (r & 3) + 8 (r >> 2) + 4 hi, column 32 wave + l31
  const unsigned lane_off = (unsigned)(((4 * hi) * 128 + 32 * wave + l31) * 4);
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const int head = (it / HALVES) % HEADS, half = (j * 16 + it) % HALVES;
    const float* tile = gbase + ((size_t)head * HALVES + half) * TILE_FLOATS;
    asm volatile("" : "+s"(tile));
    constexpr int STEP = NATOM ? (NMFMA ? NMFMA / NATOM : 1) : 0;
#pragma unroll
    for (int u = 0; u < (NMFMA ? NMFMA : NATOM); ++u) {
      if (NMFMA) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      if (NATOM && (u % (STEP ? STEP : 1)) == 0 && u / (STEP ? STEP : 1) < NATOM) {
        const int r = (u / (STEP ? STEP : 1)) & 15;                     // NATOM = 64: each register four times (the four-partials form)
        const unsigned off = lane_off + (unsigned)((((r & 3) + 8 * (r >> 2)) * 128) * 4);
        atom<SC1>(off, 1.0f, tile);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
  if (r == 123.456f) sink[tid] = r;
  if (tid == 0) out[bid] = t1 - t0;
  (void)smem;
}

__global__ void sum_kernel(const float* p, size_t n, double* out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

struct Bufs { float* dq; unsigned long long* out; int* xcc; float* sink; double* sum; };

template <int NMFMA, int NATOM, bool SC1, bool LOCAL>
void run(const char* what, const Bufs& b, int iters) {
  constexpr int LDS = 100 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NMFMA, NATOM, SC1, LOCAL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const size_t n = 8 * GROUP_FLOATS;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(b.dq, 0, n * 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NMFMA, NATOM, SC1, LOCAL>), dim3(256), dim3(256), LDS, 0, b.dq, b.out, b.xcc, b.sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipMemset(b.sum, 0, 8);
  hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, 0, b.dq, n, b.sum);
  double got = 0;
  (void)hipMemcpy(&got, b.sum, 8, hipMemcpyDeviceToHost);
  std::vector<int> xcc(256);
  (void)hipMemcpy(xcc.data(), b.xcc, 256 * 4, hipMemcpyDeviceToHost);
  int placed = 0;
  for (int i = 0; i < 256; ++i) placed += xcc[i] == (i & 7);
  const double want = 256.0 * 4 * 64 * (double)NATOM * iters;
  const double bytes = want * 4, flop = 256.0 * 4 * (double)NMFMA * iters * 32768.0;
  printf("%-86s %8.1f ns/iter  atomics %6.2f TB/s  MFMA %5.0f TFLOP/s  sum %s (%.0f / %.0f)  b%%8==XCC_ID for %d/256\n", what,
         ms * 1e6 / iters, bytes / ms / 1e9, flop / ms / 1e9, got == want ? "ok" : "WRONG", got, want, placed);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  Bufs b;
  (void)hipMalloc(&b.dq, 8 * GROUP_FLOATS * 4);
  (void)hipMalloc(&b.out, 256 * 8); (void)hipMalloc(&b.xcc, 256 * 4); (void)hipMalloc(&b.sink, 4096); (void)hipMalloc(&b.sum, 8);
  run<64, 0, false, true>("warm-up", b, iters);
  run<64, 0, false, true>("64 MFMA per wave and iteration, no atomics", b, iters);
  run<128, 0, false, true>("128 MFMA, no atomics", b, iters);
  run<0, 16, true, true>("16 atomic instr (16 KB / workgroup), agent scope, XCD-local lines, no MFMA", b, iters);
  run<0, 16, false, true>("16 atomic instr, no scope bits (L2), XCD-local lines, no MFMA", b, iters);
  run<0, 16, true, false>("16 atomic instr, agent scope, lines shared by all XCDs, no MFMA", b, iters);
  run<0, 16, false, false>("16 atomic instr, no scope bits, lines shared by all XCDs (expected WRONG or slow), no MFMA", b, iters);
  run<64, 16, true, true>("64 MFMA + 16 atomic instr, agent scope, XCD-local", b, iters);
  run<64, 16, false, true>("64 MFMA + 16 atomic instr, no scope bits, XCD-local", b, iters);
  run<64, 16, true, false>("64 MFMA + 16 atomic instr, agent scope, shared by all XCDs", b, iters);
  run<128, 16, true, true>("128 MFMA + 16 atomic instr (the real kernel's ratio at ~1 PFLOP/s), agent, XCD-local", b, iters);
  run<128, 16, false, true>("128 MFMA + 16 atomic instr, no scope bits, XCD-local", b, iters);
  run<128, 16, true, false>("128 MFMA + 16 atomic instr, agent, shared by all XCDs", b, iters);
  run<64, 64, true, true>("64 MFMA + 64 atomic instr (every wave adds its own 64-key partial), agent, XCD-local", b, iters);
  run<64, 64, false, true>("64 MFMA + 64 atomic instr, no scope bits, XCD-local", b, iters);
  return 0;
}
