// Probe (gfx950), round 4: dQ of the attention backward by ORDERED read-modify-write through one XCD's L2 instead of fp32 atomics.
//
// profiles/r04_hwprobe_atomic_dq.txt: fp32 atomics cap at 1.36 TB/s chip-wide, which bounds a KV-outer kernel that owns 128 keys per
// workgroup (the dK / dV accumulators a workgroup's registers hold) at 0.44 PFLOP/s.  The other 5-unit form keeps the KV-outer loop
// but serialises the contributors to a dQ tile instead of letting the atomic unit do it: the workgroups of one query head all run on
// ONE XCD (head <-> XCD via blockIdx % 8, verified by HW_REG_XCC_ID), workgroup j (key tile j) walks the query tiles i = j, j+1, ...
// and adds its [64 q x 128 d] fp32 partial into dQ[i] with plain loads (sc1: past the CU's L1, served by the XCD's L2) and plain
// stores (write-through to the same L2), AFTER key tile j - 1 has done so: a per-tile counter in the same L2, bumped by every wave
// once its stores are acknowledged (s_waitcnt vmcnt(0)), polled by the successor.  Because workgroup j - 1 reaches tile i one step
// before workgroup j does, the wait is normally already satisfied: a software wavefront, no cross-XCD coherence traffic.
//
// This program reproduces that stream — 256 workgroups x 4 waves (one per SIMD, 100 KB of LDS requested), XCD x = blockIdx % 8 owns
// HEADS query heads, key tile j = blockIdx / 8 (32 tiles in flight per XCD, as with one workgroup per CU), NQ query tiles per head —
// with NMFMA MFMAs (32 x 32 x 16 bf16) per wave and step (80 = five GEMM units of a 64 x 128 x 128 block) around the RMW, and reports
// ns per step against the MFMA-only floor, the L2 byte rate of the RMW, whether every tile holds exactly the ordered sum, how many
// polls found the counter not ready, and whether any poll hit the bound (polls are BOUNDED: a workgroup that gives up sets a global
// flag, every other workgroup stops waiting, the run reports FAILED instead of hanging).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

constexpr int HEADS = 3, NQ = 160, KT = 32, TILE_FLOATS = 64 * 128;      // per XCD: 3 heads x 160 tiles x 32 KB = 15 MB of dQ
constexpr int POLL_LIMIT = 1 << 14;

__device__ __forceinline__ f32x4 load_l2(const float* p) {                // past the L1: the XCD's L2 answers
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ int load_l2_int(const int* p) {
  int v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// MODE 0: MFMA only   1: ordered RMW (the design)   2: RMW without the wait (L2 cost alone; sums may be wrong)
template <int NMFMA, int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* dq, int* counters, int* flags, unsigned long long* stats, int* xcc, float* sink) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x, x = bid & 7, j = bid >> 3;
  if (tid == 0) xcc[bid] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  bf16x8 a, b;
  unsigned rng = tid * 2654435761u + bid * 40503u + 12345u;
  for (int k = 0; k < 8; ++k) {
    rng = rng * 1664525u + 1013904223u; a[k] = (__bf16)(((int)(rng >> 16) & 255) * (1.0f / 128.f) - 1.0f);
    rng = rng * 1664525u + 1013904223u; b[k] = (__bf16)(((int)(rng >> 16) & 255) * (1.0f / 128.f) - 1.0f);
  }
  const float mine = (float)(j + 1);
  unsigned long long not_ready = 0, polls = 0;
  bool gave_up = false;
  const unsigned long long t0 = wall_clock64();
  for (int h = 0; h < HEADS; ++h) {
    for (int i = j; i < NQ; ++i) {
      float* tile = dq + (((size_t)x * HEADS + h) * NQ + i) * TILE_FLOATS + (size_t)wave * (TILE_FLOATS / 4) + lane * 4;
      int* ctr = counters + ((size_t)x * HEADS + h) * NQ + i;
#pragma unroll
      for (int u = 0; u < NMFMA / 2; ++u) { acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0); }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 old[8];
      if (MODE != 0) {
        if (MODE == 1 && !gave_up) {                                     // every wave polls for itself: no workgroup barrier in the step
          int n = 0;
          while (true) {
            const int c = __builtin_amdgcn_readfirstlane(load_l2_int(ctr));
            ++polls;
            if (c >= 4 * j) break;                                        // >=: this workgroup's own waves bump the same counter
            ++not_ready;
            if (++n > POLL_LIMIT || __builtin_amdgcn_readfirstlane(load_l2_int(flags)) != 0) { gave_up = true; if (lane == 0) atomicExch(flags, 1); break; }
            __builtin_amdgcn_s_sleep(4);
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) old[r] = load_l2(tile + r * 256);     // 8 x (64 lanes x 16 B): this wave's quarter of the tile
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NMFMA - NMFMA / 2; ++u) { acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0); }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 0) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]),
                     "+v"(old[7]) :: "memory");
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          f32x4 v = old[r];
          v[0] += mine; v[1] += mine; v[2] += mine; v[3] += mine;
          *reinterpret_cast<f32x4*>(tile + r * 256) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the stores are in the L2
        if (lane == 0) atomicAdd(ctr, 1);                                 // executes in the same L2: 4 j + 4 once all four waves are through
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
  if (r == 123.456f) sink[tid] = r;
  if (lane == 0) {
    atomicAdd(&stats[0], not_ready);
    atomicAdd(&stats[1], polls);
    if (gave_up) atomicAdd(&stats[2], 1ull);
  }
  if (tid == 0) stats[8 + bid] = t1 - t0;
  (void)smem;
}

__global__ void check_kernel(const float* dq, unsigned long long* bad) {
  // tile (x, h, i) must hold sum_{j <= min(i, KT - 1)} (j + 1) in every element
  const size_t n = (size_t)8 * HEADS * NQ * TILE_FLOATS;
  unsigned long long b = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)((e / TILE_FLOATS) % NQ);
    const int m = (i < KT - 1 ? i : KT - 1) + 1;
    b += dq[e] != (float)(m * (m + 1) / 2);
  }
  if (b) atomicAdd(bad, b);
}

struct Bufs { float* dq; int* counters; int* flags; unsigned long long* stats; int* xcc; float* sink; unsigned long long* bad; };

template <int NMFMA, int MODE>
double run(const char* what, const Bufs& b, double floor_ns) {
  constexpr int LDS = 100 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NMFMA, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const size_t n = (size_t)8 * HEADS * NQ * TILE_FLOATS;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(b.dq, 0, n * 4);
    (void)hipMemset(b.counters, 0, (size_t)8 * HEADS * NQ * 4);
    (void)hipMemset(b.flags, 0, 64);
    (void)hipMemset(b.stats, 0, (8 + 256) * 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NMFMA, MODE>), dim3(256), dim3(256), LDS, 0, b.dq, b.counters, b.flags, b.stats, b.xcc, b.sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipMemset(b.bad, 0, 8);
  hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, b.dq, b.bad);
  unsigned long long bad = 0, stats[8 + 256];
  (void)hipMemcpy(&bad, b.bad, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(stats, b.stats, sizeof(stats), hipMemcpyDeviceToHost);
  double first = 0, last = 0;                                            // key tile 0 never waits; key tile 31 waits for everybody
  for (int x = 0; x < 8; ++x) { first += stats[8 + x] / 8.0; last += stats[8 + 8 * (KT - 1) + x] / 8.0; }
  const double tick_ns = 10.0;                                           // wall_clock64: 100 MHz
  const double ns_first = first * tick_ns / (HEADS * NQ), ns_last = last * tick_ns / (HEADS * (NQ - (KT - 1)));
  std::vector<int> xcc(256);
  (void)hipMemcpy(xcc.data(), b.xcc, 256 * 4, hipMemcpyDeviceToHost);
  int placed = 0;
  for (int i = 0; i < 256; ++i) placed += xcc[i] == (i & 7);
  // the longest workgroup (key tile 0) walks HEADS x NQ steps; all steps of all workgroups:
  double steps_total = 0;
  for (int j = 0; j < KT; ++j) steps_total += 8.0 * HEADS * (NQ - j);
  const double ns_step = ms * 1e6 / ((double)HEADS * NQ);
  const double rmw_bytes = MODE ? steps_total * TILE_FLOATS * 4 * 2 : 0, flop = steps_total * 4 * (double)NMFMA * 32768.0;
  printf("%-64s %8.1f ns/step (x%.2f of the MFMA floor; key tile 0: %7.1f, key tile 31: %7.1f ns per own step)  RMW %5.2f TB/s through the L2s  MFMA %5.0f TFLOP/s  sums %s  polls %llu, not ready %llu, gave up %llu  b%%8==XCC_ID %d/256\n",
         what, ns_step, floor_ns > 0 ? ns_step / floor_ns : 1.0, ns_first, ns_last, rmw_bytes / ms / 1e9, flop / ms / 1e9,
         MODE == 0 ? "-" : (bad == 0 ? "ok" : "WRONG"), stats[1], stats[0], stats[2], placed);
  if (bad) printf("    %llu elements differ from the ordered sum%s\n", bad, stats[2] ? " (a workgroup gave up: FAILED, not hung)" : "");
  fflush(stdout);
  return ns_step;
}

int main() {
  Bufs b;
  const size_t n = (size_t)8 * HEADS * NQ * TILE_FLOATS;
  (void)hipMalloc(&b.dq, n * 4); (void)hipMalloc(&b.counters, (size_t)8 * HEADS * NQ * 4); (void)hipMalloc(&b.flags, 64);
  (void)hipMalloc(&b.stats, (8 + 256) * 8); (void)hipMalloc(&b.xcc, 256 * 4); (void)hipMalloc(&b.sink, 4096); (void)hipMalloc(&b.bad, 8);
  run<80, 0>("warm-up", b, 0);
  const double f80 = run<80, 0>("80 MFMA per wave and step (5 units of 64 x 128 x 128), no RMW", b, 0);
  run<80, 2>("80 MFMA + 32 KB RMW per step, no ordering (L2 cost alone)", b, f80);
  run<80, 1>("80 MFMA + ordered 32 KB RMW (the design)", b, f80);
  const double f112 = run<112, 0>("112 MFMA (today's 7 units), no RMW", b, 0);
  run<112, 1>("112 MFMA + ordered RMW", b, f112);
  const double f48 = run<48, 0>("48 MFMA (3 units: a dQ-only kernel's step), no RMW", b, 0);
  run<48, 1>("48 MFMA + ordered RMW", b, f48);
  run<0, 2>("RMW only, no ordering", b, 0);
  run<0, 1>("ordered RMW only (the chain's latency per step)", b, 0);
  return 0;
}
