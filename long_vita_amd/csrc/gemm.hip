// bf16 GEMM on gfx950 MFMA:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), fp32 accumulation.
//
// Bound: MFMA (dense bf16 peak ~2.5 PFLOP/s).  Algorithmic work 2*M*N*K flop.
//
// Structure (one workgroup = 4 waves = one 128x128 output tile, BK = 64):
//   * both operands are K-contiguous, so an A tile and a W tile are each [128][64] bf16 = 16 KiB,
//     staged HBM/L2 -> LDS with the LDS-DMA `global_load_lds_dwordx4` (16 B per lane, no VGPR
//     round trip), double buffered (2 x 32 KiB), one barrier per K tile;
//   * LDS-DMA writes lane-linear (wave base + lane*16 B), so the bank-conflict swizzle is applied
//     to the per-lane *global source* address and mirrored on the ds_read_b128 side:
//         physical 16-B slot = logical slot ^ ((row >> 1) & 7)        (conflict-free for the
//     four 16-lane groups a ds_read_b128 is serviced in on gfx950);
//   * each wave owns a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 blocks; the W fragment is fed
//     as the MFMA "A" operand and the activation fragment as "B", so a lane ends up with four
//     *consecutive output columns* of one row per register quad -> 8-byte vector epilogue
//     (bias / residual / LayerScale loads and the bf16 store are all 4-wide);
//   * workgroup -> tile mapping is XCD-aware: the 8 XCDs (block id % 8) each walk a contiguous
//     range of tiles in grouped (8 tile-rows) order so that concurrently resident workgroups of
//     one XCD share A/W panels in that XCD's private 4 MiB L2.
//
// Replaces torch.matmul / TE linears (see include/vita_hip.h).
#include "vita_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  bf16_t* C; int64_t ldc;
  int64_t M, N, K;        // N = output columns (for SWIGLU: W has 2N rows)
  const bf16_t* bias; const bf16_t* scale;
  const bf16_t* R; int64_t ldr;
  int tiles_m, tiles_n;
  int stagger;            // K-loop start offset policy (see gemm_bf16_kernel)
  int splits = 1;         // r04, TN mode only: the contraction is cut into `splits` ranges of `k_tiles_per_split` K tiles (the last may be
  int k_tiles_per_split = 0;   // shorter); workgroup (split, tile) writes its fp32 partial tile to ((float*)C)[split][M][N]
};
constexpr int VITA_EPI_F32_PARTIAL = 100;       // internal: the split-K epilogue of the TN kernel (never passed through the C ABI)

__device__ __forceinline__ float gelu_tanh(float x) {       // F.gelu(x, approximate="tanh")
  return 0.5f * x * (1.0f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// LDS image of a [rows][64] bf16 tile: 128-byte rows, 16-byte slot s of row r stored at
// slot s ^ ((r >> 1) & 7)  (conflict-free for the 16-lane groups of ds_read_b128).
__device__ __forceinline__ int tile_off(int row, int slot) {
  return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// ---- epilogue arithmetic on four consecutive output columns n..n+3 of row m (shared by all tile kernels) ------------------
// o = the raw fp32 accumulators; the bf16 rounding chains are the reference's unfused ones (include/vita_hip.h)
// pure arithmetic of the non-SwiGLU epilogues: o = raw accumulators -> values to round and store; b / sc / r = bias, LayerScale and
// residual of the same four columns as floats (ignored where the epilogue has none)
template <int EPI>
__device__ __forceinline__ void epilogue_math(float (&o)[4], bool has_bias, const float (&b)[4], const float (&sc)[4],
                                              const float (&r)[4]) {
  if (EPI == VITA_EPI_BIAS2_GELU_TANH || EPI == VITA_EPI_BIAS2_RES || EPI == VITA_EPI_BIAS2_GELU) {      // bias added to the ROUNDED product
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = bf16_round(o[j]);
  }
  if (EPI != VITA_EPI_NONE && has_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += b[j];
  }
  if (EPI == VITA_EPI_BIAS_GELU || EPI == VITA_EPI_BIAS2_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gelu_erf(bf16_round(o[j]));
  }
  if (EPI == VITA_EPI_BIAS2_GELU_TANH) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gelu_tanh(bf16_round(o[j]));
  }
  if (EPI == VITA_EPI_BIAS_SCALE_RES) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = bf16_round(bf16_round(o[j]) * sc[j]);
  }
  if (EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_BIAS2_RES) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = bf16_round(o[j]);
  }
  if (EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_BIAS_SCALE_RES || EPI == VITA_EPI_BIAS2_RES) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += r[j];
  }
}
template <int EPI> constexpr bool epi_has_residual() { return EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_BIAS_SCALE_RES || EPI == VITA_EPI_BIAS2_RES; }
__device__ __forceinline__ void unpack_quad(const u32x2 v, float (&f)[4]) {
  f[0] = bf16lo_to_f32(v[0]); f[1] = bf16hi_to_f32(v[0]); f[2] = bf16lo_to_f32(v[1]); f[3] = bf16hi_to_f32(v[1]);
}

// ---- epilogue on four consecutive output columns n..n+3 of row m, with the tail handling of ragged N (shared by all tile kernels)
template <int EPI>
__device__ __forceinline__ void epilogue_quad(const GemmArgs& p, float (&o)[4], int64_t m, int64_t n) {
  const bool full = n + 3 < p.N;
  float b[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
  const bool has_bias = EPI != VITA_EPI_NONE && p.bias;
  if (has_bias) {
    if (full) {
      unpack_quad(*reinterpret_cast<const u32x2*>(p.bias + n), b);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (n + j < p.N) b[j] = bf16_to_f32(p.bias[n + j]);     // static indices: no scratch
    }
  }
  if (EPI == VITA_EPI_BIAS_SCALE_RES) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = bf16_to_f32(p.scale[(n + j < p.N) ? n + j : p.N - 1]);
  }
  if (epi_has_residual<EPI>()) {
    const bf16_t* rsrc = p.R + m * p.ldr + n;
    if (full) {
      unpack_quad(*reinterpret_cast<const u32x2*>(rsrc), r);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (n + j < p.N) r[j] = bf16_to_f32(rsrc[j]);
    }
  }
  epilogue_math<EPI>(o, has_bias, b, sc, r);
  bf16_t* dst = p.C + m * p.ldc + n;
  if (full) {
    u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(dst) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (n + j < p.N) dst[j] = f32_to_bf16(o[j]);
  }
}

// SwiGLU: g / u = the raw accumulators of the gate and up rows of output columns n..n+3
__device__ __forceinline__ void swiglu_quad(const GemmArgs& p, const float (&gt)[4], const float (&up)[4], int64_t m, int64_t n) {
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g = bf16_round(gt[j]);
    const float u = bf16_round(up[j]);
    const float s = bf16_round(g / (1.0f + __expf(-g)));
    o[j] = s * u;
  }
  bf16_t* dst = p.C + m * p.ldc + n;
  if (n + 3 < p.N) {
    u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(dst) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (n + j < p.N) dst[j] = f32_to_bf16(o[j]);
  }
}

// ---- epilogue of the 32x32-block kernels: lane holds D^T: row m = ... + (lane & 31), columns 8*rg + 4*(lane>>5) + 0..3 --
template <int EPI, int MI, int NI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[NI][MI], int64_t m0, int64_t n0, int wm,
                                              int wn, int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int64_t m = m0 + wm * TM + mi * 32 + (lane & 31);
    if (m >= p.M) continue;
    if (EPI == VITA_EPI_SWIGLU) {
      // the wave's tile rows of W = NI/2 x [gate 32 | up 32]; pair pi -> output columns n0 + (wn*NI/2 + pi)*32 + 0..31
#pragma unroll
      for (int pi = 0; pi < NI / 2; ++pi) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int64_t n = n0 + (wn * (NI / 2) + pi) * 32 + rg * 8 + hi * 4;
          if (n >= p.N) continue;
          float g[4], u[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { g[j] = acc[2 * pi][mi][rg * 4 + j]; u[j] = acc[2 * pi + 1][mi][rg * 4 + j]; }
          swiglu_quad(p, g, u, m, n);
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int64_t n = n0 + wn * TN + ni * 32 + rg * 8 + hi * 4;
          if (n >= p.N) continue;
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = acc[ni][mi][rg * 4 + j];
          epilogue_quad<EPI>(p, o, m, n);
        }
      }
    }
  }
}

// One workgroup = WM x WN waves = one BM x BN output tile; each wave owns (BM/WM) x (BN/WN).
//   <128,128,2,2>: 4 waves, 64 KiB LDS, 2 workgroups per CU (small / ragged problems)
//   <256,256,2,4>: 8 waves, 128 KiB LDS, per-wave tile 128 x 64 -> 0.75 LDS fragment reads and
//                  0.25 KiB of DMA per MFMA instead of 1.0 / 0.5 (the 128^2 tile is LDS-bound)
template <int EPI, int BM, int BN, int WM, int WN, bool PIN = true>
__global__ __launch_bounds__(64 * WM * WN, (BM == 256 && WM * WN == 4) ? 1 : 2) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;          // per-wave tile
  constexpr int MI = TM / 32, NI = TN / 32;          // 32x32 MFMA blocks per wave
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int QA = BM / 8 / NW, QW = BN / 8 / NW;  // 1-KiB DMA pieces per wave per operand
  static_assert(EPI != VITA_EPI_SWIGLU || NI % 2 == 0, "SwiGLU epilogue pairs 32-column blocks (gate, up) of a wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware, grouped tile order -------------------------------------------------------
  const int nwg = p.tiles_m * p.tiles_n;
  int pid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = (BM == 256) ? 4 : 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = pid - group * per_group;
  const int tm = first_m + in_group % gsz;
  const int tn = in_group / gsz;

  const int64_t m0 = (int64_t)tm * BM;
  // output-column origin of this tile: SWIGLU packs BN/2 outputs (gate+up) per BN W rows
  const int64_t n0 = (int64_t)tn * (EPI == VITA_EPI_SWIGLU ? BN / 2 : BN);

  // ---- DMA source pointers (advanced by BK per K tile) and LDS piece offsets ------------------
  // a wave-instruction moves 8 tile rows (1 KiB); lane -> (row rbase + lane/8, physical slot lane&7),
  // which must receive logical slot ps ^ ((row>>1)&7): the swizzle is applied on the SOURCE side.
  const bf16_t* a_src[QA];
  const bf16_t* w_src[QW];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int lr = (wave * QA + q) * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((lr >> 1) & 7);
    int64_t g = m0 + lr;
    g = g < p.M ? g : p.M - 1;
    a_src[q] = p.A + g * p.lda + ls * 8;
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int lr = (wave * QW + q) * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((lr >> 1) & 7);
    int64_t g;
    if (EPI == VITA_EPI_SWIGLU) {
      // tile row lr: blk = lr / 32; blk&1 = 0 gate / 1 up; output column = n0 + (blk>>1)*32 + lr%32
      const int blk = lr >> 5;
      int64_t oc = n0 + (blk >> 1) * 32 + (lr & 31);
      oc = oc < p.N ? oc : p.N - 1;
      g = ((blk & 1) ? p.N : 0) + oc;
    } else {
      g = n0 + lr;
      g = g < p.N ? g : p.N - 1;
    }
    w_src[q] = p.W + g * p.ldw + ls * 8;
  }
  auto stage = [&](unsigned sl, int adv) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)a_src[q], (lds_void*)(uintptr_t)(sl + (wave * QA + q) * 1024), 16, 0, 0);
      a_src[q] += adv;
    }
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)w_src[q], (lds_void*)(uintptr_t)(sl + A_BYTES + (wave * QW + q) * 1024), 16, 0, 0);
      w_src[q] += adv;
    }
  };

  // ---- per-lane fragment offsets: rows wm*TM + l31 (+32*mi as immediates), slot 2*kk + (lane>>5) ----
  unsigned fa[4], fw[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fa[kk] = tile_off(wm * TM + (lane & 31), kk * 2 + (lane >> 5));
    fw[kk] = A_BYTES + tile_off(wn * TN + (lane & 31), kk * 2 + (lane >> 5));
  }

  f32x16 acc[NI][MI];  // [ni][mi]: 32 output columns x 32 rows each (transposed MFMA output)
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)(p.K / BK);
  // measurement aid (EPI 0 only, VITA_GEMM_EXP=10): ldr = -1 makes every K tile re-read tile 0, so all loads hit L1/L2
  const int kstep = (EPI == VITA_EPI_NONE && p.ldr == -1) ? 0 : BK;
  // Staggered K start: workgroups that run at the same time walk K from different offsets (wrapping around), so that
  // they do not all ask the memory system for the same K columns — the same few channels — at the same moment.
  //   1 = by XCD (the workgroups of one XCD stay in phase and keep sharing their operands through that XCD's L2)
  //   2 = by M tile, 3 = by tile id (two K tiles apart, modulo 32)
  int kt = 0;
  if (p.stagger == 1) kt = (int)(((int64_t)(blockIdx.x & 7) * nk) >> 3);
  else if (p.stagger == 2) kt = ((tm & 31) * 2) % nk;
  else if (p.stagger == 3) kt = ((pid & 31) * 2) % nk;
  kt = __builtin_amdgcn_readfirstlane(kt);
  if (kt && kstep) {
#pragma unroll
    for (int q = 0; q < QA; ++q) a_src[q] += (int64_t)kt * BK;
#pragma unroll
    for (int q = 0; q < QW; ++q) w_src[q] += (int64_t)kt * BK;
  }
  auto next_adv = [&]() __attribute__((always_inline)) {     // pointer step after staging tile kt; wraps at the end of K
    int adv = kstep;
    if (++kt == nk) { kt = 0; adv = kstep ? kstep - (int)p.K : 0; }
    return adv;
  };
  stage(lds0, next_adv());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const unsigned cur = lds0 + (t & 1) * STAGE;
    if (t + 1 < nk) stage(lds0 + ((t + 1) & 1) * STAGE, next_adv());
    bf16x8 af[4][MI], wf[4][NI];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[kk][mi] = *(lds_bf16x8*)(uintptr_t)(cur + fa[kk] + mi * 32 * 128);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[kk][ni] = *(lds_bf16x8*)(uintptr_t)(cur + fw[kk] + ni * 32 * 128);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][ni], af[kk][mi], acc[ni][mi], 0, 0, 0);
    if (PIN) {
      // issue order: fragments of k-step s+1 are read while the MFMAs of k-step s run
      constexpr int NF = MI + NI, NM = MI * NI;
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            if (i < NM) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  gemm_epilogue<EPI, MI, NI, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// ---- 4 waves x (128 x 128): one wave per SIMD, operands HBM/L2 -> VGPR -> LDS ------------------------------------------
// Per MFMA this geometry moves 0.5 KiB of fragment reads + 0.25 KiB of staging writes through the LDS port (the 8-wave
// kernel above: 0.75 + 0.25), and the port is what bounds these kernels.  With one wave per SIMD nothing hides a stall,
// so program order is the issue order (sched_group_barrier pins the MFMAs between the memory instructions):
// (slot plan above the main loop).
// MODE (measurement only, EPI 0; results are wrong for MODE > 0): 1 = every K tile re-reads tile 0 (all loads hit L1/L2),
// 2 = no global loads / LDS writes, 3 = no fragment reads either (MFMA stream only) — the ablation ladder in DESIGN.md §4.2
template <int EPI, int MODE = 0>
__global__ __launch_bounds__(256, 1) void gemm4_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, WN = 2, TM = 128, TN = 128, MI = 4, NI = 4;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int QA = 8, QW = 8, NP = QA + QW, NF = MI + NI, NM = MI * NI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  constexpr bool SPLITK = EPI == VITA_EPI_F32_PARTIAL;
  const int nwg = p.tiles_m * p.tiles_n * (SPLITK ? p.splits : 1);
  int pid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // split-K: the split is the SLOW index, so that the workgroups an XCD runs side by side are tiles of one K range sharing operand panels
  const int split = SPLITK ? pid / (p.tiles_m * p.tiles_n) : 0;
  if (SPLITK) pid -= split * (p.tiles_m * p.tiles_n);
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = pid - group * per_group;
  const int tm = first_m + in_group % gsz;
  const int tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * (EPI == VITA_EPI_SWIGLU ? BN / 2 : BN);

  // piece j of a wave = 8 tile rows x 128 B: lane -> (row (wave*8 + j)*8 + lane/8, 16-B slot lane&7), stored swizzled
  const bf16_t* src[NP];
  unsigned dst[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const bool isw = j >= QA;
    const int lr = (wave * 8 + (isw ? j - QA : j)) * 8 + (lane >> 3);
    int64_t g;
    if (!isw) {
      g = m0 + lr;
      g = g < p.M ? g : p.M - 1;
      src[j] = p.A + g * p.lda + (lane & 7) * 8;
    } else {
      if (EPI == VITA_EPI_SWIGLU) {
        const int blk = lr >> 5;
        int64_t oc = n0 + (blk >> 1) * 32 + (lr & 31);
        oc = oc < p.N ? oc : p.N - 1;
        g = ((blk & 1) ? p.N : 0) + oc;
      } else {
        g = n0 + lr;
        g = g < p.N ? g : p.N - 1;
      }
      src[j] = p.W + g * p.ldw + (lane & 7) * 8;
    }
    dst[j] = (isw ? A_BYTES : 0) + tile_off(lr, lane & 7);
  }
  unsigned fa[4], fw[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fa[kk] = tile_off(wm * TM + (lane & 31), kk * 2 + (lane >> 5));
    fw[kk] = A_BYTES + tile_off(wn * TN + (lane & 31), kk * 2 + (lane >> 5));
  }
  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)(p.K / BK);
  u32x4 g[NP];
  bf16x8 af[2][MI], wf[2][NI];
  auto read_frag = [&](unsigned base, int kk, int set, int i) __attribute__((always_inline)) {
    if (i == 0) wf[set][0] = *(lds_bf16x8*)(uintptr_t)(base + fw[kk]);
    else if (i <= MI) af[set][i - 1] = *(lds_bf16x8*)(uintptr_t)(base + fa[kk] + (i - 1) * 32 * 128);
    else wf[set][i - MI] = *(lds_bf16x8*)(uintptr_t)(base + fw[kk] + (i - MI) * 32 * 128);
  };
  // prologue: tile 0 -> stage 0, tile 1 -> registers, first fragments
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    g[j] = *(const u32x4*)src[j];
    src[j] += (nk > 1 && MODE == 0) ? BK : 0;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(lds0 + dst[j]) = g[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    g[j] = *(const u32x4*)src[j];
    src[j] += (nk > 2 && MODE == 0) ? BK : 0;
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) read_frag(lds0, 0, 0, i);
  if (MODE == 3) {
#pragma unroll
    for (int i = 0; i < NF; ++i) read_frag(lds0, 1, 1, i);
  }

  // Every MFMA has exactly one memory instruction behind it (64 + 64 per K tile):
  //   positions 0..7 of k-steps 0..2 : fragment reads of the next k-step
  //   positions 8..15 of k-steps 0, 1 : ds_write_b128 of tile t+1 (in registers since the previous iteration) -> other stage
  //   positions 8..15 of k-step 2, 0..7 of k-step 3 : global_load_dwordx4 of tile t+2
  //   position 8 of k-step 3        : lgkmcnt(0) + s_barrier (no vmcnt: the loads just issued stay in flight)
  //   positions 8..15 of k-step 3   : first fragment reads of tile t+1
  for (int t = 0; t < nk; ++t) {
    const unsigned cur = lds0 + (t & 1) * STAGE, nxt = lds0 + ((t + 1) & 1) * STAGE;
    const int adv = (MODE == 0 && t + 3 < nk) ? BK : 0;   // the pointers stop at the last K tile (re-staged, never read)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int i = ni * MI + mi;
          if (kk == 3 && i == NM / 2) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][ni], af[kk & 1][mi], acc[ni][mi], 0, 0, 0);
          if (kk == 3 && i >= NM / 2) __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
          else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (MODE < 3 && kk < 3 && i < NF) {
            read_frag(cur, kk + 1, (kk + 1) & 1, i);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (MODE < 3 && kk == 3 && i >= NM / 2) {
            read_frag(nxt, 0, 0, i - NM / 2);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
          }
          if (MODE >= 2) continue;
          if (kk < 2 && i >= NM / 2) {
            const int j = kk * 8 + i - NM / 2;
            *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(nxt + dst[j]) = g[j];
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          if ((kk == 2 && i >= NM / 2) || (kk == 3 && i < NM / 2)) {
            const int j = kk == 2 ? i - NM / 2 : 8 + i;
            g[j] = *(const u32x4*)src[j];
            src[j] += adv;
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        }
    }
  }
  gemm_epilogue<EPI, MI, NI, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

template <int EPI, int MODE = 0>
int launch_gemm4(GemmArgs a, hipStream_t st) {
  constexpr int lds = 2 * (256 + 256) * BK * 2;
  const int64_t tm = (a.M + 255) / 256;
  const int64_t bn_out = EPI == VITA_EPI_SWIGLU ? 128 : 256;
  const int64_t tn = (a.N + bn_out - 1) / bn_out;
  if (tm * tn > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<EPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL((gemm4_kernel<EPI, MODE>), dim3((unsigned)(tm * tn)), dim3(256), lds, st, a);
  return vita_check_launch();
}

// ---- 4 waves x (128 x 128), v_mfma_f32_16x16x32_bf16, accumulators pinned in AGPRs: the large-problem kernel (round 2) ------------
// Same macro tile as <256,256,2,4>, one wave per SIMD.  What makes it ~20 % faster than the 8-wave kernel (tools/hwprobe/gemmd.hip is
// the stand-alone version with its schedule sweep):
//   * the 64 accumulator blocks (256 registers) stay in AGPRs for the whole K loop — the MFMAs are inline asm ("+a") — and the 32
//     operand fragments of a K tile sit in VGPRs: nothing ever moves between the two halves of the register file;
//   * LDS layout "interleaved rows, padded lines": line (h, r16) = the eight tile rows 128 h + 16 rb + r16 (rb = 0..7), 128 B each,
//     + 16 B of padding.  One LDS-DMA instruction (buffer_load_dwordx4 ... lds, lane -> rb = lane / 8, 16-B chunk lane % 8) fills
//     exactly one line from whole 128-B row segments, and the 16x16x32 fragment read (lane -> row lane % 16, k chunk lane / 16)
//     strides 1040 B = 260 dwords between lanes: conflict-free, with row block and k half as address constants;
//   * two stages with prefetch distance two: the DMA of tile t+2 goes into the stage tile t is being read from as soon as every wave
//     holds its second-half fragments (barrier in the first half of the iteration), so a piece has about 1.5 iterations to land;
//   * the loop is one fixed instruction order (asm volatile statements + sched_barrier): MFMA slots 0..127, a fragment read behind
//     every second MFMA of slots 0..30, lgkmcnt(0) + s_barrier at 36, one DMA piece every fourth slot 40..100 (a burst costs 6 %),
//     vmcnt + s_barrier at 103, the next tile's 16 first-half reads behind slots 104..119.
// Ragged M / N: source rows are clamped (their products land in rows / columns the epilogue does not store).
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F& f) {            // f(std::integral_constant<int, B>) ... f(<E - 1>), in order
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
namespace w4 {
constexpr int LINE = 1040, HALF = 16 * LINE, OPB = 2 * HALF, STAGE = 2 * OPB, LDS_BYTES = 2 * STAGE;   // 133120 B
// the operand-split schedule's LDS-DMA slots: piece j (0 .. 7: A lines, 8 .. 15: W lines) is issued behind MFMA slot PIECE_SLOT[variant][j];
// A pieces need slot > 21 (barrier 1), W pieces slot > 51 (barrier 2).  Variant 0: the vendor kernel's positions (three bursts of five, every
// third slot); variant 1: the same window, evenly spaced (every 6 - 7 slots).
constexpr int PIECE_SLOT[2][16] = {{22, 25, 28, 31, 34, 52, 55, 58, 61, 64, 85, 87, 89, 96, 100, 124},
                                   {22, 28, 34, 40, 46, 52, 58, 65, 71, 78, 84, 91, 97, 104, 110, 117}};
constexpr int piece_at(int variant, int slot) {
  for (int j = 0; j < 16; ++j) if (PIECE_SLOT[variant][j] == slot) return j;
  return -1;
}
constexpr int pieces_before(int variant, int slot) {
  int n = 0;
  for (int j = 0; j < 16; ++j) n += PIECE_SLOT[variant][j] < slot;
  return n;
}
// what the schedule relies on: A pieces (0 .. 7) behind barrier 1 (slot 21), W pieces (8 .. 15) behind barrier 2 (slot 51), increasing slots (the
// vmcnt of barrier 3 counts the pieces in front of it), and room for the three slots a staggered wave adds
constexpr bool piece_table_ok(int variant) {
  for (int j = 0; j < 16; ++j) {
    const int sl = PIECE_SLOT[variant][j];
    if (sl <= (j < 8 ? 21 : 51) || sl + 3 > 127) return false;
    if (j > 0 && sl <= PIECE_SLOT[variant][j - 1]) return false;
  }
  return true;
}
static_assert(piece_table_ok(0) && piece_table_ok(1), "LDS-DMA piece slots of the operand-split schedule");
}

// INTERIOR = every tile of the problem is whole (M % 256 == 0, N % tile width == 0): chosen at launch, so that each instantiation has
// ONE straight-line epilogue (two epilogue paths behind a run-time branch made hipcc shuffle the pinned accumulators between AGPRs
// and produced wrong blocks).
//
// TN = true (r03, vita_gemm_bf16_tn): BOTH operands are stored contraction-major — A_t [K][M] and W_t [K][N], C = A_t^T W_t — which is
// what a weight gradient is (dW [N, K_in] = dY^T X with dY [tokens, N] and X [tokens, K_in] as the forward left them), so wgrad needs
// no vita_transpose_bf16 passes.  Same MFMA order, barriers, DMA slots and epilogue; what changes is the LDS image and the fragment
// reads: a K tile of an operand is 64 contraction rows x 256 columns = 64 rows of 512 B (DMA piece = two whole rows), the 16 x 32
// fragment (lane -> column lane % 16, contraction rows 8 (lane / 16) .. + 7) comes from TWO ds_read_b64_tr_b16 (rows .. + 0..3 and
// .. + 4..7; the hardware transposes 4 x 16 blocks inside 16-lane groups), and the eight rows a 32-lane half reads (r and r + 8,
// r = 0..3) are spread over the 64 banks by XOR-ing the 32-byte chunk index with key(row) = (row & 3) | ((row >> 3) & 1) << 2 — on
// the DMA's SOURCE address and on the read address.
//
// OPM = 2 (r04, vita_gemm_bf16_nn): A as in the NT kernel ([M][K], row-major), W contraction-major ([K][N]) as in the TN kernel —
// C = A W, which is what an input gradient is (dX [tokens, K_in] = dY [tokens, N] W [N, K_in] with the weight as the forward holds it),
// so dgrad needs no vita_transpose_bf16 pass over the weight either.  A's half of a stage is the NT image, W's half the TN image.
#ifndef VITA_GEMM_EARLY_NEXT
#define VITA_GEMM_EARLY_NEXT 1
#endif
#ifndef VITA_GEMM_DMA_STEP
#define VITA_GEMM_DMA_STEP 4       // two-barrier schedule: one LDS-DMA piece every DMA_STEP slots from slot 40 on (5: a same-box A / B of the window's width alone)
#endif
#ifndef VITA_GEMM_DMA_STEP_OTHER
#define VITA_GEMM_DMA_STEP_OTHER 5 // the TN / NN / split-K modes (weight and input gradients): one piece every DMA_STEP_OTHER slots from slot 40 on
#endif
#ifndef VITA_GEMM_RD_STEP
#define VITA_GEMM_RD_STEP 2        // two-barrier schedule: 1 = the 16 second-half reads in consecutive slots, barrier 1 at slot 20, pieces from slot 22
#endif
#ifndef VITA_GEMM_SPLIT_TN
#define VITA_GEMM_SPLIT_TN 1       // the operand-split schedule in the TN mode (weight gradients, also split-K) too: -1 to -2.4 % same box; NN (input gradients) measured +0.3 to +1 % with it and keeps two barriers
#endif
#ifndef VITA_GEMM_WAVE_STAGGER
#define VITA_GEMM_WAVE_STAGGER 0   // 1: the operand-split loop in four copies, wave w's LDS-DMA pieces w slots behind the table (same-box A / B)
#endif
#ifndef VITA_GEMM_SWIGLU_STEP5
#define VITA_GEMM_SWIGLU_STEP5 1    // 1: fc1 + SwiGLU keeps the two-barrier schedule, with one piece every 5 slots (the variant it measured best with)
#endif
#ifndef VITA_GEMM_SCHED
#define VITA_GEMM_SCHED 1          // 1: operand-split release of the stage, three barriers (r06); 0: the r05 two-barrier schedule (same-box A / B builds)
#endif
template <int EPI, bool INTERIOR, int OPM>
__device__ __forceinline__ void gemm_w4_tile(const GemmArgs& p, const int bid_) {
  using namespace w4;
  constexpr bool EARLY_NEXT = VITA_GEMM_EARLY_NEXT != 0;
  // two-barrier schedule, NT one-pass kernels only (the other modes keep the r05 positions): reads of the second-half fragments every RD_STEP
  // slots from slot 0, barrier 1 at BAR1, one LDS-DMA piece every DMA_STEP slots from DMA0
  constexpr bool TUNED = OPM == 0 && EPI != VITA_EPI_F32_PARTIAL;
  constexpr int DMA_STEP = TUNED ? ((VITA_GEMM_SWIGLU_STEP5 && EPI == VITA_EPI_SWIGLU) ? 5 : VITA_GEMM_DMA_STEP) : VITA_GEMM_DMA_STEP_OTHER, RD_STEP = TUNED ? VITA_GEMM_RD_STEP : 2;
  constexpr int BAR1 = RD_STEP == 2 ? 36 : 20, DMA0 = RD_STEP == 2 ? 40 : 22;
  static_assert(DMA0 + 15 * DMA_STEP < 128 && (RD_STEP == 1 || RD_STEP == 2), "pieces must fit the tile");      // (0: the r02 - r04 placement, kept for same-box A / B builds)
  constexpr bool TN = OPM == 1, TA = OPM == 1, TW = OPM != 0;          // TN: both operands contraction-major; TA / TW: per operand
  constexpr int STG = TN ? 65536 : STAGE, OPBS = TA ? 32768 : OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  constexpr bool SPLITK = EPI == VITA_EPI_F32_PARTIAL;
  const int nwg = p.tiles_m * p.tiles_n * (SPLITK ? p.splits : 1);
  int pid;
  {
    const int bid = bid_, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // split-K: the split is the SLOW index, so that the workgroups an XCD runs side by side are tiles of one K range sharing operand panels
  const int split = SPLITK ? pid / (p.tiles_m * p.tiles_n) : 0;
  if (SPLITK) pid -= split * (p.tiles_m * p.tiles_n);
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = pid - group * per_group;
  const int tm = first_m + in_group % gsz;
  const int tn = in_group / gsz;
  const int64_t m0 = (int64_t)tm * 256;
  const int64_t n0 = (int64_t)tn * (EPI == VITA_EPI_SWIGLU ? 128 : 256);
  const int nk_all = (int)(p.K / BK);
  const int nk = SPLITK ? min(p.k_tiles_per_split, nk_all - split * p.k_tiles_per_split) : nk_all;

  // ---- DMA geometry: wave w fills lines (h = w >> 1, r16 = (w & 1) * 8 + i), i = 0..7, of both operands; per-lane byte offsets of
  // the eight pieces relative to the tile's first row (clamped rows for ragged edges; SwiGLU: 16-row blocks alternate gate / up) ------
  const int h = wave >> 1, r0 = (wave & 1) * 8;
  const unsigned d_line0_a = TA ? (unsigned)(wave * 8 * 1024) : (unsigned)(h * HALF + r0 * LINE);
  const unsigned d_line0_w = TW ? (unsigned)(wave * 8 * 1024) : (unsigned)(h * HALF + r0 * LINE);
  constexpr int PIECE_A = TA ? 1024 : LINE, PIECE_W = TW ? 1024 : LINE;
  unsigned voff_a[8], voff_w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // contraction-major operand: piece 8 wave + i = contraction rows 16 wave + 2 i + {0, 1}; lane -> (row lane / 32, 16-byte chunk lane % 32)
    const int krow = 16 * wave + 2 * i + (lane >> 5), c16 = lane & 31;
    const int key = (krow & 3) | (((krow >> 3) & 1) << 2);         // the 8 rows one 32-lane half reads (r, r + 8; r = 0..3 [+ 4]) get 8 keys
    const int col = ((((c16 >> 1) ^ key) << 1) | (c16 & 1)) * 8;                            // source-side swizzle (32-byte chunks)
    const int lr = 128 * h + 16 * (lane >> 3) + r0 + i;
    if (TA) {
      voff_a[i] = (unsigned)((krow * p.lda + col) * 2);
    } else {
      int64_t g = m0 + lr;
      g = g < p.M ? g : p.M - 1;
      voff_a[i] = (unsigned)((g - m0) * p.lda * 2 + (lane & 7) * 16);
    }
    if (TW) {
      voff_w[i] = (unsigned)((krow * p.ldw + col) * 2);
      continue;
    }
    int64_t g;
    if (EPI == VITA_EPI_SWIGLU) {
      const int blk = lr >> 4;
      int64_t oc = n0 + (blk >> 1) * 16 + (lr & 15);
      oc = oc < p.N ? oc : p.N - 1;
      g = ((blk & 1) ? p.N : 0) + oc;
    } else {
      g = n0 + lr;
      g = g < p.N ? g : p.N - 1;
    }
    voff_w[i] = (unsigned)((g - n0) * p.ldw * 2 + (lane & 7) * 16);
  }
  // the descriptor base must stay in SGPRs (a VGPR descriptor costs a waterfall loop per DMA): uniform halves, a scalar K offset
  auto uniform_ptr = [](const void* q) __attribute__((always_inline)) {
    const uint64_t v = (uint64_t)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
  };
  const char* const ap = uniform_ptr(TA ? p.A + m0 : p.A + m0 * p.lda);
  const char* const wp = uniform_ptr(TW ? p.W + n0 : p.W + n0 * p.ldw);
  const int64_t kstep_a = TA ? (int64_t)BK * p.lda * 2 : BK * 2, kstep_w = TW ? (int64_t)BK * p.ldw * 2 : BK * 2;
  // developer aid VITA_GEMM_STAGGER=1 (NT, no split): XCD x starts its K loop at tile x nk / 8 and wraps, so that the eight XCDs do not
  // pull the same K range of their panels through the fabric at the same time (the accumulation order of a tile then depends on its XCD)
  int kt_fetch = (OPM == 0 && !SPLITK && p.stagger == 1) ? (int)(((int64_t)(bid_ & 7) * nk) >> 3) : 0;
  int64_t koff_a = SPLITK ? (int64_t)split * p.k_tiles_per_split * kstep_a : kt_fetch * kstep_a;     // byte offsets of the K tile fetched next
  int64_t koff_w = SPLITK ? (int64_t)split * p.k_tiles_per_split * kstep_w : kt_fetch * kstep_w;
  auto dma_piece = [&](unsigned stage, int j) __attribute__((always_inline)) {              // j = 0..7: A lines, 8..15: W lines
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(j < 8 ? ap + koff_a : wp + koff_w), 0, 0x7fffffff, 0x00020000);
    const int i = j & 7;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(uintptr_t)(stage + (j < 8 ? d_line0_a + i * PIECE_A : OPBS + d_line0_w + i * PIECE_W)), 16,
                                             j < 8 ? voff_a[i] : voff_w[i], 0, 0, 0);
  };
  auto next_tile = [&]() __attribute__((always_inline)) {
    koff_a += kstep_a; koff_w += kstep_w;
    if (OPM == 0 && !SPLITK && ++kt_fetch == nk) { kt_fetch = 0; koff_a = 0; koff_w = 0; }
  };

  // ---- fragment reads: lane -> (row lane % 16) * LINE + (k chunk lane / 16) * 16, + row block * 128 + k half * 64 ------------------
  const unsigned rd_a = (unsigned)(wm * HALF + (lane & 15) * LINE + (lane >> 4) * 16);
  const unsigned rd_w = (unsigned)(OPB + wn * HALF + (lane & 15) * LINE + (lane >> 4) * 16);
  bf16x8 af[2][8], wf[2][8];
  // TN: lane -> (16-lane group g = contraction rows 8 g .., row inside the 4 x 16 block (lane % 16) / 4, 4-column chunk lane % 4);
  // block x of an operand sits in 32-byte chunk (8 wm + x) ^ (row & 7) of each 512-byte row
  u32x2 tlo[16], thi[16];                                                  // the two halves of a fragment until its wait has passed
  unsigned toff_a[8], toff_w[8];
  if (TW) {
    const int i16 = lane & 15, sw = (i16 >> 2) | (((lane >> 4) & 1) << 2);          // = key(row) of both reads of this lane
    const unsigned lane_base = (unsigned)((lane >> 4) * 8 * 512 + (i16 >> 2) * 512 + (i16 & 3) * 8);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      toff_a[x] = lane_base + (unsigned)((wm * 8 + (x ^ sw)) * 32);
      toff_w[x] = (unsigned)OPBS + lane_base + (unsigned)((wn * 8 + (x ^ sw)) * 32);
    }
  }
  // read q (0..15) of k half ks: W block 0 first, then the eight A blocks, then W blocks 1..7 (the order the MFMAs need them)
  // (r05 late: the plain reads take a per-tile base VGPR — stage + rd_a / rd_w, `StageBases` — and an IMMEDIATE offset; before, every one of the
  // 32 reads of a K tile cost an s_add + v_add for its address: VALU 1.28 -> 1.03, SALU 0.45 -> 0.33 instructions per MFMA, the vendor kernel's
  // figures in profiles/r05_gemm_vs_vendor_pmc.txt.  "i" and not "n": the offset is a constant only after inlining and unrolling.)
  struct StageBases { unsigned stage, a, w; };
  auto bases_of = [&](unsigned stage) __attribute__((always_inline)) {
    StageBases b = {stage, stage + rd_a, stage + rd_w};
    asm volatile("" : "+v"(b.a), "+v"(b.w));                                // computed HERE, once, and kept
    return b;
  };
  auto frag_read = [&](const StageBases& sb, int ks, int q) __attribute__((always_inline)) {
    const unsigned stage = sb.stage;
    const bool is_w = q == 0 || q > 8;
    if (is_w ? TW : TA) {                                                  // this operand is contraction-major: transposed reads
      const int x = is_w ? (q == 0 ? 0 : q - 8) : q - 1;
      const unsigned a1 = stage + (is_w ? toff_w[x] : toff_a[x]);
      if (ks == 0) {                                                       // rows 32 ks + 8 g + {0..3}, then + {4..7} (same key)
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tlo[q]) : "v"(a1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(thi[q]) : "v"(a1));
      } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(tlo[q]) : "v"(a1));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(thi[q]) : "v"(a1));
      }
      return;
    }
    if (q == 0 || q > 8) {
      const int nb = q == 0 ? 0 : q - 8;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[ks][nb]) : "v"(sb.w), "i"(nb * 128 + ks * 64));
    } else {
      const int mb = q - 1;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[ks][mb]) : "v"(sb.a), "i"(mb * 128 + ks * 64));
    }
  };
  // TN: after the s_waitcnt that covers them, the 16 fragments of k half ks are assembled from their halves (the empty asm makes the
  // halves opaque HERE, so no compiler-placed copy of them can sit above the wait)
  auto frag_commit = [&](int ks) __attribute__((always_inline)) {
    if (!TW) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (!((q == 0 || q > 8) ? TW : TA)) continue;                        // (OPM = 2: only the W fragments come in halves)
      asm volatile("" : "+v"(tlo[q]), "+v"(thi[q]));
      const u32x4 f = {tlo[q][0], tlo[q][1], thi[q][0], thi[q][1]};
      if (q == 0 || q > 8) wf[ks][q == 0 ? 0 : q - 8] = __builtin_bit_cast(bf16x8, f);
      else af[ks][q - 1] = __builtin_bit_cast(bf16x8, f);
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
    for (int j = 0; j < 16; ++j) dma_piece(lds0 + STG, j);
    next_tile();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  {
    const StageBases b0 = bases_of(lds0);
#pragma unroll
    for (int q = 0; q < 16; ++q) frag_read(b0, 0, q);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  frag_commit(0);

  // ---- r06: the operand-split schedule (VITA_GEMM_SCHED 1; NT mode, one pass over K).  The 2-barrier schedule below frees `cur` for the DMA of
  // tile t + 2 only once ALL 16 second-half fragments are in registers (slot 36), which squeezes the 16 pieces of a tile — 64 KiB per CU — into
  // slots 40 .. 100.  Here the stage is released one OPERAND at a time: the eight second-half A fragments are read first (slots 0 .. 14; their
  // first-half twins were read in the previous tile's tail), a barrier at slot 21 frees A's half of the stage and A's pieces start at slot 22; the W
  // fragments follow (slots 24 .. 42), a second barrier at 51 frees W's half; the barrier that opens tile t + 1's stage sits at slot 92 with
  // vmcnt(13) — the 13 pieces of tile t + 2 issued by then stay in flight — and tile t + 1's first-half fragments are read in slots 93 .. 123.
  // The pieces go out in three bursts of five (every third slot) with 18 .. 21 slots of nothing between them, over slots 22 .. 124: the slot
  // positions are the ones the vendor library's MT256x256x64 kernel uses (read off its disassembly: profiles/r06_gemm_schedules.txt).
  constexpr bool SPLIT_SCHED = VITA_GEMM_SCHED >= 1 && ((OPM == 0 && !SPLITK) || (VITA_GEMM_SPLIT_TN && OPM == 1)) && !(VITA_GEMM_SWIGLU_STEP5 && EPI == VITA_EPI_SWIGLU);
  constexpr int SPLIT_VARIANT = VITA_GEMM_SCHED >= 2 ? 1 : 0;
  constexpr bool PACED_READS = VITA_GEMM_SCHED >= 3;       // 3: fragment reads every other slot (the two-barrier schedule's pace) instead of the vendor's positions
  auto tile_split = [&](auto W_, const bool DMA, const bool NEXT, unsigned cur, unsigned nxt) __attribute__((always_inline)) {
    constexpr int WOFF = decltype(W_)::value;      // VITA_GEMM_WAVE_STAGGER: this wave's pieces go out WOFF slots behind the table's positions
    const StageBases bcur = bases_of(cur), bnxt = bases_of(NEXT ? nxt : cur);
    // (the slot number is a template constant — std::integral_constant through a generic lambda — so that every fragment / piece index below is a
    // constant expression: the "i" operands of the fragment reads need that, and a run-time `s` left them to the unroller's mercy)
    auto slot = [&](auto S_) __attribute__((always_inline)) {
      constexpr int s = decltype(S_)::value;
      constexpr int ks = s >> 6, nb = (s >> 3) & 7, mb = s & 7;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nb][mb]) : "v"(wf[ks][nb]), "v"(af[ks][mb]));
      // second-half fragments of `cur`: A blocks 0 .. 7 (q = 1 .. 8), then W blocks 0 .. 7 (q = 0, 9 .. 15)
      if constexpr (s <= 14 && (s & 1) == 0) frag_read(bcur, 1, 1 + (s >> 1));
      if constexpr (!PACED_READS) {
        if constexpr (s == 24) frag_read(bcur, 1, 0);
        if constexpr (s == 27 || s == 30 || s == 33 || s == 36) frag_read(bcur, 1, 9 + (s - 27) / 3);
        if constexpr (s == 38 || s == 40 || s == 42) frag_read(bcur, 1, 13 + (s - 38) / 2);
      } else {                                       // every other slot, 24 .. 38
        if constexpr (s == 24) frag_read(bcur, 1, 0);
        if constexpr (s >= 26 && s <= 38 && (s & 1) == 0) frag_read(bcur, 1, 9 + (s - 26) / 2);
      }
      if constexpr (s == 21 || s == 51) {
        if (DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (DMA) {
        constexpr int pj = w4::piece_at(SPLIT_VARIANT, s - WOFF);
        if constexpr (pj >= 0) dma_piece(cur, pj);
      }
      if constexpr (s == 92) {
        if (NEXT) {
          if (DMA) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(w4::pieces_before(SPLIT_VARIANT, 93 - WOFF)) : "memory");   // (a piece AT slot 92 is issued in front of this wait)
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
      if (NEXT) {                                   // first-half fragments of `nxt`: A blocks, then W blocks
        if constexpr (!PACED_READS) {
          if constexpr (s == 93 || s == 94 || s == 95) frag_read(bnxt, 0, 1 + (s - 93));
          if constexpr (s == 97 || s == 98) frag_read(bnxt, 0, 4 + (s - 97));
          if constexpr (s == 102 || s == 103 || s == 104) frag_read(bnxt, 0, 6 + (s - 102));
          if constexpr (s == 105) frag_read(bnxt, 0, 0);
          if constexpr (s == 106) frag_read(bnxt, 0, 9);
          if constexpr (s == 109 || s == 112) frag_read(bnxt, 0, 10 + (s - 109) / 3);
          if constexpr (s == 114) frag_read(bnxt, 0, 12);
          if constexpr (s == 117 || s == 120 || s == 123) frag_read(bnxt, 0, 13 + (s - 117) / 3);
        } else {                                     // every other slot, 93 .. 123
          if constexpr (s >= 93 && s <= 107 && (s & 1) == 1) frag_read(bnxt, 0, 1 + (s - 93) / 2);
          if constexpr (s == 109) frag_read(bnxt, 0, 0);
          if constexpr (s >= 111 && s <= 123 && (s & 1) == 1) frag_read(bnxt, 0, 9 + (s - 111) / 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // the second-half fragments are consumed from slot 64 on; their reads (slots 0 .. 42) are waited for by the lgkmcnt(0) of the barriers when
    // there is a DMA, and explicitly otherwise
    static_for<0, 52>(slot);
    if (!DMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    frag_commit(1);                                     // (contraction-major operands: assemble the second-half fragments behind their wait)
    static_for<52, 128>(slot);
    if (DMA) next_tile();
    if (NEXT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_commit(0);
    }
  };

  // one K tile: DMA = tile t+2 exists (goes into `cur`), NEXT = tile t+1 exists (its first-half fragments come from `nxt`)
  auto tile_two_barriers = [&](const bool DMA, const bool NEXT, unsigned cur, unsigned nxt) __attribute__((always_inline)) {
    const StageBases bcur = bases_of(cur), bnxt = bases_of(NEXT ? nxt : cur);
    auto slot = [&](const int s) __attribute__((always_inline)) {
      const int ks = s >> 6, nb = (s >> 3) & 7, mb = s & 7;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nb][mb]) : "v"(wf[ks][nb]), "v"(af[ks][mb]));
      if (s < 16 * RD_STEP && s % RD_STEP == 0) frag_read(bcur, 1, s / RD_STEP);
      if (s == BAR1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // (r05 null: wave w issuing its piece at slot 40 + 4 j + w instead of all four waves in the same slot — so that the four requests do
      // not queue in the CU's one texture addresser — is 19 % SLOWER: the 64 wave-dependent scalar branches per tile cost more than the queue.)
      if (DMA && s >= DMA0 && s < DMA0 + 16 * DMA_STEP && (s - DMA0) % DMA_STEP == 0) dma_piece(cur, (s - DMA0) / DMA_STEP);
      if (EARLY_NEXT) {
        // r05: the next tile's first-half fragments are read at the LDS port's pace — one ds_read_b128 every other slot from slot 64 on,
        // where the first-half registers are dead (slots 0 .. 63 were their last readers) — instead of 16 back to back behind slot
        // 103: four waves x 16 KiB in one burst take the port 256 - 512 cycles, of which only the 8 slots up to the tile's end covered 128.
        // The pieces of tile t + 1 (issued during tile t - 1) must have landed by then: all but the 6 pieces of tile t + 2 issued so far.
        // Same-box A / B at 128K rows: fc2 13.44 -> 12.96 ms, fc1 + SwiGLU 26.3 -> 25.5, o 4.91 -> 4.81, qkv at 16K 0.859 -> 0.851.
        // (Going on to ONE barrier per tile — at slot 63, the DMA of tile t + 2 issued behind it — was mixed: fc1 - 3 %, fc2 + 1 %.)
        if (NEXT && s == 63) {
          if (DMA) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"((63 - DMA0) / DMA_STEP + 1) : "memory");    // the pieces of tile t + 2 issued by slot 63 stay in flight
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (NEXT && s >= 64 && s < 96 && (s & 1) == 0) frag_read(bnxt, 0, (s - 64) >> 1);
      } else {
        if (NEXT && s == 103) {
          if (DMA) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (NEXT && s >= 104 && s < 120) frag_read(bnxt, 0, s - 104);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // (two loops: the TN fragment assembly sits between them, outside either body — with it inside, the fully unrolled 128-slot
    // loop exceeded the optimizer's pragma-unroll size limit, stayed a loop, and the accumulators went to scratch)
#pragma unroll
    for (int s = 0; s <= BAR1; ++s) slot(s);
    frag_commit(1);
#pragma unroll
    for (int s = BAR1 + 1; s < 128; ++s) slot(s);
    if (DMA) next_tile();
    if (NEXT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_commit(0);
    }
  };

  // the K loop; W_ = std::integral_constant<int, slots this wave's LDS-DMA pieces are shifted by>
  auto k_loop = [&](auto W_) __attribute__((always_inline)) {
    auto tile = [&](const bool DMA, const bool NEXT, unsigned cur, unsigned nxt) __attribute__((always_inline)) {
      if (SPLIT_SCHED) tile_split(W_, DMA, NEXT, cur, nxt);
      else tile_two_barriers(DMA, NEXT, cur, nxt);
    };
    int t = 0;
    for (; t + 2 < nk; ++t) tile(true, true, lds0 + (t & 1) * STG, lds0 + ((t + 1) & 1) * STG);
    if (t + 1 < nk) {
      tile(false, true, lds0 + (t & 1) * STG, lds0 + ((t + 1) & 1) * STG);
      ++t;
    }
    tile(false, false, lds0 + (t & 1) * STG, 0u);
  };
  if (VITA_GEMM_WAVE_STAGGER && SPLIT_SCHED) {
    // r06 A / B: the four waves of a workgroup issue their pieces in DIFFERENT slots (wave w one slot behind wave w - 1), so that the texture
    // addresser does not get four 1-KiB requests in the same cycle.  Four copies of the loop behind one branch on the wave number (r05's attempt
    // branched per piece and lost 19 % to the branches); barriers and waits are at the same slots in all four.
    if (wave == 0) k_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
    else k_loop(std::integral_constant<int, 3>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }

  // ---- epilogue.  The inline-asm MFMAs are invisible to the compiler's hazard tracking: one wait for the matrix pipe, then every row
  // block's accumulators pass through an (empty) asm statement of their own right before they are read — asm volatile statements
  // keep their order, so no accumulator read can be placed above the wait, and the AGPR -> VGPR copies stay next to their use ---------
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  auto pin_row_block = [&](int mb) __attribute__((always_inline)) {
    asm volatile("" : "+a"(acc[0][mb]), "+a"(acc[1][mb]), "+a"(acc[2][mb]), "+a"(acc[3][mb]), "+a"(acc[4][mb]), "+a"(acc[5][mb]),
                 "+a"(acc[6][mb]), "+a"(acc[7][mb]));
  };
  // block (nb, mb): lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4)
  const int64_t mrow = m0 + wm * 128 + (lane & 15);
  const int64_t ncol = n0 + wn * (EPI == VITA_EPI_SWIGLU ? 64 : 128) + 4 * (lane >> 4);
  if (SPLITK) {
    // split-K partial: the raw fp32 accumulators of this (split, tile) -> ((float*)C)[split][m][n .. n+3]; vita_splitk_reduce sums the splits
    float* const cf = reinterpret_cast<float*>(p.C) + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      pin_row_block(mb);
      float* crow = cf + (mrow + mb * 16) * p.N + ncol;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) *reinterpret_cast<f32x4*>(crow + nb * 16) = acc[nb][mb];
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if (INTERIOR) {
    // whole tiles only (all decoder shapes): no masks.  With one workgroup per CU nothing else runs while a wave waits
    // for memory, so the bias / LayerScale / residual operands of one row block are loaded ahead of the previous block's arithmetic
    const bool has_bias = EPI != VITA_EPI_NONE && EPI != VITA_EPI_SWIGLU && p.bias;
    float bq[8][4], sq[8][4];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) { bq[nb][j] = 0.f; sq[nb][j] = 0.f; }
    if (has_bias) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) unpack_quad(*reinterpret_cast<const u32x2*>(p.bias + ncol + nb * 16), bq[nb]);
    }
    if (EPI == VITA_EPI_BIAS_SCALE_RES) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) unpack_quad(*reinterpret_cast<const u32x2*>(p.scale + ncol + nb * 16), sq[nb]);
    }
    u32x2 rq[2][8];
    auto load_res = [&](int mb) __attribute__((always_inline)) {
      if (epi_has_residual<EPI>()) {
        const bf16_t* rrow = p.R + (mrow + mb * 16) * p.ldr + ncol;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) rq[mb & 1][nb] = *reinterpret_cast<const u32x2*>(rrow + nb * 16);
      }
    };
    load_res(0);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if (mb + 1 < 8) load_res(mb + 1);
      pin_row_block(mb);
      bf16_t* crow = p.C + (mrow + mb * 16) * p.ldc + ncol;
      if (EPI == VITA_EPI_SWIGLU) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = bf16_round(acc[2 * pi][mb][j]);
            const float u = bf16_round(acc[2 * pi + 1][mb][j]);
            o[j] = bf16_round(g / (1.0f + __expf(-g))) * u;
          }
          u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(crow + pi * 16) = v;
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          float o[4], r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = acc[nb][mb][j];
          if (epi_has_residual<EPI>()) unpack_quad(rq[mb & 1][nb], r);
          epilogue_math<EPI>(o, has_bias, bq[nb], sq[nb], r);
          u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(crow + nb * 16) = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one row block at a time: keeps the accumulator reads from being hoisted wholesale
    }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    pin_row_block(mb);
    const int64_t m = mrow + mb * 16;
    if (m >= p.M) continue;
    if (EPI == VITA_EPI_SWIGLU) {
      // the wave's W rows = 4 x [gate 16 | up 16]; pair pi -> output columns n0 + wn * 64 + pi * 16 + 0..15
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        const int64_t n = ncol + pi * 16;
        if (n >= p.N) continue;
        float g[4], u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = acc[2 * pi][mb][j]; u[j] = acc[2 * pi + 1][mb][j]; }
        swiglu_quad(p, g, u, m, n);
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const int64_t n = ncol + nb * 16;
        if (n >= p.N) continue;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = acc[nb][mb][j];
        epilogue_quad<EPI>(p, o, m, n);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// r06, VITA_GEMM_PERSIST 1 (a same-box A / B of PERSISTENCE ALONE; off): a grid of one workgroup per CU, each walking the tiles bid, bid + grid, ...
// (bid % 8 — the XCD the tile order is built on — is the same for all of them).  Nothing is carried from one tile to the next: the prologue of tile
// i + 1 starts behind the epilogue of tile i, after a workgroup barrier (a wave that is done must not refill a stage a slower wave still reads).
#ifndef VITA_GEMM_PERSIST
#define VITA_GEMM_PERSIST 0
#endif
template <int EPI, bool INTERIOR, int OPM = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmArgs p) {
  if (VITA_GEMM_PERSIST && OPM == 0 && EPI != VITA_EPI_F32_PARTIAL) {
    const int nwg = p.tiles_m * p.tiles_n;
    for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
      gemm_w4_tile<EPI, INTERIOR, OPM>(p, bid);
      __syncthreads();
    }
  } else {
    gemm_w4_tile<EPI, INTERIOR, OPM>(p, blockIdx.x);
  }
}

template <int EPI, bool INTERIOR>
int launch_gemm_w4_cfg(const GemmArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<EPI, INTERIOR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              w4::LDS_BYTES);
  });
  unsigned grid = (unsigned)(a.tiles_m * a.tiles_n);
  if (VITA_GEMM_PERSIST && grid > 256u) grid = 256u;            // one workgroup per CU (256 = 8 XCDs x 32)
  hipLaunchKernelGGL((gemm_w4_kernel<EPI, INTERIOR>), dim3(grid), dim3(256), w4::LDS_BYTES, st, a);
  return vita_check_launch();
}

template <int EPI>
int launch_gemm_w4(GemmArgs a, hipStream_t st) {
  const int64_t tm = (a.M + 255) / 256;
  const int64_t bn_out = EPI == VITA_EPI_SWIGLU ? 128 : 256;
  const int64_t tn = (a.N + bn_out - 1) / bn_out;
  if (tm * tn > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  const bool whole_n = a.N % bn_out == 0 && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0);   // 8-byte accesses of whole quads
  if (whole_n && a.M % 256 == 0) return launch_gemm_w4_cfg<EPI, true>(a, st);
  if (whole_n && a.M > 256) {
    // ragged M only (the ViT's frames x 1025 rows): whole tile rows through the mask-free instantiation, the last rows separately
    GemmArgs top = a;
    top.M = a.M / 256 * 256; top.tiles_m = (int)(top.M / 256);
    const int rc = launch_gemm_w4_cfg<EPI, true>(top, st);
    if (rc != VITA_OK) return rc;
    GemmArgs rest = a;
    rest.M = a.M - top.M; rest.tiles_m = 1;
    rest.A = a.A + top.M * a.lda; rest.C = a.C + top.M * a.ldc;
    if (a.R) rest.R = a.R + top.M * a.ldr;
    return launch_gemm_w4_cfg<EPI, false>(rest, st);
  }
  return launch_gemm_w4_cfg<EPI, false>(a, st);
}

// the w4 kernel addresses a tile's rows with 32-bit byte offsets from the tile's first row
inline bool gemm_w4_addressable(const GemmArgs& a, bool swiglu) {
  const int64_t a_span = 256 * a.lda * 2, w_span = (swiglu ? a.N + 128 : 256) * a.ldw * 2;
  return a_span < 0x7fff0000LL && w_span < 0x7fff0000LL;
}

// ---- skinny-M (M <= 16): one wave per output column, x rows cached in LDS ------------------
// HBM-bound on W: algorithmic bytes = N*K*2.  Each lane streams 16-byte pieces of one W row.
template <int MAXM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                          const bf16_t* __restrict__ W, int64_t ldw,
                                                          void* __restrict__ C, int64_t ldc, int M,
                                                          int64_t N, int64_t K, int out_f32) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= N) return;
  float acc[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
  const int nvec = (int)(K >> 3);
  const u32x4* wr = reinterpret_cast<const u32x4*>(W + n * ldw);
  for (int v = lane; v < nvec; v += 64) {
    const u32x4 wv = wr[v];
    float wf[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wf[2 * j] = bf16lo_to_f32(wv[j]); wf[2 * j + 1] = bf16hi_to_f32(wv[j]); }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        const u32x4 av = *reinterpret_cast<const u32x4*>(A + m * lda + v * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[m] += bf16lo_to_f32(av[j]) * wf[2 * j] + bf16hi_to_f32(av[j]) * wf[2 * j + 1];
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    if (m < M) {
      const float s = wave_reduce_sum(acc[m]);
      if (lane == 0) {
        if (out_f32) reinterpret_cast<float*>(C)[m * ldc + n] = s;
        else reinterpret_cast<bf16_t*>(C)[m * ldc + n] = f32_to_bf16(s);
      }
    }
  }
}

template <int EPI, int BM, int BN, int WM, int WN, bool PIN = true>
int launch_gemm_cfg(GemmArgs a, hipStream_t st) {
  constexpr int lds = 2 * (BM + BN) * BK * 2;
  const int64_t tm = (a.M + BM - 1) / BM;
  const int64_t bn_out = EPI == VITA_EPI_SWIGLU ? BN / 2 : BN;
  const int64_t tn = (a.N + bn_out - 1) / bn_out;
  if (tm * tn > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, BM, BN, WM, WN, PIN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, WM, WN, PIN>), dim3((unsigned)(tm * tn)), dim3(64 * WM * WN), lds, st, a);
  return vita_check_launch();
}

// VITA_GEMM_TILE=128|256 forces a tile configuration (developer tuning aid).
inline int gemm_tile_override() {
  static int v = -1;
  if (v < 0) {
    const char* e = vita_dev_getenv("VITA_GEMM_TILE");
    v = e ? atoi(e) : 0;
  }
  return v;
}

template <int EPI>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
  // the 256x256 tile needs enough tiles to fill 256 CUs with one workgroup each
  const int64_t bn_out = EPI == VITA_EPI_SWIGLU ? 128 : 256;
  const int64_t big_tiles = ((a.M + 255) / 256) * ((a.N + bn_out - 1) / bn_out);
  bool big = big_tiles >= 192;
  if (gemm_tile_override() == 128) big = false;
  if (gemm_tile_override() == 256) big = true;
  static const bool nopin = vita_dev_getenv("VITA_GEMM_NOPIN") != nullptr;     // developer tuning aid
  if (gemm_tile_override() == 2564) return launch_gemm_cfg<EPI, 256, 256, 2, 2>(a, st);   // 4 waves x (128 x 128), compiler-scheduled
  // VITA_GEMM_KERNEL (developer aid, read per launch): "w4" / "w8" force the large-problem kernel, "128" the small tile
  const char* kn = vita_dev_getenv("VITA_GEMM_KERNEL");
  const bool w4_ok = gemm_w4_addressable(a, EPI == VITA_EPI_SWIGLU);
  if (kn && kn[0] == 'w' && kn[1] == '4' && w4_ok) return launch_gemm_w4<EPI>(a, st);
  if (kn && kn[0] == 'w' && kn[1] == '8') return launch_gemm_cfg<EPI, 256, 256, 2, 4>(a, st);
  if (kn && kn[0] == '1') return launch_gemm_cfg<EPI, 128, 128, 2, 2>(a, st);
  if (EPI == VITA_EPI_NONE) {     // developer measurement aids, read per launch (tools/microbench.py variants)
    const char* e = vita_dev_getenv("VITA_GEMM_EXP");
    const int v = e ? atoi(e) : 0;
    if (v == 10) { GemmArgs b = a; b.ldr = -1; return launch_gemm_cfg<EPI, 256, 256, 2, 4>(b, st); }   // 8 waves, L2-hit loads
    if (v == 4) return launch_gemm4<EPI>(a, st);                                                        // 4 waves x (128 x 128)
    if (v == 41) return launch_gemm4<VITA_EPI_NONE, 1>(a, st);
    if (v == 42) return launch_gemm4<VITA_EPI_NONE, 2>(a, st);
    if (v == 43) return launch_gemm4<VITA_EPI_NONE, 3>(a, st);
  }
  if (nopin) return big ? launch_gemm_cfg<EPI, 256, 256, 2, 4, false>(a, st) : launch_gemm_cfg<EPI, 128, 128, 2, 2, false>(a, st);
  if (big && w4_ok) return launch_gemm_w4<EPI>(a, st);
  return big ? launch_gemm_cfg<EPI, 256, 256, 2, 4>(a, st) : launch_gemm_cfg<EPI, 128, 128, 2, 2>(a, st);
}

}  // namespace

extern "C" int vita_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C,
                              int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                              const void* bias, const void* scale, const void* R, int64_t ldr,
                              void* stream) {
  if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if (K % BK) return VITA_ERR_UNSUPPORTED;
  if ((lda & 7) || (ldw & 7) || (ldc & 3)) return VITA_ERR_UNSUPPORTED;
  if (M == 0) return VITA_OK;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw;
  a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias = (const bf16_t*)bias; a.scale = (const bf16_t*)scale; a.R = (const bf16_t*)R; a.ldr = ldr;
  a.tiles_m = a.tiles_n = 0;
  {
    const char* e = vita_dev_getenv("VITA_GEMM_STAGGER");          // developer tuning aid, read per launch
    a.stagger = e ? atoi(e) : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VITA_EPI_NONE: return launch_gemm<VITA_EPI_NONE>(a, st);
    case VITA_EPI_BIAS:
      if (!bias) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_BIAS>(a, st);
    case VITA_EPI_BIAS_GELU: return launch_gemm<VITA_EPI_BIAS_GELU>(a, st);
    case VITA_EPI_RESIDUAL:
      if (!R || (ldr & 3)) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_RESIDUAL>(a, st);
    case VITA_EPI_BIAS_SCALE_RES:
      if (!R || !scale || (ldr & 3)) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_BIAS_SCALE_RES>(a, st);
    case VITA_EPI_SWIGLU: return launch_gemm<VITA_EPI_SWIGLU>(a, st);
    case VITA_EPI_BIAS2_GELU_TANH:
      if (!bias) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_BIAS2_GELU_TANH>(a, st);
    case VITA_EPI_BIAS2_RES:
      if (!bias || !R || (ldr & 3)) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_BIAS2_RES>(a, st);
    case VITA_EPI_BIAS2_GELU:
      if (!bias) return VITA_ERR_INVALID_ARG;
      return launch_gemm<VITA_EPI_BIAS2_GELU>(a, st);
    default: return VITA_ERR_INVALID_ARG;
  }
}

// C[M, N] = A_t^T W_t with both operands contraction-major (A_t [K, M], W_t [K, N]): the weight-gradient GEMM without transposes
extern "C" int vita_gemm_bf16_tn(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, void* stream) {
  if (!At || !Wt || !C || M <= 0 || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if ((M % 256) || (N % 256) || (K % BK) || (lda & 7) || (ldw & 7) || (ldc & 3)) return VITA_ERR_UNSUPPORTED;
  if (BK * lda * 2 + 512 >= 0x7fff0000LL || BK * ldw * 2 + 512 >= 0x7fff0000LL) return VITA_ERR_UNSUPPORTED;
  GemmArgs a;
  a.A = (const bf16_t*)At; a.lda = lda; a.W = (const bf16_t*)Wt; a.ldw = ldw;
  a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias = nullptr; a.scale = nullptr; a.R = nullptr; a.ldr = 0; a.stagger = 0;
  const int64_t tm = M / 256, tn = N / 256;
  if (tm * tn > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<VITA_EPI_NONE, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES);
  });
  hipLaunchKernelGGL((gemm_w4_kernel<VITA_EPI_NONE, true, 1>), dim3((unsigned)(tm * tn)), dim3(256), w4::LDS_BYTES, (hipStream_t)stream, a);
  return vita_check_launch();
}

// C[M, N] = A W with A [M, K] row-major and W [K, N] contraction-major: the input-gradient GEMM (dX = dY W, the weight as the forward
// holds it) without a transposed copy of the weight.  Whole tiles only; anything else returns VITA_ERR_UNSUPPORTED (callers transpose).
extern "C" int vita_gemm_bf16_nn(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                                 int64_t K, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if ((M % 256) || (N % 256) || (K % BK) || (lda & 7) || (ldw & 7) || (ldc & 3)) return VITA_ERR_UNSUPPORTED;
  if (BK * ldw * 2 + 512 >= 0x7fff0000LL || 256 * lda * 2 + 512 >= 0x7fff0000LL) return VITA_ERR_UNSUPPORTED;
  GemmArgs a;
  a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw;
  a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias = nullptr; a.scale = nullptr; a.R = nullptr; a.ldr = 0; a.stagger = 0;
  const int64_t tm = M / 256, tn = N / 256;
  if (tm * tn > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<VITA_EPI_NONE, true, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES);
  });
  hipLaunchKernelGGL((gemm_w4_kernel<VITA_EPI_NONE, true, 2>), dim3((unsigned)(tm * tn)), dim3(256), w4::LDS_BYTES, (hipStream_t)stream, a);
  return vita_check_launch();
}

namespace {
// C[m][n] = bf16(sum over the splits of part[s][m][n]); one thread per four columns
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, bf16_t* __restrict__ C, int64_t ldc, int64_t M,
                                                            int64_t N, int splits) {
  const int64_t quads = M * (N / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / (N / 4), n = (i - m * (N / 4)) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + m * N + n);
    for (int k = 1; k < splits; ++k) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(part + ((int64_t)k * M + m) * N + n);
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    const u32x2 w = {pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
    *reinterpret_cast<u32x2*>(C + m * ldc + n) = w;
  }
}

// out[c] += sum over rows of float(x[r][c]) (fp32, atomics: the caller zeroes `out`); one thread per four columns of a row block
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t rows, int cols,
                                                     int rows_per_block) {
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(x + r * ldx + c4);
    a0 += bf16lo_to_f32(v[0]); a1 += bf16hi_to_f32(v[0]); a2 += bf16lo_to_f32(v[1]); a3 += bf16hi_to_f32(v[1]);
  }
  atomicAdd(out + c4, a0); atomicAdd(out + c4 + 1, a1); atomicAdd(out + c4 + 2, a2); atomicAdd(out + c4 + 3, a3);
}
// The ORDERED form (r05): fp32 atomics add in arrival order, so two runs of the same step could differ in the last bit of a bias gradient
// (seen: the recompute test's bit-identity assertion on linear_qkv.bias).  Row block y writes its partial sums to part[y][cols]; one thread
// per four columns then adds the partials in block order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ part, int64_t rows,
                                                             int cols, int rows_per_block) {
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0; r < r1; ++r) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(x + r * ldx + c4);
    a[0] += bf16lo_to_f32(v[0]); a[1] += bf16hi_to_f32(v[0]); a[2] += bf16lo_to_f32(v[1]); a[3] += bf16hi_to_f32(v[1]);
  }
  *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.y * cols + c4) = a;
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int blocks, int cols) {
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= cols) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(out + c4);
  for (int y = 0; y < blocks; ++y) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(part + (int64_t)y * cols + c4);
    s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
  }
  *reinterpret_cast<f32x4*>(out + c4) = s;
}
// rows per block / number of row blocks of both forms: <= 1024 blocks (enough workgroups to stream at HBM speed), >= 32 rows each
inline int64_t colsum_rows_per_block(int64_t rows) {
  const int64_t rpb = (rows + 1023) / 1024;
  return rpb < 32 ? 32 : rpb;
}
}  // namespace

// The same product with the contraction cut into `splits` ranges (r04): a weight gradient whose output is a handful of 256 x 256 tiles
// (the ViT's linears: 16 - 64 tiles for 256 CUs; the decoder's qkv / proj at config 5: 1.1 rounds of the chip) over a contraction of
// 16 K - 260 K token rows.  workspace: fp32 [splits][M][N].  splits = 1 is vita_gemm_bf16_tn.
extern "C" size_t vita_gemm_tn_splitk_workspace_bytes(int64_t M, int64_t N, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int vita_gemm_bf16_tn_splitk(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                        int64_t N, int64_t K, int splits, void* workspace, void* stream) {
  if (splits <= 1) return vita_gemm_bf16_tn(At, lda, Wt, ldw, C, ldc, M, N, K, stream);
  if (!At || !Wt || !C || !workspace || M <= 0 || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if ((M % 256) || (N % 256) || (K % BK) || (lda & 7) || (ldw & 7) || (ldc & 3)) return VITA_ERR_UNSUPPORTED;
  if (BK * lda * 2 + 512 >= 0x7fff0000LL || BK * ldw * 2 + 512 >= 0x7fff0000LL) return VITA_ERR_UNSUPPORTED;
  const int64_t nk = K / BK;
  const int64_t per = (nk + splits - 1) / splits;
  if (per < 2 || nk - per * (splits - 1) < 1) return VITA_ERR_INVALID_ARG;            // every split gets at least one K tile
  GemmArgs a;
  a.A = (const bf16_t*)At; a.lda = lda; a.W = (const bf16_t*)Wt; a.ldw = ldw;
  a.C = (bf16_t*)workspace; a.ldc = N; a.M = M; a.N = N; a.K = K;
  a.bias = nullptr; a.scale = nullptr; a.R = nullptr; a.ldr = 0; a.stagger = 0;
  a.splits = splits; a.k_tiles_per_split = (int)per;
  const int64_t tm = M / 256, tn = N / 256;
  if (tm * tn * splits > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<VITA_EPI_F32_PARTIAL, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES);
  });
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((gemm_w4_kernel<VITA_EPI_F32_PARTIAL, true, 1>), dim3((unsigned)(tm * tn * splits)), dim3(256), w4::LDS_BYTES, st, a);
  const int64_t quads = M * (N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192)), dim3(256), 0, st,
                     (const float*)workspace, (bf16_t*)C, ldc, M, N, splits);
  return vita_check_launch();
}

// out[c] += sum_r float(x[r][c]): the bias gradient grad_output.sum(dim = 0) (M/core/tensor_parallel/layers.py:524) as ONE pass over
// grad_output — through r03 it was a GEMM against a block of ones (a 259 K-row contraction on 4 - 16 workgroups: 5 ms per ViT linear).
// out fp32 [cols], zeroed by the caller; cols % 4 == 0.
extern "C" int vita_colsum_bf16(const void* x, int64_t ldx, float* out, int64_t rows, int cols, void* stream) {
  if (!x || !out || rows < 0 || cols <= 0) return VITA_ERR_INVALID_ARG;
  if ((cols & 3) || (ldx & 3)) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  const int gx = (cols / 4 + 255) / 256;
  const int64_t rpb = colsum_rows_per_block(rows);       // <= 1024 row blocks: enough workgroups to stream at HBM speed, few atomics
  const int64_t gy = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, out, rows,
                     cols, (int)rpb);
  return vita_check_launch();
}

// ABI 18: the same sums added in a FIXED order (bit-reproducible from run to run, as torch's sum is): workspace = fp32 [row blocks][cols]
extern "C" size_t vita_colsum_workspace_bytes(int64_t rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  const int64_t rpb = colsum_rows_per_block(rows);
  return (size_t)((rows + rpb - 1) / rpb) * (size_t)cols * sizeof(float);
}
extern "C" int vita_colsum_bf16_ordered(const void* x, int64_t ldx, float* out, int64_t rows, int cols, void* workspace, void* stream) {
  if (!x || !out || rows < 0 || cols <= 0 || (rows > 0 && !workspace)) return VITA_ERR_INVALID_ARG;
  if ((cols & 3) || (ldx & 3) || ((uintptr_t)out & 15) || ((uintptr_t)workspace & 15)) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  const int gx = (cols / 4 + 255) / 256;
  const int64_t rpb = colsum_rows_per_block(rows);
  const int64_t gy = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (float*)workspace, rows, cols, (int)rpb);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, (int)gy, cols);
  return vita_check_launch();
}

extern "C" int vita_gemm_skinny_bf16(const void* A, int64_t lda, const void* W, int64_t ldw,
                                     void* C, int64_t ldc, int M, int64_t N, int64_t K,
                                     int out_f32, void* stream) {
  if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return VITA_ERR_INVALID_ARG;
  if (M > 16 || (K & 7) || (lda & 7) || (ldw & 7)) return VITA_ERR_UNSUPPORTED;
  if (M == 0) return VITA_OK;
  dim3 grid((unsigned)((N + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (M <= 2)
    hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, block, 0, st, (const bf16_t*)A, lda,
                       (const bf16_t*)W, ldw, C, ldc, M, N, K, out_f32);
  else if (M <= 8)
    hipLaunchKernelGGL(gemm_skinny_kernel<8>, grid, block, 0, st, (const bf16_t*)A, lda,
                       (const bf16_t*)W, ldw, C, ldc, M, N, K, out_f32);
  else
    hipLaunchKernelGGL(gemm_skinny_kernel<16>, grid, block, 0, st, (const bf16_t*)A, lda,
                       (const bf16_t*)W, ldw, C, ldc, M, N, K, out_f32);
  return vita_check_launch();
}
