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
};

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

// ---- epilogue shared by the tile kernels: lane holds D^T: row m = ... + (lane & 31), columns 8*rg + 4*(lane>>5) + 0..3 --
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
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = bf16_round(acc[2 * pi][mi][rg * 4 + j]);
            const float u = bf16_round(acc[2 * pi + 1][mi][rg * 4 + j]);
            const float s = bf16_round(g / (1.0f + __expf(-g)));
            o[j] = s * u;
          }
          bf16_t* dst = p.C + m * p.ldc + n;
          if (n + 3 < p.N) {
            u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(dst) = v;
          } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) dst[j] = f32_to_bf16(o[j]);
          }
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int64_t n = n0 + wn * TN + ni * 32 + rg * 8 + hi * 4;
          if (n >= p.N) continue;
          const bool full = n + 3 < p.N;
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = acc[ni][mi][rg * 4 + j];
          if (EPI == VITA_EPI_BIAS2_GELU_TANH || EPI == VITA_EPI_BIAS2_RES) {      // bias added to the ROUNDED product
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = bf16_round(o[j]);
          }
          if (EPI != VITA_EPI_NONE && p.bias) {
            if (full) {
              const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
              o[0] += bf16lo_to_f32(b[0]); o[1] += bf16hi_to_f32(b[0]);
              o[2] += bf16lo_to_f32(b[1]); o[3] += bf16hi_to_f32(b[1]);
            } else {
              for (int j = 0; j < 4 && n + j < p.N; ++j) o[j] += bf16_to_f32(p.bias[n + j]);
            }
          }
          if (EPI == VITA_EPI_BIAS_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = gelu_erf(bf16_round(o[j]));
          }
          if (EPI == VITA_EPI_BIAS2_GELU_TANH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = gelu_tanh(bf16_round(o[j]));
          }
          if (EPI == VITA_EPI_BIAS_SCALE_RES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int64_t nn = (n + j < p.N) ? n + j : p.N - 1;
              o[j] = bf16_round(bf16_round(o[j]) * bf16_to_f32(p.scale[nn]));
            }
          }
          if (EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_BIAS_SCALE_RES || EPI == VITA_EPI_BIAS2_RES) {
            const bf16_t* rsrc = p.R + m * p.ldr + n;
            if (EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_BIAS2_RES) {
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = bf16_round(o[j]);
            }
            if (full) {
              const u32x2 rv = *reinterpret_cast<const u32x2*>(rsrc);
              o[0] += bf16lo_to_f32(rv[0]); o[1] += bf16hi_to_f32(rv[0]);
              o[2] += bf16lo_to_f32(rv[1]); o[3] += bf16hi_to_f32(rv[1]);
            } else {
              for (int j = 0; j < 4 && n + j < p.N; ++j) o[j] += bf16_to_f32(rsrc[j]);
            }
          }
          bf16_t* dst = p.C + m * p.ldc + n;
          if (full) {
            u32x2 v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2*>(dst) = v;
          } else {
            for (int j = 0; j < 4 && n + j < p.N; ++j) dst[j] = f32_to_bf16(o[j]);
          }
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

  const int nwg = p.tiles_m * p.tiles_n;
  int pid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
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
    const char* e = getenv("VITA_GEMM_TILE");
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
  static const bool nopin = getenv("VITA_GEMM_NOPIN") != nullptr;     // developer tuning aid
  if (gemm_tile_override() == 2564) return launch_gemm_cfg<EPI, 256, 256, 2, 2>(a, st);   // 4 waves x (128 x 128)
  if (EPI == VITA_EPI_NONE) {     // developer measurement aids, read per launch (tools/microbench.py variants)
    const char* e = getenv("VITA_GEMM_EXP");
    const int v = e ? atoi(e) : 0;
    if (v == 10) { GemmArgs b = a; b.ldr = -1; return launch_gemm_cfg<EPI, 256, 256, 2, 4>(b, st); }   // 8 waves, L2-hit loads
    if (v == 4) return launch_gemm4<EPI>(a, st);                                                        // 4 waves x (128 x 128)
    if (v == 41) return launch_gemm4<VITA_EPI_NONE, 1>(a, st);
    if (v == 42) return launch_gemm4<VITA_EPI_NONE, 2>(a, st);
    if (v == 43) return launch_gemm4<VITA_EPI_NONE, 3>(a, st);
  }
  if (nopin) return big ? launch_gemm_cfg<EPI, 256, 256, 2, 4, false>(a, st) : launch_gemm_cfg<EPI, 128, 128, 2, 2, false>(a, st);
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
    const char* e = getenv("VITA_GEMM_STAGGER");          // developer tuning aid, read per launch
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
    default: return VITA_ERR_INVALID_ARG;
  }
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
