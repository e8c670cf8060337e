// Flash attention forward for gfx950 (MI355X): online softmax, bf16 I/O, fp32 statistics.
// head_dim 128 (LLM: causal GQA 40:8, zig-zag context-parallel chunk geometry) and 64 (ViT,
// non-causal, 1025 tokens).  Bound: MFMA; algorithmic work 4 * d flop per visible (q, k) pair.
//
// Structure (one workgroup = 8 waves = 256 query rows of ONE query head; KV tile = 64 keys):
//   * K and V tiles are staged HBM/L2 -> registers -> LDS (double buffered, one barrier per tile):
//     the global loads of tile t+1 are issued before tile t is computed and written to the other
//     LDS buffer afterwards, so HBM/L2 latency hides under the MFMA work of the current tile.
//   * S^T = K Q^T ("swapped" product, v_mfma_f32_32x32x16_bf16): a lane then owns ONE query row
//     (column lane&31) and 32 of the 64 keys, so row max / row sum are in-lane reductions plus one
//     v_permlane32_swap with the partner lane -- no LDS traffic for the softmax.
//   * O^T = V^T P^T: the P^T operand is exactly the packed S^T accumulator (any consistent
//     assignment of keys to MFMA k-slots is valid because the contraction is a sum), and the V^T
//     operand comes from the row-major V tile through the LDS transpose read ds_read_b64_tr_b16.
//   * LDS layouts are XOR-swizzled so that ds_read_b128 (K fragments) and ds_read_b64_tr_b16
//     (V^T fragments) are bank-conflict free for the lane groups gfx950 services them in.
//   * Sequence geometry is chunked (see vita_attn_params): causal visibility is decided per
//     (query chunk id, key chunk id) pair, element masks are only evaluated on diagonal tiles.
//   * Workgroup order: kv head = block id % n_kv_heads (= the XCD when n_kv_heads = 8, so one
//     XCD's L2 serves one kv head's K/V stream to all its concurrently running query tiles), query
//     tiles heaviest-first so the causal tail is short.
//
// Reference behaviour restated: M/core/transformer/dot_product_attention.py:186-289 (unfused
// math: softmax(QK^T / sqrt(d)) V with GQA repeat :171-175), :312-329 (ViT, non-causal),
// :374-390 (LLM causal).  Zig-zag chunk ownership: M/training/utils.py:329-341.
#include "attn_args.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// developer aid (VITA_ATTN_VARIANT bit 2): per-phase shader-clock totals, [group A|B][top, qk, sm_pv, barrier, n]
__device__ unsigned long long g_attn_timing[16];
constexpr int QTILE = 256;   // query rows per workgroup (8 waves x 32)
constexpr int KVT = 64;      // keys per tile

__device__ __forceinline__ float swap32_max(float x) {
  const unsigned xi = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap32_sum(float x) {
  const unsigned xi = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS byte offset of K element block (row, 16-byte slot) ; row = key within tile
template <int D>
__device__ __forceinline__ int k_lds_off(int row, int slot) {
  if (D == 128) return row * 256 + ((slot ^ (row & 15)) << 4);
  else if (D == 96) return row * 192 + (((slot + ((row >> 2) & 3)) % 12) << 4);    // see the note on d = 96 below
  else return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}
// d = 96 (r05: SigLIP's head size 72 zero-padded to 96 instead of 128): a row is 192 bytes = 12 slots = 3/4 of a bank period, so an XOR
// swizzle does not fit.  K fragments (ds_read_b128, 16 consecutive rows per pass): 16-byte unit = (12 row + pslot) mod 16 with pslot =
// (slot + (row >> 2 & 3)) mod 12 is a permutation of 0 .. 15 over 16 consecutive rows (12 = -4 mod 16: rows r, r + 1, r + 2, r + 3 start
// 0, 12, 8, 4 units in; the +0 .. 3 rotation of rows r + 4 k fills each group of four; a wrap mod 12 moves a whole group by +4).
// V^T fragments (ds_read_b64_tr_b16, 4 rows x 64 bytes per 32-lane pass): 192 = -64 mod 256, so rows r .. r + 3 already sit in four
// different 64-byte windows of the bank period: no swizzle at all.
// LDS byte offset of V (row, 32-byte chunk c, byte b within chunk)
template <int D>
__device__ __forceinline__ int v_lds_off(int row, int chunk, int b) {
  if (D == 128) return row * 256 + ((chunk ^ ((row & 3) << 1)) << 5) + b;
  else if (D == 96) return row * 192 + (chunk << 5) + b;
  else return row * 128 + ((chunk ^ (row & 2)) << 5) + b;
}

// One kv tile position of the iteration space (all fields wave-uniform -> SGPRs).
struct TileIt {
  int c, j, n;        // chunk, tile inside chunk, tiles to visit in this chunk; c == n_kv_chunks -> end
  int rows;           // valid rows of chunk c
  int diag;           // CAUSAL and chunk c is the query tile's own chunk
  int64_t crow;       // first row of chunk c in the K/V buffers
};

// VARIANT (developer tuning aid, see attn_variant()):
//   0 = K/V tiles go HBM/L2 -> LDS directly with the LDS-DMA (global_load_lds_dwordx4); the DMA writes
//       lane-linear, so the bank swizzles are applied to the per-lane SOURCE address (same involution
//       on the fragment reads); no staging VGPRs, no ds_write burst behind the barrier;
//   1 = register staging (global_load -> VGPR -> ds_write_b128), loads issued two tiles ahead;
//   bits 1-2 = QK_AHEAD: K-fragment reads pinned that many k-steps ahead of their MFMAs.
// Measured and dropped: a 3-slot "staggered" schedule (waves 4..7 half a tile behind
// waves 0..3): 0 to -3 %; s_setprio around the MFMA clusters: -2 %; a 4-wave x 64-row variant with
// one wave per SIMD and compiler-allocated AGPRs: -19 %; an in-wave software-pipelined 4-wave x 32-row kernel
// (exp of tile t between the MFMAs of tile t+1, 4-slot ring): -35 % (LDS-bound; see DESIGN.md 4.1, git history).
template <int D, bool CAUSAL, int VARIANT = 0, bool TIMING = false>
__global__ __launch_bounds__(512, 2) void flash_fwd_kernel(AttnArgs p) {
  constexpr int DS = D / 16;              // QK^T k-steps
  constexpr int DB = D / 32;              // O^T row blocks
  constexpr int ROWB = D * 2;             // bytes per K/V row
  constexpr int TILEB = KVT * ROWB;       // bytes per K (or V) tile
  constexpr int SLOTB = 2 * TILEB;        // bytes per ring slot (K tile | V tile)
  constexpr int SLOTS = ROWB / 16;        // 16-byte slots per row
  constexpr int LD_PER_THR = (KVT * SLOTS) / 512;  // 16-byte loads per thread per operand
  constexpr bool GLDS = !(VARIANT & 1);
  static_assert(GLDS || (KVT * SLOTS) % 512 == 0, "register staging deals whole 16-byte pieces to 512 threads (d = 96: LDS-DMA only)");
  constexpr int QK_AHEAD = (VARIANT >> 1) & 3;          // 0 = compiler's own schedule

  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
  // LDS byte address of the ring (32-bit); fragment reads use  VGPR(lane offset + slot base) + imm
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- work decomposition -------------------------------------------------------------------
  const int G = p.n_q_heads / p.n_kv_heads;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int hq = bid % G; bid /= G;
  const int n_q_tiles = p.n_q_chunks * p.tiles_per_q_chunk;
  const int qt_order = bid % n_q_tiles;
  const int b = bid / n_q_tiles;
  const int head = kvh * G + hq;
  const int qc = p.q_order[qt_order / p.tiles_per_q_chunk];
  const int qti = p.tiles_per_q_chunk - 1 - qt_order % p.tiles_per_q_chunk;
  const int gq = p.q_gid[qc];
  const int q_rows_in_chunk = (qc == p.n_q_chunks - 1) ? p.q_valid : p.chunk_len;
  const int q_off_wg = qti * QTILE;                 // offset of this tile inside its chunk
  const int q_off = q_off_wg + wave * 32;           // this wave's first row inside the chunk
  const int my_q = q_off + l31;                     // this lane's row inside the chunk
  const bool q_live = my_q < q_rows_in_chunk;
  const int64_t q_local_row = (int64_t)qc * p.chunk_len + (q_live ? my_q : q_rows_in_chunk - 1);
  const int q_last_wg = min(q_off_wg + QTILE, q_rows_in_chunk) - 1;  // last valid row of the WG
  const float scale_log2e = p.scale_log2e;
  // packed sequences (block-diagonal causal, single chunk): a query only sees keys >= the first row of its segment.
  // seg_start is non-decreasing, so the workgroup starts at the tile of its first row's segment and only the tiles
  // below its last row's segment start need the extra element mask.
  const int my_start = p.seg_start ? p.seg_start[q_local_row] : 0;
  const int wg_first_start = p.seg_start ? p.seg_start[(int64_t)qc * p.chunk_len + q_off_wg] : 0;
  const int wg_last_start = p.seg_start ? p.seg_start[(int64_t)qc * p.chunk_len + q_last_wg] : 0;

  // ---- Q fragments (B operand of S^T = K Q^T): lane = (query row l31, k-slot half hi) --------
  bf16x8 qf[DS];
  {
    const bf16_t* qp = p.q + (int64_t)b * p.q_bs + q_local_row * p.q_rs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  // ---- per-lane LDS read offsets (everything else is a compile-time immediate) ----------------
  // K fragment (A operand): row l31 (+32), 16-byte slot 2*ds + hi
  unsigned koff[DS];
#pragma unroll
  for (int ds = 0; ds < DS; ++ds) koff[ds] = k_lds_off<D>(l31, 2 * ds + hi);
  // V^T fragment for O^T row block db: inside a 16-lane group, lane i supplies the 8-byte piece
  // (row i>>2, columns 4*(i&3)..) of a [4 keys][16 d] block; keys 16t + 4*hi' + {0..3} (+8)
  unsigned voff[DB];
  {
    const int g16 = lane >> 4, i16 = lane & 15;
    const int key_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      voff[db] = TILEB + v_lds_off<D>(key_l, col >> 4, (col & 15) * 2);
    }
  }
  // staging: this thread's LD_PER_THR 16-byte pieces of a tile -> global element offsets
  // (relative to the tile's first row, 32-bit) and LDS byte offsets (relative to the slot)
  unsigned g_koff[LD_PER_THR], g_voff[LD_PER_THR], l_koff[LD_PER_THR], l_voff[LD_PER_THR];
#pragma unroll
  for (int it = 0; it < LD_PER_THR; ++it) {
    const int e = tid + it * 512;
    const int row = e / SLOTS, slot = e % SLOTS;
    g_koff[it] = (unsigned)(row * p.k_rs + slot * 8);
    g_voff[it] = (unsigned)(row * p.v_rs + slot * 8);
    l_koff[it] = k_lds_off<D>(row, slot);
    l_voff[it] = TILEB + v_lds_off<D>(row, slot >> 1, (slot & 1) << 4);
  }

  f32x16 o_acc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const bf16_t* kbase = p.k + (int64_t)b * p.k_bs + (int64_t)kvh * p.k_hs;
  const bf16_t* vbase = p.v + (int64_t)b * p.v_bs + (int64_t)kvh * p.v_hs;

  // ---- tile iterator ------------------------------------------------------------------------------
  auto enter_chunk = [&](TileIt& t) __attribute__((always_inline)) {   // skip chunks with nothing to visit
    while (t.c < p.n_kv_chunks) {
      t.rows = (t.c == p.n_kv_chunks - 1) ? p.kv_valid : p.chunk_len;
      const int all = (t.rows + KVT - 1) / KVT;
      const int gk = p.kv_gid[t.c];
      t.diag = CAUSAL && gk == gq;
      t.n = (!CAUSAL || gk < gq) ? all : (gk > gq ? 0 : min(all, q_last_wg / KVT + 1));
      if (t.n > 0) { t.crow = p.kv_row[t.c]; t.j = wg_first_start / KVT; return; }
      ++t.c;
    }
  };
  auto advance = [&](TileIt& t) __attribute__((always_inline)) {
    if (++t.j == t.n) { ++t.c; enter_chunk(t); }
  };

  // LDS-DMA path: wave w issues pieces q = 0..PIECES-1 of K and of V; piece (w, q) = rows
  // 4*(w*PIECES+q)*(256/ROWB).. of the tile (1 KiB), lane i lands at byte 16*i of the piece.
  // (d = 96: a piece is 5 1/3 rows and an operand 12 pieces; the 24 pieces of a slot are dealt three per wave — waves 0 .. 3 K, 4 .. 7 V —
  // and lane i of piece x fills the 16-byte unit u = 64 x + i = (row u / 12, physical slot u % 12).)
  constexpr bool D96 = D == 96;
  constexpr int PIECES = D96 ? 3 : TILEB / 1024 / 8;   // wave-instructions per operand per wave (d = 96: per wave, of ITS operand)
  constexpr int RPP = 1024 / ROWB;                      // tile rows per 1-KiB piece (d = 96: unused)
  const bool v_wave = D96 && wave >= 4;
  typedef __attribute__((address_space(1))) const void gvoid;
  typedef __attribute__((address_space(3))) void lvoid;
  unsigned dk_off[PIECES], dv_off[PIECES];             // per-lane source offsets (elements) inside a tile
  int d_row[PIECES], d_ks[PIECES], d_vs[PIECES];
#pragma unroll
  for (int q = 0; q < PIECES; ++q) {
    const int unit = (D96 ? (wave & 3) * PIECES + q : 0) * 64 + lane;
    const int row = D96 ? unit / SLOTS : (wave * PIECES + q) * RPP + lane / SLOTS;   // tile row this lane fills
    const int ps = D96 ? unit % SLOTS : lane % SLOTS;           // physical 16-byte slot in the row
    // logical slot whose data must land at physical slot ps (inverse of the read swizzles)
    d_ks[q] = (D == 128) ? (ps ^ (row & 15)) : D96 ? (ps + 12 - ((row >> 2) & 3)) % 12 : (ps ^ ((row >> 1) & 7));
    d_vs[q] = (D == 128) ? ((((ps >> 1) ^ ((row & 3) << 1)) << 1) | (ps & 1))
              : D96      ? ps
                         : ((((ps >> 1) ^ (row & 2)) << 1) | (ps & 1));
    d_row[q] = row;
    dk_off[q] = (unsigned)(row * p.k_rs + d_ks[q] * 8);
    dv_off[q] = (unsigned)(row * p.v_rs + d_vs[q] * 8);
  }
  auto dma_tile = [&](const TileIt& t, unsigned sl) __attribute__((always_inline)) {
    const int64_t row0 = t.crow + (int64_t)t.j * KVT;
    const bf16_t* kp = kbase + row0 * p.k_rs;           // wave-uniform bases + 32-bit lane offsets
    const bf16_t* vp = vbase + row0 * p.v_rs;
    const int left = t.rows - t.j * KVT;               // valid rows in this tile (>= 1)
    if constexpr (D96) {                               // this wave's three pieces of ITS operand; rows past the end clamped (masked later)
      const bf16_t* xp = v_wave ? vp : kp;
      const int64_t rs = v_wave ? p.v_rs : p.k_rs;
      const unsigned dst = sl + (v_wave ? TILEB : 0) + (wave & 3) * PIECES * 1024;
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int row = d_row[q] < left ? d_row[q] : left - 1;
        __builtin_amdgcn_global_load_lds((gvoid*)(xp + (int64_t)row * rs + (v_wave ? d_vs[q] : d_ks[q]) * 8),
                                         (lvoid*)(uintptr_t)(dst + q * 1024), 16, 0, 0);
      }
    } else if (left >= KVT) {
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int piece = wave * PIECES + q;
        __builtin_amdgcn_global_load_lds((gvoid*)(kp + dk_off[q]), (lvoid*)(uintptr_t)(sl + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gvoid*)(vp + dv_off[q]), (lvoid*)(uintptr_t)(sl + TILEB + piece * 1024), 16, 0, 0);
      }
    } else {                                            // padded tail: clamp rows (masked later)
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int piece = wave * PIECES + q;
        const int row = d_row[q] < left ? d_row[q] : left - 1;
        __builtin_amdgcn_global_load_lds((gvoid*)(kp + (int64_t)row * p.k_rs + d_ks[q] * 8),
                                         (lvoid*)(uintptr_t)(sl + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gvoid*)(vp + (int64_t)row * p.v_rs + d_vs[q] * 8),
                                         (lvoid*)(uintptr_t)(sl + TILEB + piece * 1024), 16, 0, 0);
      }
    }
  };

  u32x4 kreg[LD_PER_THR], vreg[LD_PER_THR];
  auto issue_loads = [&](const TileIt& t) __attribute__((always_inline)) {
    const int64_t row0 = t.crow + (int64_t)t.j * KVT;
    const bf16_t* kp = kbase + row0 * p.k_rs;          // wave-uniform base + 32-bit lane offset
    const bf16_t* vp = vbase + row0 * p.v_rs;
    const int left = t.rows - t.j * KVT;               // valid rows in this tile
    if (left >= KVT) {
#pragma unroll
      for (int it = 0; it < LD_PER_THR; ++it) {
        kreg[it] = *reinterpret_cast<const u32x4*>(kp + g_koff[it]);
        vreg[it] = *reinterpret_cast<const u32x4*>(vp + g_voff[it]);
      }
    } else {                                            // padded tail: clamp rows (masked later)
#pragma unroll
      for (int it = 0; it < LD_PER_THR; ++it) {
        const int e = tid + it * 512;
        const int row = min(e / SLOTS, left - 1), slot = e % SLOTS;
        kreg[it] = *reinterpret_cast<const u32x4*>(kp + (int64_t)row * p.k_rs + slot * 8);
        vreg[it] = *reinterpret_cast<const u32x4*>(vp + (int64_t)row * p.v_rs + slot * 8);
      }
    }
  };
  auto write_lds = [&](unsigned sl) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < LD_PER_THR; ++it) {
      *(lds_u32x4*)(uintptr_t)(sl + l_koff[it]) = kreg[it];
      *(lds_u32x4*)(uintptr_t)(sl + l_voff[it]) = vreg[it];
    }
  };

  // ---- the phases of one tile -----------------------------------------------------------------
  f32x16 s0, s1;
  auto qk_phase = [&](unsigned sl) __attribute__((always_inline)) {     // sl = LDS address of the slot
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    bf16x8 ka[DS], kb[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      const unsigned a = sl + koff[ds];
      ka[ds] = *(lds_bf16x8*)(uintptr_t)(a);
      kb[ds] = *(lds_bf16x8*)(uintptr_t)(a + 32 * ROWB);
    }
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ds], qf[ds], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[ds], qf[ds], s1, 0, 0, 0);
    }
    if (QK_AHEAD > 0) {
      // pin the issue order: K fragments run QK_AHEAD k-steps ahead of the MFMAs that consume them
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * QK_AHEAD, 0);
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (ds + QK_AHEAD < DS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
  };
  auto sm_pv_phase = [&](unsigned sl, int kv_off, bool diag, int kv_rows) __attribute__((always_inline)) {
    // key index (inside the tile) of accumulator register r: (r&3) + 8*(r>>2) + 4*hi (+32 for s1)
    const bool need_mask = (diag && kv_off + KVT - 1 > q_off) || (kv_off + KVT > kv_rows) || (kv_off < wg_last_start);
    if (need_mask) {
      const int lim_c = diag ? (my_q - kv_off) : 0x7fffffff;        // key <= lim_c visible
      const int lim = min(lim_c, kv_rows - kv_off - 1);              // key <= .. valid
      const int lo = my_start - kv_off;                              // key >= lo: same packed segment
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key > lim || key < lo) s0[r] = -INFINITY;
        if (key + 32 > lim || key + 32 < lo) s1[r] = -INFINITY;
      }
    }
    // online softmax, log2 domain
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(s0[r], s1[r]), mx);
    mx = swap32_max(mx);
    const float m_new = fmaxf(m_run, mx * scale_log2e);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], scale_log2e, -m_new));
      s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], scale_log2e, -m_new));
      psum += s0[r] + s1[r];
    }
    l_run = l_run * alpha + psum;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    // P^T operand: step t uses registers 8*(t&1).. of block t>>1 (any consistent key <-> k-slot
    // assignment is valid; the V^T fragment below uses the same one)
    bf16x8 pf[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf[0][j] = (__bf16)s0[j];
      pf[1][j] = (__bf16)s0[8 + j];
      pf[2][j] = (__bf16)s1[j];
      pf[3][j] = (__bf16)s1[8 + j];
    }
    unsigned va[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) va[db] = sl + voff[db];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const unsigned vp = va[db] + 16 * t * ROWB;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(vp));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(vp + 8 * ROWB));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 ac = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
        o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac), pf[t], o_acc[db], 0, 0, 0);
      }
    }
  };

  // ---- prologue: tile 0 -> LDS slot 0, tile 1 -> registers --------------------------------------
  TileIt cur;
  cur.c = 0; cur.j = 0; cur.n = 0; cur.rows = 0; cur.diag = 0; cur.crow = 0;
  enter_chunk(cur);
  TileIt nx1 = cur;                                   // tile t+1 (its data sits in registers)
  if (cur.c < p.n_kv_chunks) {
    if (GLDS) {
      dma_tile(cur, lds0);
      advance(nx1);
    } else {
      issue_loads(cur);
      write_lds(lds0);
      advance(nx1);
      if (nx1.c < p.n_kv_chunks) issue_loads(nx1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned long long t_top = 0, t_qk = 0, t_sm = 0, t_bar = 0, t_n = 0, tc = 0;
  auto tick = [&](unsigned long long& acc) __attribute__((always_inline)) {
    if (TIMING) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc += now - tc; tc = now; }
  };
  if (TIMING) tc = __builtin_amdgcn_s_memtime();

  int slot = 0;
  while (cur.c < p.n_kv_chunks) {
    const bool has_n1 = nx1.c < p.n_kv_chunks;
    TileIt nx2 = nx1;
    // tile t+1 -> the other LDS slot (every wave finished reading it before the last barrier)
    if (has_n1) {
      if (GLDS) {
        dma_tile(nx1, lds0 + (slot ^ 1) * SLOTB);      // lands under this tile's MFMA work
        advance(nx2);
      } else {
        write_lds(lds0 + (slot ^ 1) * SLOTB);          // registers -> LDS
        advance(nx2);
        if (nx2.c < p.n_kv_chunks) issue_loads(nx2);    // tile t+2: HBM/L2 -> registers
      }
    }
    const unsigned sl = lds0 + slot * SLOTB;
    const int kv_off = cur.j * KVT;                  // tile offset inside its chunk
    // wave-uniform skip: the whole tile lies after this wave's last query row
    const bool skip = cur.diag && kv_off > q_off + 31;
    tick(t_top);
    if (!skip) {
      qk_phase(sl);
      if (TIMING) { asm volatile("s_nop 0" :: "v"(s0[15]), "v"(s1[15])); tick(t_qk); }
      sm_pv_phase(sl, kv_off, cur.diag, cur.rows);
      if (TIMING) { asm volatile("s_nop 0" :: "v"(o_acc[DB - 1][15])); tick(t_sm); }
    }
    if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __syncthreads();
    tick(t_bar);
    ++t_n;
    slot ^= 1;
    cur = nx1;
    nx1 = nx2;
  }
  if (TIMING && lane == 0 && (wave == 0 || wave == 4)) {
    unsigned long long* g = g_attn_timing + (wave == 4 ? 8 : 0);
    atomicAdd(g + 0, t_top); atomicAdd(g + 1, t_qk); atomicAdd(g + 2, t_sm); atomicAdd(g + 3, t_bar); atomicAdd(g + 4, t_n);
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  const float l_tot = swap32_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_live) {
    const int64_t orow = (int64_t)qc * p.chunk_len + my_q;
    bf16_t* op = p.o + (int64_t)b * p.o_bs + orow * p.o_rs + (int64_t)kvh * p.o_gs + (int64_t)hq * p.o_hs;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * db + 8 * rg + 4 * hi;
        u32x2 w = {pack_bf16x2(o_acc[db][rg * 4 + 0] * inv, o_acc[db][rg * 4 + 1] * inv),
                   pack_bf16x2(o_acc[db][rg * 4 + 2] * inv, o_acc[db][rg * 4 + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = w;
      }
    }
    if (p.lse && hi == 0) {
      const float lse = l_tot > 0.f ? (m_run + log2f(l_tot)) * 0.69314718055994530942f : -INFINITY;
      p.lse[((int64_t)b * p.n_q_heads + head) * p.n_q_rows + orow] = lse;
    }
  }
}

template <int D, bool CAUSAL, int VARIANT, bool TIMING = false>
int launch_attn_v(const AttnArgs& a, int64_t nblocks, hipStream_t st) {
  constexpr int lds = 2 * 2 * KVT * D * 2;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_kernel<D, CAUSAL, VARIANT, TIMING>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL((flash_fwd_kernel<D, CAUSAL, VARIANT, TIMING>), dim3((unsigned)nblocks), dim3(512), lds, st, a);
  return vita_check_launch();
}

// VITA_ATTN_VARIANT (developer tuning aid): bits 0-3 = kernel VARIANT (default 6 = LDS-DMA + QK_AHEAD 3), bit 4 = phase timers.
inline int attn_variant() {
  static const int v = [] {
    const char* e = vita_dev_getenv("VITA_ATTN_VARIANT");
    return e ? atoi(e) : 6;
  }();
  return v;
}

template <int D, bool CAUSAL>
int launch_attn(const AttnArgs& a, int64_t nblocks, hipStream_t st) {
  const int v = attn_variant();
  if constexpr (D == 96) {
    return launch_attn_v<D, CAUSAL, 6>(a, nblocks, st);                           // LDS-DMA staging only (see the kernel's static_assert)
  } else {
    if ((v & 16) && D == 128 && CAUSAL) {
      switch (v & 15) {
        case 1: return launch_attn_v<128, true, 1, true>(a, nblocks, st);
        case 0: return launch_attn_v<128, true, 0, true>(a, nblocks, st);
        default: return launch_attn_v<128, true, 6, true>(a, nblocks, st);
      }
    }
    switch (v & 15) {
      case 1: return launch_attn_v<D, CAUSAL, 1>(a, nblocks, st);
      case 0: return launch_attn_v<D, CAUSAL, 0>(a, nblocks, st);
      default: return launch_attn_v<D, CAUSAL, 6>(a, nblocks, st);
    }
  }
}

}  // namespace

// developer aid, not part of the public ABI: copy (and optionally clear) the phase timers
extern "C" int vita_debug_attn_timing(unsigned long long* host_out16, int reset) {
  if (host_out16 && hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_attn_timing), 16 * sizeof(unsigned long long)) != hipSuccess)
    return VITA_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_timing), z, sizeof(z)) != hipSuccess) return VITA_ERR_LAUNCH;
  }
  return VITA_OK;
}

// ---- merge of two attention partials over disjoint key sets (context parallelism: own chunks first, remote chunks after the gather)
//   lse = log(exp(lse_a) + exp(lse_b)),  O = O_a exp(lse_a - lse) + O_b exp(lse_b - lse);  a part that saw no key has lse = -inf.
// HBM-bound: one thread per (row, head, 16-byte piece of d = 128); O_a / lse_a are updated in place.
namespace {
__global__ __launch_bounds__(256) void attn_merge_kernel(bf16_t* __restrict__ oa, int64_t oa_rs, int64_t oa_hs, float* __restrict__ lse_a,
                                                         const bf16_t* __restrict__ ob, int64_t ob_rs, int64_t ob_hs,
                                                         const float* __restrict__ lse_b, int64_t rows, int heads) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int piece = (int)(idx & 15);
  const int64_t rh = idx >> 4;
  if (rh >= rows * heads) return;
  const int64_t row = rh / heads;
  const int h = (int)(rh % heads);
  const float la = lse_a[(int64_t)h * rows + row], lb = lse_b[(int64_t)h * rows + row];
  const float mx = fmaxf(la, lb);
  float wa = 1.f, wb = 0.f, l = la;
  if (mx > -INFINITY) {
    const float ea = __expf(la - mx), eb = __expf(lb - mx);
    const float inv = 1.0f / (ea + eb);
    wa = ea * inv; wb = eb * inv;
    l = mx + __logf(ea + eb);
  }
  u32x4* pa = reinterpret_cast<u32x4*>(oa + row * oa_rs + (int64_t)h * oa_hs + piece * 8);
  const u32x4 a = *pa, bq = *reinterpret_cast<const u32x4*>(ob + row * ob_rs + (int64_t)h * ob_hs + piece * 8);
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    r[j] = pack_bf16x2(bf16lo_to_f32(a[j]) * wa + bf16lo_to_f32(bq[j]) * wb, bf16hi_to_f32(a[j]) * wa + bf16hi_to_f32(bq[j]) * wb);
  *pa = r;
  if (piece == 0) lse_a[(int64_t)h * rows + row] = l;
}
}  // namespace

extern "C" int vita_attn_merge(void* o_a, int64_t oa_row_stride, int64_t oa_head_stride, float* lse_a, const void* o_b,
                               int64_t ob_row_stride, int64_t ob_head_stride, const float* lse_b, int64_t rows, int heads,
                               int head_dim, void* stream) {
  if (!o_a || !o_b || !lse_a || !lse_b || rows < 0 || heads <= 0) return VITA_ERR_INVALID_ARG;
  if (head_dim != 128) return VITA_ERR_UNSUPPORTED;
  if ((oa_row_stride & 7) || (oa_head_stride & 7) || (ob_row_stride & 7) || (ob_head_stride & 7)) return VITA_ERR_UNSUPPORTED;
  if (rows == 0) return VITA_OK;
  const int64_t n = rows * heads * 16;
  hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)o_a, oa_row_stride,
                     oa_head_stride, lse_a, (const bf16_t*)o_b, ob_row_stride, ob_head_stride, lse_b, rows, heads);
  return vita_check_launch();
}

extern "C" int vita_flash_attn_fwd(const vita_attn_params* p, void* stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o) return VITA_ERR_INVALID_ARG;
  if (p->batch <= 0 || p->n_q_heads <= 0 || p->n_kv_heads <= 0 || p->chunk_len <= 0 ||
      p->n_q_chunks <= 0 || p->n_kv_chunks <= 0 || !p->q_chunk_gid || !p->kv_chunk_gid ||
      !p->kv_chunk_row)
    return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 64 && p->head_dim != 96 && p->head_dim != 128) return VITA_ERR_UNSUPPORTED;
  if (p->n_q_heads % p->n_kv_heads) return VITA_ERR_INVALID_ARG;
  if (p->n_q_chunks > kMaxChunks || p->n_kv_chunks > kMaxChunks) return VITA_ERR_UNSUPPORTED;
  if (p->chunk_len > 0x7fffff00LL) return VITA_ERR_UNSUPPORTED;
  if (p->q_valid <= 0 || p->q_valid > p->chunk_len || p->kv_valid <= 0 || p->kv_valid > p->chunk_len)
    return VITA_ERR_INVALID_ARG;
  // partial chunks only for single-chunk geometry; multi-chunk needs whole 64-key tiles
  if (p->n_q_chunks > 1 && p->q_valid != p->chunk_len) return VITA_ERR_UNSUPPORTED;
  if (p->n_kv_chunks > 1 && (p->kv_valid != p->chunk_len || p->chunk_len % KVT)) return VITA_ERR_UNSUPPORTED;
  const int64_t strides[] = {p->q_batch_stride, p->q_row_stride, p->q_head_stride, p->k_batch_stride,
                             p->k_row_stride, p->k_head_stride, p->v_batch_stride, p->v_row_stride,
                             p->v_head_stride, p->q_group_stride};
  for (int64_t s : strides)
    if (s & 7) return VITA_ERR_UNSUPPORTED;          // 16-byte vector loads
  if ((p->o_batch_stride & 3) || (p->o_row_stride & 3) || (p->o_head_stride & 3) || (p->o_group_stride & 3)) return VITA_ERR_UNSUPPORTED;

  AttnArgs a;
  a.q = (const bf16_t*)p->q; a.q_bs = p->q_batch_stride; a.q_rs = p->q_row_stride; a.q_hs = p->q_head_stride;
  a.q_gs = p->q_group_stride ? p->q_group_stride : p->q_head_stride * (p->n_q_heads / p->n_kv_heads);
  a.k = (const bf16_t*)p->k; a.k_bs = p->k_batch_stride; a.k_rs = p->k_row_stride; a.k_hs = p->k_head_stride;
  a.v = (const bf16_t*)p->v; a.v_bs = p->v_batch_stride; a.v_rs = p->v_row_stride; a.v_hs = p->v_head_stride;
  a.o = (bf16_t*)p->o; a.o_bs = p->o_batch_stride; a.o_rs = p->o_row_stride; a.o_hs = p->o_head_stride;
  a.o_gs = p->o_group_stride ? p->o_group_stride : p->o_head_stride * (p->n_q_heads / p->n_kv_heads);
  a.lse = p->lse;
  a.batch = p->batch; a.n_q_heads = p->n_q_heads; a.n_kv_heads = p->n_kv_heads;
  a.chunk_len = (int)p->chunk_len; a.q_valid = (int)p->q_valid; a.kv_valid = (int)p->kv_valid;
  a.n_q_chunks = p->n_q_chunks; a.n_kv_chunks = p->n_kv_chunks;
  a.tiles_per_q_chunk = (int)((p->chunk_len + QTILE - 1) / QTILE);
  a.n_q_rows = (int)((int64_t)(p->n_q_chunks - 1) * p->chunk_len + p->q_valid);
  a.scale_log2e = p->softmax_scale * 1.44269504088896340736f;
  a.seg_start = p->q_seg_start;
  if (p->q_seg_start && (p->n_q_chunks != 1 || p->n_kv_chunks != 1 || !p->causal || p->batch != 1)) return VITA_ERR_UNSUPPORTED;
  for (int i = 0; i < p->n_q_chunks; ++i) { a.q_gid[i] = p->q_chunk_gid[i]; a.q_order[i] = i; }
  // heaviest (largest global chunk id) first
  for (int i = 1; i < p->n_q_chunks; ++i)
    for (int j = i; j > 0 && a.q_gid[a.q_order[j]] > a.q_gid[a.q_order[j - 1]]; --j) {
      const int t = a.q_order[j]; a.q_order[j] = a.q_order[j - 1]; a.q_order[j - 1] = t;
    }
  for (int i = 0; i < p->n_kv_chunks; ++i) { a.kv_gid[i] = p->kv_chunk_gid[i]; a.kv_row[i] = p->kv_chunk_row[i]; }
  const int64_t nblocks = (int64_t)p->batch * p->n_q_heads * p->n_q_chunks * a.tiles_per_q_chunk;
  if (nblocks > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // d = 128 causal with whole 256-row / 64-key tiles: the 4 x 64-row in-wave-pipelined kernel (attn64.hip)
  if (vita_attn64_eligible(a, p->head_dim, p->causal != 0)) return vita_attn64_launch(a, nblocks, st);
  // d = 64 non-causal (the vision towers): the same structure at head size 64, ragged rows / keys (attn64v.hip)
  if (vita_attn64v_eligible(a, p->head_dim, p->causal != 0)) return vita_attn64v_launch(a, st);
  if (p->head_dim == 128) return p->causal ? launch_attn<128, true>(a, nblocks, st) : launch_attn<128, false>(a, nblocks, st);
  if (p->head_dim == 96) return p->causal ? launch_attn<96, true>(a, nblocks, st) : launch_attn<96, false>(a, nblocks, st);
  return p->causal ? launch_attn<64, true>(a, nblocks, st) : launch_attn<64, false>(a, nblocks, st);
}
