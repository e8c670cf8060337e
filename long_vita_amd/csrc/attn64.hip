// Flash attention forward, d = 128, causal, whole tiles: 4 waves x 64 query rows, one wave per SIMD (gfx950 / MI355X).
// The fast path of vita_flash_attn_fwd for the LLM prefill (plain causal and the zig-zag context-parallel chunk tables);
// attn.hip keeps every other geometry (d = 64 ViT, non-causal, ragged tails, packed samples).
//
// Why a second structure (tools/hwprobe/attn_ladder.hip, attn64.hip; profiles/r02_hwprobe_*): with 32 query rows per wave every
// K / V^T fragment read from LDS feeds ONE MFMA and two waves share a SIMD, where they hide only half of each other's VALU time;
// with 64 rows per wave a fragment feeds two MFMAs and the only way to overlap the softmax with the matrix pipe is inside one
// instruction stream — so this kernel is an in-wave software pipeline:
//   * one workgroup = 4 waves = 256 query rows of ONE query head; a wave owns 64 rows = two 32-row blocks qb; 64-key tiles
//   * register classes: O^T (2 x 4 x 16 = 128 registers) lives in AGPRs and is touched by MFMAs only (inline asm, "+a"); the
//     Q fragments (64) are pinned in AGPRs and read from there as MFMA B operands; S^T (two tiles in flight, 2 x 64) comes
//     from builtin MFMAs in VGPR form (this file is compiled with -mllvm -amdgpu-mfma-vgpr-form=1) so the softmax reads it
//     without accumulator moves; -fno-slp-vectorize keeps the fp32 softmax math out of v_pk_* (it shares the matrix datapath)
//   * phase 1: 32 MFMAs  S(t+1) = K(t+1) Q^T  ||  exp2 / row sum / bf16 pack of the last 8 - NF2 P fragments of tile t
//     phase 2: 32 MFMAs  O += V(t)^T P(t)^T   ||  row maxima of tile t+1, the running-maximum decision, its first NF2 fragments
//     the VALU work is dealt to the MFMA slots by weight; sched_barrier(0) after every slot keeps the order
//   * LAZY running maximum: the maximum (and with it O and l) only moves when some row of the wave exceeds it by more than
//     2^THR (THR = 8), so P <= 2^8 instead of <= 1 — bf16 / fp32 have the range, the relative rounding of P is unchanged — and
//     the 400-instruction accumulator rescale (AGPR -> VGPR -> AGPR) leaves the steady state.  With exact maxima a 64-row
//     wave hits it on ~20 % of the tiles of a 128K row (measured: 0.74 -> 0.90 PFLOP/s at 16K, 1.11 -> 1.19 at 128K)
//   * K / V tiles HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds: SGPR descriptor re-based per tile + the lane's 32-bit offset,
//     no address VALU), separate K and V rings of two 16 KiB slots, one barrier per tile; attn.hip's swizzled layouts
//     (conflict-free ds_read_b128 / ds_read_b64_tr_b16), the swizzle applied to the DMA's per-lane SOURCE address
//   * masks: a tile of the diagonal chunk that reaches past the workgroup's first row is masked element-wise for every wave
//     (tiles wholly past a wave's rows come out as exp2(-inf) = 0: no per-wave control flow in the pipeline)
// Reference behaviour restated: M/core/transformer/dot_product_attention.py:186-289,374-390; zig-zag chunk ownership
// M/training/utils.py:329-341.
#include "attn_args.h"
#include <stdlib.h>

namespace {

constexpr int D = 128, KVT = 64, QTILE = 256, ROWB = D * 2, TILEB = KVT * ROWB;     // 16 KiB per K (or V) tile
constexpr int LDS_K = 0;                                                             // K ring [NSLOT] | V ring [NSLOT]
constexpr int NF2 = 3;                                                               // P fragments of tile t+1 done in phase 2
constexpr int THR = 8;                                                               // lazy running maximum, log2 units

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lvoid;

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

// ---- filler schedule: units dealt to the 32 MFMA slots of a phase by cumulative weight ---------------------------------
// phase 2 units: 0..31 max3 steps (weight 1), 32..33 running-maximum decision (weight 6), then 8 * NF2 exp half-units
// (half 0: two fma + two exp2, half 1: two row-sum adds + one bf16 pack); phase 1 units: 8 * (8 - NF2) exp half-units
struct SlotMap { int first[33]; };
constexpr int unit_w2(int u) { return u < 32 ? 2 : (u < 34 ? 12 : ((u & 1) ? 6 : 8)); }
constexpr SlotMap make_map2() {
  SlotMap m{};
  const int n = 34 + 8 * NF2;
  int tot = 0;
  for (int u = 0; u < n; ++u) tot += unit_w2(u);
  int acc = 0, u = 0;
  for (int s = 0; s < 32; ++s) {
    m.first[s] = u;
    const int lim = (tot * (s + 1) + 31) / 32;
    while (u < n && acc + unit_w2(u) <= lim) { acc += unit_w2(u); ++u; }
  }
  m.first[32] = n;
  return m;
}
constexpr SlotMap make_map1() {
  SlotMap m{};
  const int n = 8 * (8 - NF2);
  for (int s = 0; s <= 32; ++s) m.first[s] = (n * s) / 32;
  return m;
}

// One kv tile position of the iteration space (all fields wave-uniform -> SGPRs); c == n_kv_chunks: end.
struct TileIt {
  int c, j, n;        // chunk, tile inside chunk, tiles to visit in this chunk
  int diag;           // chunk c is the query tile's own chunk
  const char* kp;     // first K / V row of the tile (running pointers: one 64-bit add per tile, no multiplies in the loop)
  const char* vp;
};

// NSLOT = 2: one barrier per tile (rings of two slots).  NSLOT = 4 (r03 experiment, VITA_ATTN64_RING=4): rings of four slots,
// TWO tiles between barriers — the DMAs of tiles t+2 .. t+4 are issued at the start of a pair and have two tiles to land.
// PACKED (r03): packed samples (p.seg_start: first key row of each query row's segment, non-decreasing; one chunk).  The workgroup
// starts at the (even) tile of its first row's segment, and a tile that begins before the segment of the wave's LAST row gets a
// second arithmetic mask (key >= seg_start[row]); rows whose segment starts later see such tiles as all-masked: P = 0, the running
// maximum stays at its initial -1e30 and the first visible tile rescales the (zero) state by exp2(-1e30 - m) = 0.
#ifndef VITA_ATTN64_DMA_SPREAD
#define VITA_ATTN64_DMA_SPREAD 1
#endif
template <int NSLOT, bool PACKED, bool LTILE = true>
__global__ __launch_bounds__(256, 1) void flash_fwd64_kernel(AttnArgs p) {
  constexpr int LDS_V = NSLOT * TILEB;
  constexpr bool DMA_SPREAD = VITA_ATTN64_DMA_SPREAD != 0;      // (0: the r02 - r04 burst at the top of a tile / pair, for A / B builds)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work decomposition (attn.hip's: kv head = block id % n_kv_heads = the XCD for 8 kv heads; heaviest query tiles first) ----
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
  const int q_off_wg = qti * QTILE;                 // offset of this workgroup inside its chunk
  const int q_off = q_off_wg + wave * 64;           // this wave's first row inside the chunk
  const float scale_log2e = p.scale_log2e;
  // packed samples: first tile to visit, the per-lane segment starts of the wave's rows, and the largest of them (wave-uniform)
  int seg_j0 = 0, seg_lo[2] = {0, 0}, seg_lo_max = 0;
  if constexpr (PACKED) {
    const int* ss = p.seg_start + (int64_t)qc * p.chunk_len;
    seg_j0 = (ss[q_off_wg] / KVT) & ~1;                          // even: the pipeline consumes tiles in pairs
    seg_lo[0] = ss[q_off + l31]; seg_lo[1] = ss[q_off + 32 + l31];
    seg_lo_max = __builtin_amdgcn_readfirstlane(ss[q_off + 63]);
  }

  // ---- Q fragments (B operand of S^T = K Q^T): block qb, k-step ds: row q_off + 32 qb + l31, d = 16 ds + 8 hi .. + 7 ----------
  bf16x8 qf[2][8];
  {
    const bf16_t* qp = p.q + (int64_t)b * p.q_bs + ((int64_t)qc * p.chunk_len + q_off + l31) * p.q_rs + (int64_t)kvh * p.q_gs +
                       (int64_t)hq * p.q_hs + hi * 8;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) qf[qb][ds] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)32 * qb * p.q_rs + ds * 16);
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) asm volatile("" : "+a"(qf[qb][ds]));      // live in AGPRs from here on

  // ---- LDS fragment offsets (attn.hip's layouts) ----------------------------------------------------------------------------
  unsigned koff[8], voff[4];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) koff[ds] = l31 * ROWB + (((2 * ds + hi) ^ (l31 & 15)) << 4);       // + 32 kb rows: immediate
  {
    const int g16 = lane >> 4, i16 = lane & 15, key_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      voff[db] = key_l * ROWB + (((col >> 4) ^ ((key_l & 3) << 1)) << 5) + (col & 15) * 2;
    }
  }
  // ---- LDS-DMA: wave w moves pieces 4w .. 4w+3 (1 KiB = 4 rows) of K and of V; swizzle on the SOURCE address ---------------
  unsigned dk_off[4], dv_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 4 + (lane >> 4), ps = lane & 15;
    const int ks = ps ^ (row & 15);
    const int vs = (((ps >> 1) ^ ((row & 3) << 1)) << 1) | (ps & 1);
    dk_off[q] = (unsigned)((row * p.k_rs + ks * 8) * 2);     // bytes inside the tile (64 rows x row stride < 2^32)
    dv_off[q] = (unsigned)((row * p.v_rs + vs * 8) * 2);
  }
  const char* kbase = (const char*)(p.k + (int64_t)b * p.k_bs + (int64_t)kvh * p.k_hs);
  const char* vbase = (const char*)(p.v + (int64_t)b * p.v_bs + (int64_t)kvh * p.v_hs);
  const int k_tile_bytes = (int)(p.k_rs * 2 * KVT), v_tile_bytes = (int)(p.v_rs * 2 * KVT);
  // LDS address of this wave's first piece in K / V ring slot 0.  The descriptor is re-based on the tile's first row: no
  // 4 GiB limit on the K / V buffers, no address VALU.  `opaque` keeps the 16 piece addresses from being hoisted into 16 SGPRs.
  const unsigned lds_kw = lds0 + LDS_K + wave * 4096, lds_vw = lds0 + LDS_V + wave * 4096;
  // issued from inline asm (vita_lds_dma16, vita_common.h): through the builtin, hipcc put an `s_waitcnt vmcnt(0)` in front of the first
  // V^T read of every P.V phase — the next tile's DMA, issued one phase earlier, was waited for in the middle of the current tile
  auto dma_k = [&](const TileIt& t, int slot) __attribute__((always_inline)) {
    const vita_rsrc_t r = vita_make_rsrc_uniform(t.kp);
    unsigned base = lds_kw;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) vita_lds_dma16(r, dk_off[q], base + slot * TILEB + q * 1024);
  };
  auto dma_v = [&](const TileIt& t, int slot) __attribute__((always_inline)) {
    const vita_rsrc_t r = vita_make_rsrc_uniform(t.vp);
    unsigned base = lds_vw;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) vita_lds_dma16(r, dv_off[q], base + slot * TILEB + q * 1024);
  };

  // r05: ONE piece at a time, for the spread issue inside the S = K Q^T phase (VITA_ATTN64_DMA_SPREAD): an LDS-DMA instruction holds the
  // issuing wave for ~60 cycles (MI355X_MICROARCH.md), a 32 x 32 x 16 MFMA keeps the matrix pipe busy for 32 — eight of them in a burst at
  // the top of a tile, right behind the barrier, are ~480 cycles with nothing in the pipe; one behind every fourth MFMA hides half of each.
  // Same-box A / B: 128K 147.5 - 148.0 -> 144.8 - 145.9 ms (- 1.5 ... 1.8 %), 32K - 1.7 %, 16K - 3 % (the chip returns about half of a cycle
  // saving as time: it is power-limited).  Reading the P V phase's first two V^T fragments a phase early on top of it: no further gain.
  auto dma_piece = [&](const vita_rsrc_t& r, const unsigned* off, unsigned lds_w, int slot, int q) __attribute__((always_inline)) {
    unsigned base = lds_w;
    asm volatile("" : "+s"(base));
    vita_lds_dma16(r, off[q], base + slot * TILEB + q * 1024);
  };

  // ---- tile iterator -----------------------------------------------------------------------------------------------------------
  const int tiles_per_chunk = p.chunk_len / KVT;
  auto enter_chunk = [&](TileIt& t) __attribute__((always_inline)) {   // skip chunks with nothing to visit
    while (t.c < p.n_kv_chunks) {
      const int gk = p.kv_gid[t.c];
      t.diag = gk == gq;
      t.n = gk < gq ? tiles_per_chunk : (gk > gq ? 0 : q_off_wg / KVT + 4);
      if (t.n > 0) {
        const int64_t crow = p.kv_row[t.c] + (PACKED ? seg_j0 * KVT : 0);
        t.kp = kbase + crow * p.k_rs * 2;
        t.vp = vbase + crow * p.v_rs * 2;
        t.j = PACKED ? seg_j0 : 0;                   // (t.n stays the absolute end: tiles seg_j0 .. t.n - 1)
        return;
      }
      ++t.c;
    }
  };
  auto advance = [&](TileIt& t) __attribute__((always_inline)) {
    t.kp += k_tile_bytes;
    t.vp += v_tile_bytes;
    if (++t.j == t.n) { ++t.c; enter_chunk(t); }
  };
  int n_tiles = 0;                                   // a multiple of 4 (chunk_len % 256 == 0)
  for (int c = 0; c < p.n_kv_chunks; ++c) {
    const int gk = p.kv_gid[c];
    n_tiles += gk < gq ? tiles_per_chunk : (gk > gq ? 0 : q_off_wg / KVT + 4);
  }
  if constexpr (PACKED) n_tiles -= seg_j0;           // even, >= 4: seg_start[row] <= row
  if (n_tiles == 0) {
    // a launch over REMOTE chunks only (context parallelism: the rank's own chunks are attended to before the gather lands,
    // dot_product_attention.forward_cp): these rows see none of them -> O = 0, lse = -inf, the merge ignores this part
    const u32x2 z = {0u, 0u};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int64_t orow = (int64_t)qc * p.chunk_len + q_off + 32 * qb + l31;
      bf16_t* op = p.o + (int64_t)b * p.o_bs + orow * p.o_rs + (int64_t)kvh * p.o_gs + (int64_t)hq * p.o_hs;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x2*>(op + 8 * i + 4 * hi) = z;
      if (p.lse && hi == 0) p.lse[((int64_t)b * p.n_q_heads + head) * p.n_q_rows + orow] = -INFINITY;
    }
    return;
  }

  // ---- state ----------------------------------------------------------------------------------------------------------
  f32x16 o[2][4];                                    // O^T[qb][db]: d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi, row 32 qb + l31 (AGPRs)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) asm volatile("" : "+a"(o[qb][db]));
  f32x16 sb[2][2][2];                                // S^T[parity][qb][kb]: key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][4][4];                           // packed P^T[parity][qb][frag f][4 dwords]; frag f = regs 8 (f & 1) .. of kb = f >> 1
  float m_run[2] = {-1.0e30f, -1.0e30f}, l_run[2] = {0.f, 0.f}, m_neg[2], alpha[2] = {1.f, 1.f}, mxc[4];
  // r04: the 32 probabilities a lane holds of a tile are summed into l_tile first and folded into l_run ONCE per tile.  Adding them one
  // by one stagnates on very long rows: beyond ~2 M visible keys l_run's ulp reaches the size of a single p (p ~ 2^-8 of the row
  // maximum, l_run ~ 5e4 -> ulp 4e-3), the small ones are rounded away and O / l comes out too large — 1.5e-2 at 13 M keys
  // (tools/bench_maxseq.py: one CP = 8 rank at S = 16.8 M); a tile's partial sum is 32 x larger than its terms
  float l_tile[2] = {0.f, 0.f};
  float ea = 0.f, eb = 0.f, mx0_keep = 0.f;

  constexpr SlotMap MAP1 = make_map1(), MAP2 = make_map2();

  // exp half-units (64 per tile): h -> fragment g = h >> 3 (need order of P V: g = 2 f + qb), element pair pr = (h >> 1) & 3;
  // half 0: the two fma + exp2 of the pair, half 1: row sum, bf16 pack (an exp2 result is never consumed by the next instruction)
  auto exp_half = [&](int par, int h) __attribute__((always_inline)) {
    const int g = h >> 3, pr = (h >> 1) & 3, qb = g & 1, f = g >> 1, kb = f >> 1, r = 8 * (f & 1) + 2 * pr;
    if ((h & 1) == 0) {
      ea = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][kb][r], scale_log2e, m_neg[qb]));
      eb = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][kb][r + 1], scale_log2e, m_neg[qb]));
    } else {
      if (LTILE) {
        l_tile[qb] += ea;
        l_tile[qb] += eb;
        pk[par][qb][f][pr] = pack_bf16x2(ea, eb);
        if (h == 55 || h == 63) {                    // the last half-unit of block qb (fragments g = 6 / 7): fold the tile's sum
          l_run[qb] += l_tile[qb];
          l_tile[qb] = 0.f;
        }
        asm volatile("" :: "v"(pk[par][qb][f][pr]), "v"(l_tile[qb]), "v"(l_run[qb]));            // computed HERE (no sinking past the phase)
      } else {                                       // r02-r03 (VITA_ATTN64_LTILE=0 under VITA_DEBUG: A/B timing only)
        l_run[qb] += ea;
        l_run[qb] += eb;
        pk[par][qb][f][pr] = pack_bf16x2(ea, eb);
        asm volatile("" :: "v"(pk[par][qb][f][pr]), "v"(l_run[qb]));
      }
    }
  };
  // the running-maximum decision of a tile: unit 32 keeps block 0's maximum, unit 33 decides for both blocks with ONE
  // wave-uniform flag (grow = some row exceeds its running maximum by more than 2^THR; the first tile always grows)
  auto max_unit = [&](int par, int u) __attribute__((always_inline)) {
    if (u < 32) {                                    // max3 steps: four chains (qb, kb): chain c = u & 3, step u >> 2
      const int c = u & 3, st = u >> 2, qb2 = c >> 1, kb2 = c & 1, r = 2 * st;
      const float a = sb[par][qb2][kb2][r], bb = sb[par][qb2][kb2][r + 1];
      mxc[c] = st == 0 ? fmaxf(a, bb) : fmaxf(fmaxf(a, bb), mxc[c]);
    } else if (u == 32) {
      mx0_keep = swap32_max(fmaxf(mxc[0], mxc[1])) * scale_log2e;
    } else {
      const float mx1 = swap32_max(fmaxf(mxc[2], mxc[3])) * scale_log2e;
      const bool grow = __any((mx1 > m_run[1] + (float)THR) || (mx0_keep > m_run[0] + (float)THR));
#pragma unroll
      for (int qb2 = 0; qb2 < 2; ++qb2) {
        const float mb = qb2 ? mx1 : mx0_keep;
        const float m_new = grow ? fmaxf(m_run[qb2], mb) : m_run[qb2];
        alpha[qb2] = __builtin_amdgcn_exp2f(m_run[qb2] - m_new);
        m_run[qb2] = m_new;
        m_neg[qb2] = -m_new;
        l_run[qb2] *= alpha[qb2];
      }
    }
  };
  auto k_frag = [&](unsigned kslot, int i) __attribute__((always_inline)) {        // i = 2 ds + kb
    return *(lds_bf16x8*)(uintptr_t)(kslot + koff[i >> 1] + (i & 1) * 32 * ROWB);
  };
  auto v_frag = [&](unsigned vslot, int i) __attribute__((always_inline)) {        // i = 4 t + db
    const unsigned va = vslot + voff[i & 3] + 16 * (i >> 2) * ROWB;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va + 8 * ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 ac = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, ac);
  };
  // S(buffer `dst`) = K(kslot) Q^T: slot = 4 ds + 2 kb + qb; K fragment (kb, ds) read two fragments ahead (ring of four);
  // FILL: the exp half-units 8 NF2 .. 63 of tile `par` go behind the MFMAs
  vita_rsrc_t rk_next = vita_make_rsrc_uniform(kbase), rv_next = rk_next;     // descriptors of the tiles being fetched (spread issue)
  bool spread_k = false, spread_v = false;
  int spread_k_slot = 0, spread_v_slot = 0;
  // ring of four (r06): a pair of tiles fetches FOUR tiles; the second two (K(c+4), V(c+3)) go out behind the MFMAs of the pair's first P V phase
  vita_rsrc_t rk_next2 = rk_next, rv_next2 = rk_next;
  bool spread2_k = false, spread2_v = false;
  int spread2_k_slot = 0, spread2_v_slot = 0;
  auto qk_phase = [&](int dst, unsigned kslot, bool fill, int par) __attribute__((always_inline)) {
    bf16x8 kr[4];
    kr[0] = k_frag(kslot, 0); kr[1] = k_frag(kslot, 1);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int i = s >> 1, qb = s & 1, ds = i >> 1, kb = i & 1;
      if (DMA_SPREAD && fill && (s & 3) == 1) {                    // pieces 0 .. 3 of K behind MFMAs 1, 5, 9, 13; of V behind 17, 21, 25, 29
        const int q = s >> 2;
        if (q < 4) { if (spread_k) dma_piece(rk_next, dk_off, lds_kw, spread_k_slot, q); }
        else if (spread_v) dma_piece(rv_next, dv_off, lds_vw, spread_v_slot, q - 4);
      }
      if (qb == 0 && i + 2 < 16) kr[(i + 2) & 3] = k_frag(kslot, i + 2);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sb[dst][qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[qb][ds], z, 0, 0, 0);
      } else {
        sb[dst][qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[qb][ds], sb[dst][qb][kb], 0, 0, 0);
      }
      if (fill) {
#pragma unroll
        for (int u = MAP1.first[s]; u < MAP1.first[s + 1]; ++u) exp_half(par, 8 * NF2 + u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // the rest of tile `par`'s softmax without a next tile to multiply (last tile)
  auto finish_sm = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 8 * NF2; u < 64; ++u) exp_half(par, u);
  };
  // O += V(vslot)^T P(par)^T  ||  (has_next) maxima / running-maximum decision of tile par ^ 1 and its first NF2 fragments
  auto pv_phase = [&](int par, unsigned vslot, bool has_next) __attribute__((always_inline)) {
    bf16x8 vr[4];
    vr[0] = v_frag(vslot, 0); vr[1] = v_frag(vslot, 1);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int i = s >> 1, qb = s & 1, t = i >> 2, db = i & 3;
      if (qb == 0 && i + 2 < 16) vr[(i + 2) & 3] = v_frag(vslot, i + 2);
      const u32x4 pw = {pk[par][qb][t][0], pk[par][qb][t][1], pk[par][qb][t][2], pk[par][qb][t][3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(vr[i & 3]), "v"(pf));
      if (DMA_SPREAD && NSLOT == 4 && has_next && (s & 3) == 1) {   // ring of four: pieces of K(c+4) behind MFMAs 1, 5, 9, 13; of V(c+3) behind 17 .. 29
        const int q = s >> 2;
        if (q < 4) { if (spread2_k) dma_piece(rk_next2, dk_off, lds_kw, spread2_k_slot, q); }
        else if (spread2_v) dma_piece(rv_next2, dv_off, lds_vw, spread2_v_slot, q - 4);
      }
      if (has_next) {
#pragma unroll
        for (int u = MAP2.first[s]; u < MAP2.first[s + 1]; ++u) {
          if (u < 34) max_unit(par ^ 1, u);
          else exp_half(par ^ 1, u - 34);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // causal mask of a diagonal-chunk tile (kv_off = its offset inside the chunk) in buffer `par`.  Pure VALU arithmetic
  // (s += min(lim - key, 0) * 3e38: exp2 of it is 0, a running maximum never sees it) — compare-and-select would park 64
  // lane masks in SGPR pairs and the per-register key constants in VGPRs, and that pressure spills into the steady state.
  auto mask_tile = [&](int par, int kv_off) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int base = q_off + 32 * qb + l31 - kv_off - 4 * hi;                 // key <= lim visible; key = const(kb, r) + 4 hi
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = 32 * kb + (r & 3) + 8 * (r >> 2);
          const float pen = fminf((float)(base - kc), 0.0f);
          sb[par][qb][kb][r] = fmaf(pen, 3.0e38f, sb[par][qb][kb][r]);
        }
    }
  };
  // packed samples: keys before a row's segment start (key >= seg_lo visible), same arithmetic
  auto seg_mask_tile = [&](int par, int kv_off) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int base = kv_off + 4 * hi - seg_lo[qb];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = 32 * kb + (r & 3) + 8 * (r >> 2);
          const float pen = fminf((float)(base + kc), 0.0f);
          sb[par][qb][kb][r] = fmaf(pen, 3.0e38f, sb[par][qb][kb][r]);
        }
    }
  };
  // O *= alpha (rare: only when a running maximum moved); every P V MFMA that precedes it has been issued
  auto rescale_o = [&]() __attribute__((always_inline)) {
    if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {
      // (the accumulators are operands of the wait: the compiler may not read them above it)
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));                       // asm MFMA -> accumulator read
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha[qb];
          asm volatile("" : "+a"(o[qb][db]));
        }
      asm volatile("s_nop 7" ::: "memory");                                     // accumulator write -> asm MFMA read
    }
  };
  auto needs_mask = [&](const TileIt& t) __attribute__((always_inline)) { return t.diag && t.j * KVT + KVT - 1 > q_off_wg; };
  auto masks = [&](const TileIt& t, int par) __attribute__((always_inline)) {          // wave-uniform conditions
    if (needs_mask(t)) mask_tile(par, t.j * KVT);
    if constexpr (PACKED) {
      if (t.j * KVT < seg_lo_max) seg_mask_tile(par, t.j * KVT);
    }
  };

  // ---- prologue: K(0), V(0), K(1) -> LDS; S(0); the start of its softmax -----------------------------------------------------
  TileIt cur;
  cur.c = 0; cur.j = 0; cur.n = 0; cur.diag = 0; cur.kp = kbase; cur.vp = vbase;
  enter_chunk(cur);
  TileIt nx1 = cur;
  advance(nx1);
  if constexpr (NSLOT == 2) {
    dma_k(cur, 0); dma_v(cur, 0);
    dma_k(nx1, 1);                                     // n_tiles >= 4
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk_phase(0, lds0 + LDS_K, false, 0);
    __syncthreads();                                  // every wave has read K(0): its ring slot may be refilled
    masks(cur, 0);
#pragma unroll
    for (int u = 0; u < 34 + 8 * NF2; ++u) {
      if (u < 34) max_unit(0, u);
      else exp_half(0, u - 34);
    }
    // (O is zero: no rescale for tile 0)

    // ---- main loop: two tiles per trip (the S / P buffer parity is a compile-time constant); n_tiles is a multiple of 4 ----------
    // full(par): `cur` sits in buffer par; K(t+2) -> K ring slot par, V(t+1) -> V ring slot par ^ 1; S(t+1) -> buffer par ^ 1
    auto full = [&](int par, bool more_k) __attribute__((always_inline)) {
      TileIt nx2 = nx1;
      if (DMA_SPREAD) {
        spread_k = more_k;
        if (more_k) { advance(nx2); rk_next = vita_make_rsrc_uniform(nx2.kp); spread_k_slot = par; }
        rv_next = vita_make_rsrc_uniform(nx1.vp); spread_v = true; spread_v_slot = par ^ 1;
      } else {
        if (more_k) { advance(nx2); dma_k(nx2, par); }  // K(t) in that slot was last read before the previous barrier
        dma_v(nx1, par ^ 1);
      }
      qk_phase(par ^ 1, lds0 + LDS_K + (par ^ 1) * TILEB, true, par);
      masks(nx1, par ^ 1);
      pv_phase(par, lds0 + LDS_V + par * TILEB, true);
      rescale_o();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      nx1 = nx2;
    };
    for (int t = 0; t + 2 < n_tiles; t += 2) {
      full(0, true);
      full(1, true);
    }
    full(0, false);
    finish_sm(1);                                      // last tile: the rest of its softmax, then P V
    pv_phase(1, lds0 + LDS_V + TILEB, false);
  } else {
    // ---- NSLOT = 4: tile t lives in ring slot t & 3 (K and V); call c (tile c in S / P buffer c & 1) reads K(c+1) and V(c);
    //      a PAIR of calls (c, c+1), c even, sits between two barriers and starts by fetching K(c+3), K(c+4), V(c+2), V(c+3) -------
    TileIt t2 = nx1;
    advance(t2);
    dma_k(cur, 0); dma_v(cur, 0);
    dma_k(nx1, 1); dma_v(nx1, 1);
    dma_k(t2, 2);                                      // n_tiles >= 4
    TileIt it_v = t2, it_k = t2;                      // next V / K tile to fetch: 2 / 3
    advance(it_k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk_phase(0, lds0 + LDS_K, false, 0);
    __syncthreads();                                  // every wave has read K(0): slot 0 is refilled by the first pair
    masks(cur, 0);
#pragma unroll
    for (int u = 0; u < 34 + 8 * NF2; ++u) {
      if (u < 34) max_unit(0, u);
      else exp_half(0, u - 34);
    }
    // one call: tile c in buffer par, K(c+1) in k_slot, V(c) in v_slot
    auto call = [&](int par, int k_slot, int v_slot) __attribute__((always_inline)) {
      qk_phase(par ^ 1, lds0 + LDS_K + k_slot * TILEB, true, par);
      masks(nx1, par ^ 1);
      pv_phase(par, lds0 + LDS_V + v_slot * TILEB, true);
      rescale_o();
      advance(nx1);
    };
    // pair with c = 4i + 2 b: fetches K(c+3), K(c+4) -> slots (3 + 2b) & 3, (4 + 2b) & 3 and V(c+2), V(c+3) -> slots (2 + 2b) & 3, (3 + 2b) & 3
    auto pair = [&](int b, bool k4) __attribute__((always_inline)) {
      if (DMA_SPREAD) {
        // r06: K(c+3), V(c+2) — what the NEXT pair's first call reads — go out one piece per four MFMAs inside this pair's first S phase, K(c+4),
        // V(c+3) inside its first P V phase; the second call issues nothing, so every piece has at least two phases to land before the
        // wait + barrier at the pair's end.  (r03's ring of four issued all 16 pieces in one burst at the top of the pair and was 6.8 % slower.)
        rk_next = vita_make_rsrc_uniform(it_k.kp); spread_k = true; spread_k_slot = (3 + 2 * b) & 3; advance(it_k);
        spread2_k = k4;
        if (k4) { rk_next2 = vita_make_rsrc_uniform(it_k.kp); spread2_k_slot = (4 + 2 * b) & 3; advance(it_k); }
        rv_next = vita_make_rsrc_uniform(it_v.vp); spread_v = true; spread_v_slot = (2 + 2 * b) & 3; advance(it_v);
        rv_next2 = vita_make_rsrc_uniform(it_v.vp); spread2_v = true; spread2_v_slot = (3 + 2 * b) & 3; advance(it_v);
      } else {
        dma_k(it_k, (3 + 2 * b) & 3); advance(it_k);
        if (k4) { dma_k(it_k, (4 + 2 * b) & 3); advance(it_k); }
        dma_v(it_v, (2 + 2 * b) & 3); advance(it_v);
        dma_v(it_v, (3 + 2 * b) & 3); advance(it_v);
      }
      call(0, (1 + 2 * b) & 3, (2 * b) & 3);
      spread_k = spread_v = spread2_k = spread2_v = false;
      call(1, (2 + 2 * b) & 3, (1 + 2 * b) & 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    for (int t = 0; t + 4 < n_tiles; t += 4) {        // (n_tiles / 4 - 1) x [pair A, pair B]
      pair(0, true);
      pair(1, true);
    }
    pair(0, false);                                    // the last pair: K(n_tiles) does not exist
    call(0, 3, 2);                                     // tile n_tiles - 2: K(n_tiles - 1) in slot 3, V(n_tiles - 2) in slot 2
    finish_sm(1);
    pv_phase(1, lds0 + LDS_V + 3 * TILEB, false);
  }

  // ---- epilogue: O[row][head][d] = O^T / l, lse ------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = swap32_sum(l_run[qb]);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int64_t orow = (int64_t)qc * p.chunk_len + q_off + 32 * qb + l31;
    bf16_t* op = p.o + (int64_t)b * p.o_bs + orow * p.o_rs + (int64_t)kvh * p.o_gs + (int64_t)hq * p.o_hs;
#pragma unroll
    for (int db = 0; db < 4; ++db) store_row_block32(op + 32 * db, o[qb][db], inv, hi);       // two 16-byte stores per block (r06)
    if (p.lse && hi == 0) {
      const float lse = l_tot > 0.f ? (m_run[qb] + log2f(l_tot)) * 0.69314718055994530942f : -INFINITY;
      p.lse[((int64_t)b * p.n_q_heads + head) * p.n_q_rows + orow] = lse;
    }
  }
}

}  // namespace

bool vita_attn64_eligible(const AttnArgs& a, int head_dim, bool causal) {
  if (head_dim != 128 || !causal) return false;
  if (a.seg_start && (a.n_q_chunks != 1 || a.n_kv_chunks != 1 || a.batch != 1)) return false;     // packed samples: one chunk (CP = 1)
  if (a.chunk_len % QTILE || a.q_valid != a.chunk_len || a.kv_valid != a.chunk_len) return false;
  // 16-byte output stores (r06)
  if (((uintptr_t)a.o & 15) || (a.o_rs & 7) || (a.o_hs & 7) || (a.o_gs & 7) || (a.o_bs & 7)) return false;
  // a tile's 64 rows x row stride must fit the 32-bit lane offset of the DMA
  if (a.k_rs * 2 * KVT >= (1ll << 31) || a.v_rs * 2 * KVT >= (1ll << 31)) return false;
  // (a query chunk sees whole chunks, its own up to the diagonal, or nothing: the tile count is 0 or a multiple of 4)
  const char* e = vita_dev_getenv("VITA_ATTN64");
  return !(e && e[0] == '0');
}

int vita_attn64_launch(const AttnArgs& a, int64_t nblocks, hipStream_t st) {
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TILEB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64_kernel<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
  });
  if (a.seg_start) {
    hipLaunchKernelGGL((flash_fwd64_kernel<2, true>), dim3((unsigned)nblocks), dim3(256), 4 * TILEB, st, a);
    return vita_check_launch();
  }
  const char* e = vita_dev_getenv("VITA_ATTN64_RING");           // developer A/B switch: 4 = four-slot rings, a barrier every two tiles
  const char* lt = vita_dev_getenv("VITA_ATTN64_LTILE");         // developer A/B switch: 0 = the r03 row sum (one add per probability)
  if (e && e[0] == '4')
    hipLaunchKernelGGL((flash_fwd64_kernel<4, false>), dim3((unsigned)nblocks), dim3(256), 8 * TILEB, st, a);
  else if (lt && lt[0] == '0')
    hipLaunchKernelGGL((flash_fwd64_kernel<2, false, false>), dim3((unsigned)nblocks), dim3(256), 4 * TILEB, st, a);
  else
    hipLaunchKernelGGL((flash_fwd64_kernel<2, false>), dim3((unsigned)nblocks), dim3(256), 4 * TILEB, st, a);
  return vita_check_launch();
}
