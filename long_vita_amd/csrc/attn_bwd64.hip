// Flash attention backward, dQ, d = 128, causal, whole tiles: 4 waves x 64 query rows, one wave per SIMD (gfx950 / MI355X).
// The fast path of vita_flash_attn_bwd's dQ pass; attn_bwd.hip keeps every other geometry (ragged chunks, packed samples of ragged length).
//
// Same recipe as the forward's attn64.hip, for the three GEMMs of the dQ pass:
//     S^T = K Q^T,   dP^T = V dO^T,   P^T = exp2(S^T c - lse),   dS^T = P^T o (dP^T - delta) scale,   dQ^T += K^T dS^T
//   * a wave owns 64 query rows (two 32-row blocks qb): every K / V / K^T fragment read from LDS feeds TWO MFMAs (attn_bwd.hip's
//     32 rows per wave need 1 KiB of LDS reads per MFMA, which saturates the LDS port at half the matrix rate);
//   * register classes: dQ^T (2 x 4 x 16 = 128 registers) lives in AGPRs and is touched by inline-asm MFMAs only; the wave's Q and dO
//     fragments (64 + 64) are pinned in AGPRs and read from there as MFMA B operands — the AGPR file is full; S^T and dP^T come
//     from builtin MFMAs in VGPR form (-mllvm -amdgpu-mfma-vgpr-form=1) so the VALU reads them without accumulator moves;
//   * the pipeline runs on HALF tiles (32 keys; u = 2 t + kb), which is what lets two generations of S^T and dP^T fit in 128 VGPRs:
//         trip u:  slots  0..15  S^T(u+1)  MFMAs        ||  dS^T(u) = P (dP scale - delta scale), bf16 pack      (5 VALU / slot)
//                  slots 16..31  dP^T(u+1) MFMAs        ||  P^T(u+1) = exp2(S^T(u+1) c - lse)                    (1 pair / 2 slots)
//                  slots 32..47  dQ^T += K^T(u) dS^T(u) ||  (the rest of) P^T(u+1)
//     sched_barrier(0) after every slot keeps program order = issue order;
//   * K (fragment layout), V (fragment layout) rings of three 16 KiB slots, K (transposed layout) ring of two; tile t+2's K / V and
//     tile t+1's K^T image are fetched by LDS-DMA (from inline asm, vita_common.h) at the start of iteration t; one barrier per tile;
//   * masks: a tile of the diagonal chunk that reaches past the workgroup's first row is masked arithmetically for every wave.
// Reference behaviour restated: the autograd of M/core/transformer/dot_product_attention.py:186-289 (flash-attn / TE backward).
#include "attn_bwd_args.h"
#include <stdlib.h>

#ifndef DQ64_DMA_SPREAD
#define DQ64_DMA_SPREAD 1
#endif

namespace {
constexpr bool DMA_SPREAD = DQ64_DMA_SPREAD != 0;     // (0: the eight pieces of a tile in a burst at the top of the iteration, for A / B builds)

constexpr int D = 128, KVT = 64, QTILE = 256, ROWB = D * 2, TILEB = KVT * ROWB;       // 16 KiB per image of a 64-key tile
// r04: ONE image per K tile serves the fragment reads (S^T = K Q^T) and the transposed reads (dQ^T += K^T dS^T): 16-byte slots XOR-ed with
// swz(row) = ((row & 3) << 2) | ((row >> 2) & 3) — conflict-free for both access patterns (attn_bwd_kv64.hip's header); r02-r03 staged
// a second, differently swizzled K image (48 -> 32 KB of LDS-DMA per tile)
constexpr int LDS_KF = 0, LDS_VF = 3 * TILEB, LDS_BYTES = 6 * TILEB;   // K [3] | V frag [3] = 96 KiB

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;

struct TileIt {
  int c, j, n;        // chunk, tile inside chunk, tiles to visit in this chunk; c == n_kv_chunks: end
  int diag;           // chunk c is the query tile's own chunk
  const char* kp;     // first K / V row of the tile
  const char* vp;
};

// PACKED (r03): packed samples (p.seg_start, one chunk) — the workgroup starts at the tile of its first row's segment; halves that begin
// before the segment of the wave's last row get a second arithmetic mask (key >= seg_start[row]), as in attn64.hip.
template <bool PACKED>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int G = p.n_q_heads / p.n_kv_heads;
  const int tiles_per_chunk = p.chunk_len / QTILE;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int hq = bid % G; bid /= G;
  const int n_qt = p.n_q_chunks * tiles_per_chunk;
  const int qt_rev = n_qt - 1 - bid;                   // heaviest query tiles first
  const int qc = qt_rev / tiles_per_chunk;
  const int qti = qt_rev % tiles_per_chunk;
  const int gq = p.q_gid[qc];
  const int head = kvh * G + hq;
  const int q_off_wg = qti * QTILE;
  const int q_off = q_off_wg + wave * 64;
  const float scale_log2e = p.scale_log2e, scale = p.scale;
  int seg_j0 = 0, seg_lo[2] = {0, 0}, seg_lo_max = 0;
  if constexpr (PACKED) {
    const int* ss = p.seg_start + (int64_t)qc * p.chunk_len;
    seg_j0 = ss[q_off_wg] / KVT;
    seg_lo[0] = ss[q_off + l31]; seg_lo[1] = ss[q_off + 32 + l31];
    seg_lo_max = __builtin_amdgcn_readfirstlane(ss[q_off + 63]);
  }

  // ---- the wave's own rows: Q and dO fragments (MFMA B operands: row q_off + 32 qb + l31, d = 16 ds + 8 hi .. + 7), lse, delta ------
  bf16x8 qf[2][8], dof[2][8];
  float lse2[2], dlt_s[2];
  {
    const int64_t row0 = (int64_t)qc * p.chunk_len + q_off + l31;
    const bf16_t* qp = p.q + row0 * p.q_rs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
    const bf16_t* dp_ = p.d_o + row0 * p.do_rs + (int64_t)head * p.do_hs + hi * 8;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        qf[qb][ds] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)32 * qb * p.q_rs + ds * 16);
        dof[qb][ds] = *reinterpret_cast<const bf16x8*>(dp_ + (int64_t)32 * qb * p.do_rs + ds * 16);
      }
      lse2[qb] = p.lse[(int64_t)head * p.n_q_rows + row0 + 32 * qb] * 1.44269504088896340736f;
      dlt_s[qb] = p.delta[(int64_t)head * p.n_q_rows + row0 + 32 * qb] * scale;
    }
  }
  // (consumed here: the compiler waits for the loads HERE, not inside the loop where its vmcnt would also wait for DMA in flight)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) { asm volatile("" : "+a"(qf[qb][ds])); asm volatile("" : "+a"(dof[qb][ds])); }
    asm volatile("" : "+v"(lse2[qb]), "+v"(dlt_s[qb]));
  }

  // ---- LDS fragment offsets (attn.hip's layouts) -----------------------------------------------------------------------------------
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  unsigned koff[8], toff[4], toff8[4];               // toff8: the second transposed read, 8 keys further down
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) koff[ds] = l31 * ROWB + (((2 * ds + hi) ^ swz(l31 & 15)) << 4);       // + 32 kb rows: immediate
  {
    const int g16 = lane >> 4, i16 = lane & 15, key_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      toff[db] = key_l * ROWB + (((col >> 3) ^ swz(key_l)) << 4) + (col & 7) * 2;
      toff8[db] = (key_l + 8) * ROWB + (((col >> 3) ^ swz(key_l + 8)) << 4) + (col & 7) * 2;
    }
  }
  // ---- LDS-DMA: wave w moves pieces 4w .. 4w+3 (1 KiB = 4 rows) of each image; swizzles on the SOURCE address --------------------
  unsigned off_kf[4], off_vf[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 4 + (lane >> 4), ps = lane & 15;
    const int fs = ps ^ swz(row & 15);
    off_kf[q] = (unsigned)((row * p.k_rs + fs * 8) * 2);
    off_vf[q] = (unsigned)((row * p.v_rs + fs * 8) * 2);
  }
  const char* kbase = (const char*)(p.k + (int64_t)kvh * p.k_hs);
  const char* vbase = (const char*)(p.v + (int64_t)kvh * p.v_hs);
  const int k_tile_bytes = (int)(p.k_rs * 2 * KVT), v_tile_bytes = (int)(p.v_rs * 2 * KVT);
  const unsigned lds_w = lds0 + wave * 4096;
  auto dma_kv = [&](const TileIt& t, int slot3) __attribute__((always_inline)) {                // K, V fragment images -> ring slot
    const vita_rsrc_t rk = vita_make_rsrc_uniform(t.kp), rv = vita_make_rsrc_uniform(t.vp);
    unsigned base = lds_w + slot3 * TILEB;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vita_lds_dma16(rk, off_kf[q], base + LDS_KF + q * 1024);
      vita_lds_dma16(rv, off_vf[q], base + LDS_VF + q * 1024);
    }
  };

  // one piece of the same (r05, DQ64_DMA_SPREAD: issued one per MFMA slot behind the first MFMAs of the tile instead of eight in a burst behind
  // the barrier, as attn64.hip / attn_bwd_kvp.hip): i = 0 .. 3 K line i, 4 .. 7 V line i - 4
  auto dma_kv_piece = [&](const TileIt& t, int slot3, int i) __attribute__((always_inline)) {
    unsigned base = lds_w + slot3 * TILEB;
    asm volatile("" : "+s"(base));
    if (i < 4) vita_lds_dma16(vita_make_rsrc_uniform(t.kp), off_kf[i], base + LDS_KF + i * 1024);
    else vita_lds_dma16(vita_make_rsrc_uniform(t.vp), off_vf[i - 4], base + LDS_VF + (i - 4) * 1024);
  };

  // ---- tile iterator (as attn64.hip) ---------------------------------------------------------------------------------------------
  const int kv_tiles_per_chunk = p.chunk_len / KVT;
  auto enter_chunk = [&](TileIt& t) __attribute__((always_inline)) {
    while (t.c < p.n_kv_chunks) {
      const int gk = p.kv_gid[t.c];
      t.diag = gk == gq;
      t.n = gk < gq ? kv_tiles_per_chunk : (gk > gq ? 0 : q_off_wg / KVT + 4);
      if (t.n > 0) {
        const int64_t crow = p.kv_row[t.c] + (PACKED ? seg_j0 * KVT : 0);
        t.kp = kbase + crow * p.k_rs * 2;
        t.vp = vbase + crow * p.v_rs * 2;
        t.j = PACKED ? seg_j0 : 0;                   // (t.n stays the absolute end)
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
  int n_tiles = 0;                                   // a multiple of 4, >= 4 (chunk_len % 256 == 0, the diagonal chunk is present)
  for (int c = 0; c < p.n_kv_chunks; ++c) {
    const int gk = p.kv_gid[c];
    n_tiles += gk < gq ? kv_tiles_per_chunk : (gk > gq ? 0 : q_off_wg / KVT + 4);
  }
  if constexpr (PACKED) n_tiles -= seg_j0;           // >= 4: seg_start[row] <= row

  // ---- state ---------------------------------------------------------------------------------------------------------------------
  f32x16 o[2][4];                                    // dQ^T[qb][db] (AGPRs)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
      asm volatile("" : "+a"(o[qb][db]));
    }
  f32x16 sb[2][2], dpb[2][2];                        // S^T / dP^T of a half tile [parity][qb]: key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][2][4];                           // packed dS^T[parity][qb][k-step t'][4 dwords]

  // P^T of pair e (qb = e >> 3, registers 2 (e & 7), +1) of buffer `par`, in place
  auto part1_pair = [&](int par, int e) __attribute__((always_inline)) {
    const int qb = e >> 3, r = 2 * (e & 7);
    sb[par][qb][r] = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][r], scale_log2e, -lse2[qb]));
    sb[par][qb][r + 1] = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][r + 1], scale_log2e, -lse2[qb]));
  };
  // dS^T of pair e, packed
  auto part2_pair = [&](int par, int e) __attribute__((always_inline)) {
    const int qb = e >> 3, pr = e & 7, r = 2 * pr;
    const float a = sb[par][qb][r] * fmaf(dpb[par][qb][r], scale, -dlt_s[qb]);
    const float b = sb[par][qb][r + 1] * fmaf(dpb[par][qb][r + 1], scale, -dlt_s[qb]);
    pk[par][qb][pr >> 2][pr & 3] = pack_bf16x2(a, b);
    asm volatile("" :: "v"(pk[par][qb][pr >> 2][pr & 3]));                      // computed HERE (no sinking past the phase)
  };
  auto frag = [&](unsigned slot_addr, int ds, int kb) __attribute__((always_inline)) {
    return *(lds_bf16x8*)(uintptr_t)(slot_addr + koff[ds] + kb * 32 * ROWB);
  };
  auto tr_frag = [&](unsigned slot_addr, int t4, int db) __attribute__((always_inline)) {      // keys 16 t4 .. + 15, d block db
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(slot_addr + toff[db] + 16 * t4 * ROWB));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(slot_addr + toff8[db] + 16 * t4 * ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 ac = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, ac);
  };
  // arithmetic causal mask of half (kv_off = first key of the half inside the chunk) in buffer `par` (see attn64.hip)
  auto mask_half = [&](int par, int kv_off) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int base = q_off + 32 * qb + l31 - kv_off - 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = (r & 3) + 8 * (r >> 2);
        const float pen = fminf((float)(base - kc), 0.0f);
        sb[par][qb][r] = fmaf(pen, 3.0e38f, sb[par][qb][r]);
      }
    }
  };
  auto seg_mask_half = [&](int par, int kv_off) __attribute__((always_inline)) {       // packed samples: key >= seg_lo is visible
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int base = kv_off + 4 * hi - seg_lo[qb];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = (r & 3) + 8 * (r >> 2);
        const float pen = fminf((float)(base + kc), 0.0f);
        sb[par][qb][r] = fmaf(pen, 3.0e38f, sb[par][qb][r]);
      }
    }
  };
  // slots 0..15: S^T of the next half (kb_n of the tile at kf) into buffer par ^ 1; FILL: dS(par) behind them
  auto s_group = [&](int par, unsigned kf, int kb_n, bool fill, const TileIt* dma_t = nullptr, int dma_slot = 0) __attribute__((always_inline)) {
    bf16x8 fr[4];
    fr[0] = frag(kf, 0, kb_n); fr[1] = frag(kf, 1, kb_n);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, qb = s & 1;
      if (qb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(kf, ds + 2, kb_n);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sb[par ^ 1][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], qf[qb][ds], z, 0, 0, 0);
      } else {
        sb[par ^ 1][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], qf[qb][ds], sb[par ^ 1][qb], 0, 0, 0);
      }
      if (fill) part2_pair(par, s);
      if (DQ64_DMA_SPREAD == 2) { if (dma_t && (s & 1) == 1) dma_kv_piece(*dma_t, dma_slot, s >> 1); }      // every other slot, 1 .. 15 (A / B)
      else if (dma_t && s >= 1 && s <= 8) dma_kv_piece(*dma_t, dma_slot, s - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // slots 16..31: dP^T of the next half into buffer par ^ 1; FILL: the first 8 pairs of P(par ^ 1)
  auto p_group = [&](int par, unsigned vf, int kb_n, bool fill) __attribute__((always_inline)) {
    bf16x8 fr[4];
    fr[0] = frag(vf, 0, kb_n); fr[1] = frag(vf, 1, kb_n);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, qb = s & 1;
      if (qb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(vf, ds + 2, kb_n);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        dpb[par ^ 1][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], dof[qb][ds], z, 0, 0, 0);
      } else {
        dpb[par ^ 1][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], dof[qb][ds], dpb[par ^ 1][qb], 0, 0, 0);
      }
      if (fill && (s & 1) == 0) part1_pair(par ^ 1, s >> 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto sp_group = [&](int par, unsigned kf, unsigned vf, int kb_n, bool fill, bool masked, int mask_off, const TileIt* dma_t = nullptr,
                      int dma_slot = 0) __attribute__((always_inline)) {
    s_group(par, kf, kb_n, fill, dma_t, dma_slot);
    if (masked) mask_half(par ^ 1, mask_off);          // wave-uniform, diagonal tiles only; between the groups, not inside one
    if constexpr (PACKED) {
      if (mask_off < seg_lo_max) seg_mask_half(par ^ 1, mask_off);
    }
    p_group(par, vf, kb_n, fill);
  };
  // slots 32..47: dQ^T += K^T(half kb of the tile at kt) dS^T(par); FILL: pairs 8..15 of P(par ^ 1)
  auto dq_group = [&](int par, unsigned kt, int kb, bool fill) __attribute__((always_inline)) {
    bf16x8 tr[4];
    tr[0] = tr_frag(kt, 2 * kb, 0); tr[1] = tr_frag(kt, 2 * kb, 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int i = s >> 1, qb = s & 1, t2 = i >> 2, db = i & 3;
      if (qb == 0 && i + 2 < 8) tr[(i + 2) & 3] = tr_frag(kt, 2 * kb + ((i + 2) >> 2), (i + 2) & 3);
      const u32x4 pw = {pk[par][qb][t2][0], pk[par][qb][t2][1], pk[par][qb][t2][2], pk[par][qb][t2][3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(tr[i & 3]), "v"(pf));
      if (fill && (s & 1) == 0) part1_pair(par ^ 1, 8 + (s >> 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto needs_mask = [&](const TileIt& t) __attribute__((always_inline)) { return t.diag && t.j * KVT + KVT - 1 > q_off_wg; };

  // ---- prologue: tile 0 (all three images) and tile 1 (K, V) -> LDS; S^T / dP^T / P^T of half 0 -------------------------------------
  TileIt cur;
  cur.c = 0; cur.j = 0; cur.n = 0; cur.diag = 0; cur.kp = kbase; cur.vp = vbase;
  enter_chunk(cur);
  TileIt nx1 = cur;
  advance(nx1);                                      // n_tiles >= 4
  TileIt nx2 = nx1;
  advance(nx2);
  dma_kv(cur, 0); dma_kv(nx1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  sp_group(1, lds0 + LDS_KF, lds0 + LDS_VF, 0, false, needs_mask(cur), cur.j * KVT);           // -> buffers 0
#pragma unroll
  for (int e = 0; e < 16; ++e) part1_pair(0, e);

  // ---- main loop: one tile = two trips (half kb = buffer parity kb) ----------------------------------------------------------------
  int s3 = 0, s3n = 1, s3nn = 2;                     // ring slots of tiles t, t+1, t+2 in the K / V fragment rings
  int tpar = 0;                                      // t & 1: the K^T ring slot of tile t
  // has1 / has2 (tile t+1 / t+2 exists) are compile-time constants of each call: no data-dependent branch inside the pipeline
  auto iteration = [&](const bool has1, const bool has2) __attribute__((always_inline)) {
    if (!DMA_SPREAD && has2) dma_kv(nx2, s3nn);      // that slot held tile t-1 (last read before the previous barrier)
    const unsigned kf = lds0 + LDS_KF + s3 * TILEB, vf = lds0 + LDS_VF + s3 * TILEB, kt = kf;      // K^T out of tile t's K image
    const unsigned kfn = lds0 + LDS_KF + s3n * TILEB, vfn = lds0 + LDS_VF + s3n * TILEB;
    // trip A: u = 2 t (buffers 0): next half = (tile t, kb 1)
    sp_group(0, kf, vf, 1, true, needs_mask(cur), cur.j * KVT + 32, DMA_SPREAD && has2 ? &nx2 : nullptr, s3nn);
    dq_group(0, kt, 0, true);
    // trip B: u = 2 t + 1 (buffers 1): next half = (tile t+1, kb 0)
    if (has1) {
      sp_group(1, kfn, vfn, 0, true, needs_mask(nx1), nx1.j * KVT);
      dq_group(1, kt, 1, true);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) part2_pair(1, e);
      // VALU result -> inline-asm MFMA operand: wait states the compiler does not know are needed, tied to the operands
      asm volatile("s_nop 4" : "+v"(pk[1][0][0][0]), "+v"(pk[1][0][0][1]), "+v"(pk[1][0][0][2]), "+v"(pk[1][0][0][3]),
                   "+v"(pk[1][1][0][0]), "+v"(pk[1][1][0][1]), "+v"(pk[1][1][0][2]), "+v"(pk[1][1][0][3]));
      dq_group(1, kt, 1, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = nx1; nx1 = nx2;
    if (has2) advance(nx2);
    const int tmp = s3; s3 = s3n; s3n = s3nn; s3nn = tmp;
    tpar ^= 1;
  };
  for (int t = 0; t + 2 < n_tiles; ++t) iteration(true, true);
  iteration(true, false);
  iteration(false, false);

  // ---- epilogue: dQ[row][head][d] = dQ^T -------------------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]),
               "+a"(o[1][2]), "+a"(o[1][3]));
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int64_t orow = (int64_t)qc * p.chunk_len + q_off + 32 * qb + l31;
    bf16_t* op = p.dq + orow * p.dq_rs + (int64_t)kvh * p.dq_gs + (int64_t)hq * p.dq_hs;
#pragma unroll
    for (int db = 0; db < 4; ++db) store_row_block32(op + 32 * db, o[qb][db], 1.0f, hi);      // two 16-byte stores per block (r06)
  }
}

}  // namespace

bool vita_attn_bwd_dq64_eligible(const BwdArgs& a) {
  if (a.head_dim != 128) return false;                       // the 64-rows-per-wave kernels are built for d = 128
  if (a.chunk_len % QTILE) return false;
  if (((uintptr_t)a.dq & 15) || (a.dq_rs & 7) || (a.dq_hs & 7) || (a.dq_gs & 7)) return false;                      // 16-byte output stores (r06)
  if (a.seg_start && (a.n_q_chunks != 1 || a.n_kv_chunks != 1)) return false;      // packed samples: one chunk
  for (int i = 0; i < a.n_q_chunks; ++i) {           // every query chunk meets its own keys (the diagonal) in this launch
    bool found = false;
    for (int j = 0; j < a.n_kv_chunks; ++j) found = found || a.kv_gid[j] == a.q_gid[i];
    if (!found) return false;
  }
  if ((int64_t)KVT * a.k_rs * 2 > 0x7fffffffLL || (int64_t)KVT * a.v_rs * 2 > 0x7fffffffLL) return false;
  const char* e = vita_dev_getenv("VITA_ATTN_BWD64");
  return !(e && e[0] == '0');
}

int vita_attn_bwd_dq64_launch(const BwdArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  const int64_t n = (int64_t)a.n_q_heads * a.n_q_chunks * (a.chunk_len / QTILE);
  if (n > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  if (a.seg_start) hipLaunchKernelGGL(attn_bwd_dq64_kernel<true>, dim3((unsigned)n), dim3(256), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(attn_bwd_dq64_kernel<false>, dim3((unsigned)n), dim3(256), LDS_BYTES, st, a);
  return vita_check_launch();
}
