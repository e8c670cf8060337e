// Flash attention backward, dK and dV, d = 128, causal, whole tiles: 4 waves x 64 KEYS, one wave per SIMD (gfx950 / MI355X).
// The mirror image of attn_bwd64.hip's dQ kernel: a wave owns 64 keys of one kv head and walks the query tiles (64 rows) that can
// see them, for every query head of the group.  attn_bwd.hip's dK/dV kernel (32 keys per wave, both gradients at once) needs 1 KiB
// of LDS reads per MFMA; at 64 keys per wave the accumulators of dK AND dV (256 registers) no longer fit next to the wave's own
// K / V fragments, so the pass is split into two launches of this template:
//     IS_DK:   S = Q K^T,  dP = dO V^T,  P = exp2(S c - lse),  dS = P o (dP - delta) scale,  dK^T += Q^T dS      (3 GEMM units)
//     !IS_DK:  S = Q K^T,                P = exp2(S c - lse),                                 dV^T += dO^T P      (2 GEMM units)
// 8 units instead of 7, each at half the LDS traffic and with the forward's in-stream schedule.
//   * register classes: the gradient accumulator (2 x 4 x 16 = 128 registers) lives in AGPRs (inline-asm MFMAs); the wave's own K
//     (and V) fragments are pinned in AGPRs and read from there as MFMA B operands; S and dP come from builtin MFMAs in VGPR form;
//   * half-tile pipeline (32 query rows; u = 2 t + qh), as attn_bwd64.hip:
//         trip u:  S(u+1) MFMAs || dS(u) / pack;   dP(u+1) MFMAs || P(u+1) = exp2(...);   gradient MFMAs of half u || rest of P(u+1)
//   * lse / delta belong to the TILE's rows here (they vary along the accumulator registers, not along lanes): each tile's 64 + 64
//     values are staged next to its images and read as 4-float vectors right before the half that needs them;
//   * images per tile (r04): IS_DK — Q and dO, ONE image each: the fragment reads (ds_read_b128, S and dP) and the transposed reads
//     (ds_read_b64_tr_b16, dK^T += Q^T dS) of Q come from the same LDS image.  Its 16-byte slots are XOR-ed with
//     swz(row) = ((row & 3) << 2) | ((row >> 2) & 3): 16 consecutive rows land in 16 different slots (fragment reads), and the
//     four rows a 32-lane half of a transposed read touches land in four different 64-byte bank groups — both conflict-free, where
//     r02-r03 staged a second, differently swizzled copy of the Q tile (48 -> 32 KB of LDS-DMA per tile, 8 -> 6 tiles of LDS);
//     !IS_DK — Q fragment layout + dO transposed layout (two tensors).  Rings of 3 (2 for dO^T), LDS-DMA from inline asm, one barrier per tile.
// Reference behaviour restated: the autograd of M/core/transformer/dot_product_attention.py:186-289 (flash-attn / TE backward).
#include "attn_bwd_args.h"
#include <stdlib.h>

namespace {

constexpr int D = 128, QT = 64, KTILE = 256, ROWB = D * 2, TILEB = QT * ROWB;           // 16 KiB per image of a 64-row query tile
constexpr float LOG2E = 1.44269504088896340736f;

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;

struct QTileIt {
  int hq, c, j, jend;   // query head of the group, query chunk, tile inside the chunk, one past the last tile; hq == G: end
  int diag;             // chunk c is the key block's own chunk
  const char* qp;       // first Q / dO row of the tile, lse / delta of its first row
  const char* dop;
  const char* lp;
  const char* dlp;
};

// PACKED (r03): packed samples (p.seg_end: one past the last row of each KEY's segment, non-decreasing; one chunk) — the walk over the
// query tiles ends with the segment of the workgroup's last key; halves that reach past the segment end of the wave's FIRST key get a
// second arithmetic mask (row < seg_end[key] is visible).
template <bool IS_DK, bool PACKED>
__global__ __launch_bounds__(256, 1) void attn_bwd_kv64_kernel(BwdArgs p) {
  // LDS: ring A [3] (Q) | IS_DK: ring B [3] (dO) / !IS_DK: transposed ring [2] (dO^T) | stats [3] x 512 B
  constexpr int LDS_FA = 0, LDS_FB = 3 * TILEB, LDS_TR = 3 * TILEB, LDS_ST = IS_DK ? 6 * TILEB : 5 * TILEB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int G = p.n_q_heads / p.n_kv_heads;
  const int kt_per_chunk = p.chunk_len / KTILE;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int kc = bid / kt_per_chunk;                    // kv chunk (buffer order)
  const int kti = bid % kt_per_chunk;
  const int gk = p.kv_gid[kc];
  const int k_off_wg = kti * KTILE;                     // first key of the workgroup inside its chunk
  const int k_off = k_off_wg + wave * 64;               // this wave's first key inside the chunk
  const int64_t k_row0 = p.kv_row[kc] + k_off;          // its row in the K / V buffers
  const float scale_log2e = p.scale_log2e, scale = p.scale;
  int seg_jend = 0, seg_hi[2] = {0, 0}, seg_hi_min = 0;
  if constexpr (PACKED) {
    const int* se = p.seg_end + p.kv_row[kc];
    seg_jend = (se[k_off_wg + KTILE - 1] + QT - 1) / QT;         // >= k_off_wg / QT + 4: seg_end[key] > key
    seg_hi[0] = se[k_off + l31]; seg_hi[1] = se[k_off + 32 + l31];
    seg_hi_min = __builtin_amdgcn_readfirstlane(se[k_off]);
  }

  // ---- the wave's own keys: K (and V) fragments as MFMA B operands (key k_off + 32 kb + l31, d = 16 ds + 8 hi .. + 7) ------------------
  bf16x8 kf[2][8], vf[2][8];
  {
    const bf16_t* kp = p.k + (int64_t)kvh * p.k_hs + (k_row0 + l31) * p.k_rs + hi * 8;
    const bf16_t* vp = p.v + (int64_t)kvh * p.v_hs + (k_row0 + l31) * p.v_rs + hi * 8;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        kf[kb][ds] = *reinterpret_cast<const bf16x8*>(kp + (int64_t)32 * kb * p.k_rs + ds * 16);
        if (IS_DK) vf[kb][ds] = *reinterpret_cast<const bf16x8*>(vp + (int64_t)32 * kb * p.v_rs + ds * 16);
      }
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {                     // consumed (loads waited for) and pinned in AGPRs here
      asm volatile("" : "+a"(kf[kb][ds]));
      if (IS_DK) asm volatile("" : "+a"(vf[kb][ds]));
    }

  // ---- LDS fragment offsets (attn.hip's layouts) ---------------------------------------------------------------------------------------
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };      // the slot XOR of the dual-use image (header)
  unsigned foff[8], toff[4], toff8[4];               // toff8: the second transposed read, 8 rows further down
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) foff[ds] = l31 * ROWB + (((2 * ds + hi) ^ swz(l31 & 15)) << 4);       // + 32 qh rows: immediate
  {
    const int g16 = lane >> 4, i16 = lane & 15, row_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      if (IS_DK) {                                   // Q^T out of the fragment image
        toff[db] = row_l * ROWB + (((col >> 3) ^ swz(row_l)) << 4) + (col & 7) * 2;
        toff8[db] = (row_l + 8) * ROWB + (((col >> 3) ^ swz(row_l + 8)) << 4) + (col & 7) * 2;
      } else {                                       // dO^T out of its own transposed-layout image (32-byte chunk ^ 2 (row & 3))
        toff[db] = row_l * ROWB + (((col >> 4) ^ ((row_l & 3) << 1)) << 5) + (col & 15) * 2;
        toff8[db] = toff[db] + 8 * ROWB;
      }
    }
  }
  // ---- LDS-DMA: wave w moves pieces 4w .. 4w+3 (1 KiB = 4 rows) of each image; swizzles on the SOURCE address ------------------------
  unsigned off_qf[4], off_qt[4], off_df[4], off_dt[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 4 + (lane >> 4), ps = lane & 15;
    const int fs = ps ^ swz(row & 15);
    const int ts = (((ps >> 1) ^ ((row & 3) << 1)) << 1) | (ps & 1);
    off_qf[q] = (unsigned)((row * p.q_rs + fs * 8) * 2);
    off_qt[q] = (unsigned)((row * p.q_rs + ts * 8) * 2);
    off_df[q] = (unsigned)((row * p.do_rs + fs * 8) * 2);
    off_dt[q] = (unsigned)((row * p.do_rs + ts * 8) * 2);
  }
  const unsigned lds_w = lds0 + wave * 4096;
  auto dma_frag = [&](const QTileIt& t, int slot3) __attribute__((always_inline)) {        // fragment images + statistics -> ring slot
    const vita_rsrc_t rq = vita_make_rsrc_uniform(t.qp);
    unsigned base = lds_w + slot3 * TILEB;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) vita_lds_dma16(rq, off_qf[q], base + LDS_FA + q * 1024);
    if (IS_DK) {
      const vita_rsrc_t rd = vita_make_rsrc_uniform(t.dop);
#pragma unroll
      for (int q = 0; q < 4; ++q) vita_lds_dma16(rd, off_df[q], base + LDS_FB + q * 1024);
    }
    // 64 lse (wave 0) / 64 delta (wave 1, IS_DK) of the tile's rows: lane -> row
    if (wave == 0) vita_lds_dma4(vita_make_rsrc_uniform(t.lp), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512);
    if (IS_DK && wave == 1) vita_lds_dma4(vita_make_rsrc_uniform(t.dlp), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512 + 256);
  };
  auto dma_tr = [&](const QTileIt& t, int slot2) __attribute__((always_inline)) {          // !IS_DK: transposed-layout image of dO
    if (IS_DK) return;
    const vita_rsrc_t r = vita_make_rsrc_uniform(t.dop);
    unsigned base = lds_w + LDS_TR + slot2 * TILEB;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) vita_lds_dma16(r, off_dt[q], base + q * 1024);
  };

  // ---- iteration space: (query head of the group) x (query chunks that see this key block) x (64-row tiles) ---------------------------
  const int q_tiles_per_chunk = p.chunk_len / QT;
  const int j0 = k_off_wg / QT;                          // first tile of the own chunk whose rows reach the workgroup's keys
  const char* qbase = (const char*)(p.q + (int64_t)kvh * p.q_gs);
  const char* dobase = (const char*)p.d_o;
  auto enter = [&](QTileIt& t) __attribute__((always_inline)) {     // position on the first tile of (hq, c ..), or hq == G
    while (t.hq < G) {
      while (t.c < p.n_q_chunks) {
        const int gq = p.q_gid[t.c];
        if (gq >= gk) {
          t.diag = gq == gk;
          t.j = t.diag ? j0 : 0;
          t.jend = PACKED ? seg_jend : q_tiles_per_chunk;
          const int64_t row = (int64_t)t.c * p.chunk_len + (int64_t)t.j * QT;
          const int head = kvh * G + t.hq;
          t.qp = qbase + ((int64_t)t.hq * p.q_hs + row * p.q_rs) * 2;
          t.dop = dobase + ((int64_t)head * p.do_hs + row * p.do_rs) * 2;
          t.lp = (const char*)(p.lse + (int64_t)head * p.n_q_rows + row);
          t.dlp = (const char*)(p.delta + (int64_t)head * p.n_q_rows + row);
          return;
        }
        ++t.c;
      }
      ++t.hq; t.c = 0;
    }
  };
  const int q_tile_bytes = (int)(p.q_rs * 2 * QT), do_tile_bytes = (int)(p.do_rs * 2 * QT);
  auto advance = [&](QTileIt& t) __attribute__((always_inline)) {
    t.qp += q_tile_bytes; t.dop += do_tile_bytes; t.lp += QT * 4; t.dlp += QT * 4;
    if (++t.j == t.jend) { ++t.c; enter(t); }
  };
  int n_tiles = 0;                                       // >= 4 G: the own chunk contributes at least four tiles
  for (int c = 0; c < p.n_q_chunks; ++c) {
    const int gq = p.q_gid[c];
    n_tiles += gq > gk ? q_tiles_per_chunk : (gq == gk ? (PACKED ? seg_jend : q_tiles_per_chunk) - j0 : 0);
  }
  n_tiles *= G;
  if (n_tiles == 0) {                                    // context parallelism: a key chunk none of the local queries can see
    const u32x2 z = {0u, 0u};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int64_t orow = k_row0 + 32 * kb + l31;
      bf16_t* op = IS_DK ? p.dk + orow * p.dk_rs + (int64_t)kvh * p.dk_hs : p.dv + orow * p.dv_rs + (int64_t)kvh * p.dv_hs;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x2*>(op + 8 * i + 4 * hi) = z;
    }
    return;
  }

  // ---- state -------------------------------------------------------------------------------------------------------------------------
  f32x16 o[2][4];                                        // dK^T or dV^T [kb][db] (AGPRs)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[kb][db][r] = 0.f;
      asm volatile("" : "+a"(o[kb][db]));
    }
  f32x16 sb[2][2], dpb[2][2];                            // S / dP of a half tile [parity][kb]: row 32 qh + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][2][4];                               // packed dS (IS_DK) or P [parity][kb][k-step t'][4 dwords]
  float lrow[16], drow[16];                              // lse * log2e / delta * scale of the 16 rows of a half this lane sees

  auto load_lse = [&](unsigned st, int qh) __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 l4 = *(lds_f32x4*)(uintptr_t)(st + (32 * qh + 8 * rg + 4 * hi) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) lrow[rg * 4 + j] = l4[j] * LOG2E;
    }
  };
  auto load_delta = [&](unsigned st, int qh) __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 d4 = *(lds_f32x4*)(uintptr_t)(st + 256 + (32 * qh + 8 * rg + 4 * hi) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) drow[rg * 4 + j] = d4[j] * scale;
    }
  };
  // P of pair e (kb = e >> 3, registers 2 (e & 7), +1) of buffer `par`, in place
  auto part1_pair = [&](int par, int e) __attribute__((always_inline)) {
    const int kb = e >> 3, r = 2 * (e & 7);
    sb[par][kb][r] = __builtin_amdgcn_exp2f(fmaf(sb[par][kb][r], scale_log2e, -lrow[r]));
    sb[par][kb][r + 1] = __builtin_amdgcn_exp2f(fmaf(sb[par][kb][r + 1], scale_log2e, -lrow[r + 1]));
  };
  // dS (IS_DK) or P of pair e, packed
  auto part2_pair = [&](int par, int e) __attribute__((always_inline)) {
    const int kb = e >> 3, pr = e & 7, r = 2 * pr;
    float a = sb[par][kb][r], b = sb[par][kb][r + 1];
    if (IS_DK) {
      a *= fmaf(dpb[par][kb][r], scale, -drow[r]);
      b *= fmaf(dpb[par][kb][r + 1], scale, -drow[r + 1]);
    }
    pk[par][kb][pr >> 2][pr & 3] = pack_bf16x2(a, b);
    asm volatile("" :: "v"(pk[par][kb][pr >> 2][pr & 3]));                      // computed HERE (no sinking past the phase)
  };
  auto frag = [&](unsigned slot_addr, int ds, int qh) __attribute__((always_inline)) {
    return *(lds_bf16x8*)(uintptr_t)(slot_addr + foff[ds] + qh * 32 * ROWB);
  };
  auto tr_frag = [&](unsigned slot_addr, int t4, int db) __attribute__((always_inline)) {      // rows 16 t4 .. + 15, d block db
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(slot_addr + toff[db] + 16 * t4 * ROWB));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(slot_addr + toff8[db] + 16 * t4 * ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 ac = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, ac);
  };
  // arithmetic causal mask of a half (q_off_h = first row of the half inside the chunk) in buffer `par`: key <= row is visible
  auto mask_half = [&](int par, int q_off_h) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int base = q_off_h + 4 * hi - (k_off + 32 * kb + l31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = (r & 3) + 8 * (r >> 2);
        const float pen = fminf((float)(base + rc), 0.0f);
        sb[par][kb][r] = fmaf(pen, 3.0e38f, sb[par][kb][r]);
      }
    }
  };
  auto seg_mask_half = [&](int par, int q_off_h) __attribute__((always_inline)) {       // packed samples: row < seg_hi[key] is visible
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int base = seg_hi[kb] - 1 - q_off_h - 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = (r & 3) + 8 * (r >> 2);
        const float pen = fminf((float)(base - rc), 0.0f);
        sb[par][kb][r] = fmaf(pen, 3.0e38f, sb[par][kb][r]);
      }
    }
  };
  // 16 slots: S of the next half (rows qh_n of the tile at fa) into buffer par ^ 1; FILL: dS / pack of half `par` behind them
  auto s_group = [&](int par, unsigned fa, int qh_n, bool fill) __attribute__((always_inline)) {
    bf16x8 fr[4];
    fr[0] = frag(fa, 0, qh_n); fr[1] = frag(fa, 1, qh_n);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, kb = s & 1;
      if (kb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(fa, ds + 2, qh_n);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], kf[kb][ds], z, 0, 0, 0);
      } else {
        sb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], kf[kb][ds], sb[par ^ 1][kb], 0, 0, 0);
      }
      if (fill) part2_pair(par, s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 16 slots (IS_DK): dP of the next half into buffer par ^ 1; FILL: the first 8 pairs of P(par ^ 1)
  auto p_group = [&](int par, unsigned fb, int qh_n, bool fill) __attribute__((always_inline)) {
    bf16x8 fr[4];
    fr[0] = frag(fb, 0, qh_n); fr[1] = frag(fb, 1, qh_n);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, kb = s & 1;
      if (kb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(fb, ds + 2, qh_n);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        dpb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], vf[kb][ds], z, 0, 0, 0);
      } else {
        dpb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], vf[kb][ds], dpb[par ^ 1][kb], 0, 0, 0);
      }
      if (fill && (s & 1) == 0) part1_pair(par ^ 1, s >> 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 16 slots: gradient^T += X^T(half qh of the tile at tr_) packed(par); FILL: pairs n0 .. of P(par ^ 1), `per` per slot pair
  auto g_group = [&](int par, unsigned tr_, int qh, bool fill) __attribute__((always_inline)) {
    bf16x8 tr[4];
    tr[0] = tr_frag(tr_, 2 * qh, 0); tr[1] = tr_frag(tr_, 2 * qh, 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int i = s >> 1, kb = s & 1, t2 = i >> 2, db = i & 3;
      if (kb == 0 && i + 2 < 8) tr[(i + 2) & 3] = tr_frag(tr_, 2 * qh + ((i + 2) >> 2), (i + 2) & 3);
      const u32x4 pw = {pk[par][kb][t2][0], pk[par][kb][t2][1], pk[par][kb][t2][2], pk[par][kb][t2][3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[kb][db]) : "v"(tr[i & 3]), "v"(pf));
      if (fill) {
        if (IS_DK) { if ((s & 1) == 0) part1_pair(par ^ 1, 8 + (s >> 1)); }      // pairs 0..7 went behind the dP MFMAs
        else part1_pair(par ^ 1, s);                                             // no dP group: all 16 pairs here
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto needs_mask = [&](const QTileIt& t) __attribute__((always_inline)) { return t.diag && t.j * QT < k_off_wg + KTILE; };
  // S (and dP) of the next half (qh_n of the tile in ring slot `slot3` whose iterator is `tn`) into buffers par ^ 1, filled with the
  // arithmetic of half `par` (whose statistics sit in ring slot `slot_cur`, half qh_cur)
  auto sp_group = [&](int par, int slot3, int qh_n, const QTileIt& tn, bool fill, int slot_cur, int qh_cur) __attribute__((always_inline)) {
    if (fill && IS_DK) load_delta(lds0 + LDS_ST + slot_cur * 512, qh_cur);
    s_group(par, lds0 + LDS_FA + slot3 * TILEB, qh_n, fill);
    if (needs_mask(tn)) mask_half(par ^ 1, tn.j * QT + 32 * qh_n);       // wave-uniform, diagonal tiles only
    if constexpr (PACKED) {
      if (tn.j * QT + 32 * qh_n + 31 >= seg_hi_min) seg_mask_half(par ^ 1, tn.j * QT + 32 * qh_n);
    }
    load_lse(lds0 + LDS_ST + slot3 * 512, qh_n);
    if (IS_DK) p_group(par, lds0 + LDS_FB + slot3 * TILEB, qh_n, fill);
  };

  // ---- prologue: tile 0 (all images) and tile 1 (fragment images) -> LDS; S / dP / P of half 0 -------------------------------------------
  QTileIt cur;
  cur.hq = 0; cur.c = 0; cur.j = 0; cur.jend = 0; cur.diag = 0; cur.qp = qbase; cur.dop = dobase; cur.lp = (const char*)p.lse;
  cur.dlp = (const char*)p.delta;
  enter(cur);
  QTileIt nx1 = cur;
  advance(nx1);                                          // n_tiles >= 4
  QTileIt nx2 = nx1;
  advance(nx2);
  dma_frag(cur, 0); dma_tr(cur, 0); dma_frag(nx1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  sp_group(1, 0, 0, cur, false, 0, 0);                   // -> buffers 0
#pragma unroll
  for (int e = 0; e < 16; ++e) part1_pair(0, e);

  // ---- main loop: one tile = two trips (half qh = buffer parity qh) ------------------------------------------------------------------------
  int s3 = 0, s3n = 1, s3nn = 2;                         // ring slots of tiles t, t+1, t+2 (fragment images, statistics)
  int tpar = 0;                                          // t & 1: the transposed-image ring slot of tile t
  auto iteration = [&](const bool has1, const bool has2) __attribute__((always_inline)) {
    if (has2) dma_frag(nx2, s3nn);                       // that slot held tile t-1 (last read before the previous barrier)
    if (has1) dma_tr(nx1, tpar ^ 1);
    const unsigned trs = IS_DK ? lds0 + LDS_FA + s3 * TILEB : lds0 + LDS_TR + tpar * TILEB;        // IS_DK: Q^T out of tile t's Q image
    // trip A: u = 2 t (buffers 0): next half = (tile t, qh 1)
    sp_group(0, s3, 1, cur, true, s3, 0);
    g_group(0, trs, 0, true);
    // trip B: u = 2 t + 1 (buffers 1): next half = (tile t+1, qh 0)
    if (has1) {
      sp_group(1, s3n, 0, nx1, true, s3, 1);
      g_group(1, trs, 1, true);
    } else {
      if (IS_DK) load_delta(lds0 + LDS_ST + s3 * 512, 1);
#pragma unroll
      for (int e = 0; e < 16; ++e) part2_pair(1, e);
      // VALU result -> inline-asm MFMA operand: wait states the compiler does not know are needed, tied to the operands
      asm volatile("s_nop 4" : "+v"(pk[1][0][0][0]), "+v"(pk[1][0][0][1]), "+v"(pk[1][0][0][2]), "+v"(pk[1][0][0][3]),
                   "+v"(pk[1][1][0][0]), "+v"(pk[1][1][0][1]), "+v"(pk[1][1][0][2]), "+v"(pk[1][1][0][3]));
      g_group(1, trs, 1, false);
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

  // ---- epilogue: dK / dV [key][d] -----------------------------------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]),
               "+a"(o[1][2]), "+a"(o[1][3]));
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int64_t orow = k_row0 + 32 * kb + l31;
    bf16_t* op = IS_DK ? p.dk + orow * p.dk_rs + (int64_t)kvh * p.dk_hs : p.dv + orow * p.dv_rs + (int64_t)kvh * p.dv_hs;
#pragma unroll
    for (int db = 0; db < 4; ++db) store_row_block32(op + 32 * db, o[kb][db], 1.0f, hi);      // two 16-byte stores per block (r06)
  }
}

template <bool IS_DK, bool PACKED>
int launch_kv64(const BwdArgs& a, hipStream_t st) {
  constexpr int lds = (IS_DK ? 6 : 5) * TILEB + 3 * 512;
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv64_kernel<IS_DK, PACKED>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  const int64_t n = (int64_t)a.n_kv_heads * a.n_kv_chunks * (a.chunk_len / KTILE);
  if (n > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((attn_bwd_kv64_kernel<IS_DK, PACKED>), dim3((unsigned)n), dim3(256), lds, st, a);
  return vita_check_launch();
}

}  // namespace

bool vita_attn_bwd_kv64_eligible(const BwdArgs& a) {
  if (a.head_dim != 128) return false;                       // the 64-rows-per-wave kernels are built for d = 128
  if (a.chunk_len % KTILE) return false;                    // (a key chunk sees whole chunks, its own from the diagonal on, or nothing)
  if (((uintptr_t)a.dk & 15) || ((uintptr_t)a.dv & 15) || (a.dk_rs & 7) || (a.dk_hs & 7) || (a.dv_rs & 7) || (a.dv_hs & 7)) return false;   // 16-byte stores (r06)
  if (a.seg_start && (a.n_q_chunks != 1 || a.n_kv_chunks != 1)) return false;      // packed samples: one chunk
  if ((int64_t)QT * a.q_rs * 2 > 0x7fffffffLL || (int64_t)QT * a.do_rs * 2 > 0x7fffffffLL) return false;
  const char* e = vita_dev_getenv("VITA_ATTN_BWD64");
  return !(e && e[0] == '0');
}

int vita_attn_bwd_kv64_launch(const BwdArgs& a, hipStream_t st) {
  if (a.seg_start) {
    const int rc = launch_kv64<true, true>(a, st);
    return rc != VITA_OK ? rc : launch_kv64<false, true>(a, st);
  }
  const int rc = launch_kv64<true, false>(a, st);
  return rc != VITA_OK ? rc : launch_kv64<false, false>(a, st);
}
