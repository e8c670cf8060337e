// Flash attention backward, dK AND dV in one launch, d = 128, causal, whole tiles (gfx950 / MI355X) — r04.
//
// attn_bwd_kv64.hip computes dK and dV in two launches because at 64 keys per wave the two gradient accumulators (256 registers) do
// not fit next to the wave's own K / V fragments: 5 GEMM units (S, dP, dK | S, dV) for 4 of algorithmic work.  Here the two
// accumulators live in two DIFFERENT waves.  A workgroup owns 128 keys = two wave PAIRS of 64 keys; inside a pair
//     wave A:  S = Q K^T,  P = exp2(S c - lse)  ->  bf16 P  ->  LDS,   dV^T += dO^T P                                   (2 units)
//     wave B:  dP = dO V^T,  P <- LDS,  dS = P o (dP - delta) scale,   dK^T += Q^T dS                                   (2 units)
// so S is computed once: 4 units, no duplicate, 32 MFMAs per wave and 32-row half either way.  What a wave keeps is what kv64's
// waves keep — 128 accumulator registers + its own K (A) or V (B) fragments in AGPRs, S / dP of two halves in VGPRs — and P crosses
// the pair through LDS in the accumulator's own lane layout (lane l of B reads exactly what lane l of A wrote: 4 KB per half, one
// ds_write_b128 / ds_read_b128 x 4 per lane, lanes consecutive: conflict-free), double-buffered, ONE workgroup barrier per half.
// P reaches dS rounded to bf16 — as in the reference's own chain (scale_mask_softmax hands bf16 probabilities to the backward,
// M/core/transformer/dot_product_attention.py:186-289); kv64 used the unrounded value there.
// LDS images: ONE dual-use image per Q tile and per dO tile (slot XOR swz(row), attn_bwd_kv64.hip's header): A reads Q as fragments
// and dO transposed, B reads dO as fragments and Q transposed — 32 KB of LDS-DMA per 64-row query tile, rings of three.
// Pipeline per half u (32 query rows):
//     B: [16 dP MFMAs of half u + 1 || dS(u), bf16 pack]  [16 dK MFMAs of half u]
//     A: [16 dV MFMAs of half u || first half of exp2(S(u + 1))]  [16 S MFMAs of half u + 2 || second half, bf16 pack]  hand-over of P(u + 1)
// A runs its S two halves ahead: the 32 exp2 of a half (the transcendental unit takes ~9 cycles each) then sit behind 32 MFMAs instead
// of 16 — with S only one half ahead A's second group was VALU-bound and B idled at the barrier (5.75 ms at 16K; this order: see DESIGN 5.1).
#include "attn_bwd_args.h"
#include <stdlib.h>

namespace {

constexpr int D = 128, QT = 64, KWG = 128, ROWB = D * 2, TILEB = QT * ROWB;             // 16 KiB per image of a 64-row query tile
constexpr int LDS_Q = 0, LDS_DO = 3 * TILEB, LDS_ST = 6 * TILEB, LDS_X = LDS_ST + 3 * 512, LDS_TOTAL = LDS_X + 2 * 2 * 4096;
constexpr float LOG2E = 1.44269504088896340736f;

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) char lds_char;

struct QTileIt {
  int hq, c, j, jend;   // query head of the group, query chunk, tile inside the chunk, one past the last tile; hq == G: end
  int diag;             // chunk c is the key block's own chunk
  const char* qp;       // first Q / dO row of the tile, lse / delta of its first row
  const char* dop;
  const char* lp;
  const char* dlp;
};

template <bool ROLE_B>
__device__ __forceinline__ void kvp_body(const BwdArgs& p, const unsigned lds0, const int wave, const int lane) {
  const int hi = lane >> 5, l31 = lane & 31;
  const int pair = wave >> 1;
  const int G = p.n_q_heads / p.n_kv_heads;
  const int kt_per_chunk = p.chunk_len / KWG;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int kc = bid / kt_per_chunk;                    // kv chunk (buffer order)
  const int kti = bid % kt_per_chunk;
  const int gk = p.kv_gid[kc];
  const int k_off_wg = kti * KWG;                       // first key of the workgroup inside its chunk
  const int k_off = k_off_wg + pair * 64;               // this PAIR's first key inside the chunk
  const int64_t k_row0 = p.kv_row[kc] + k_off;          // its row in the K / V buffers
  const float scale_log2e = p.scale_log2e, scale = p.scale;

  // ---- the pair's own keys: A keeps K, B keeps V, as MFMA B operands (key k_off + 32 kb + l31, d = 16 ds + 8 hi .. + 7) ------------------
  bf16x8 wf[2][8];
  {
    const bf16_t* wp = ROLE_B ? p.v + (int64_t)kvh * p.v_hs + (k_row0 + l31) * p.v_rs + hi * 8
                              : p.k + (int64_t)kvh * p.k_hs + (k_row0 + l31) * p.k_rs + hi * 8;
    const int64_t rs = ROLE_B ? p.v_rs : p.k_rs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) wf[kb][ds] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)32 * kb * rs + ds * 16);
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) asm volatile("" : "+a"(wf[kb][ds]));        // consumed (loads waited for) and pinned in AGPRs here

  // ---- LDS offsets of the dual-use images ----------------------------------------------------------------------------------------------
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  unsigned foff[8], toff[4], toff8[4];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) foff[ds] = l31 * ROWB + (((2 * ds + hi) ^ swz(l31 & 15)) << 4);       // + 32 qh rows: immediate
  {
    const int g16 = lane >> 4, i16 = lane & 15, row_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      toff[db] = row_l * ROWB + (((col >> 3) ^ swz(row_l)) << 4) + (col & 7) * 2;
      toff8[db] = (row_l + 8) * ROWB + (((col >> 3) ^ swz(row_l + 8)) << 4) + (col & 7) * 2;
    }
  }
  // ---- LDS-DMA: wave w moves pieces 4w .. 4w+3 (1 KiB = 4 rows) of both images; the swizzle goes on the SOURCE address -----------------
  unsigned off_q[4], off_do[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 4 + (lane >> 4), ps = lane & 15;
    const int fs = ps ^ swz(row & 15);
    off_q[q] = (unsigned)((row * p.q_rs + fs * 8) * 2);
    off_do[q] = (unsigned)((row * p.do_rs + fs * 8) * 2);
  }
  const unsigned lds_w = lds0 + wave * 4096;
  auto dma_tile = [&](const QTileIt& t, int slot3) __attribute__((always_inline)) {
    const vita_rsrc_t rq = vita_make_rsrc_uniform(t.qp), rd = vita_make_rsrc_uniform(t.dop);
    unsigned base = lds_w + slot3 * TILEB;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vita_lds_dma16(rq, off_q[q], base + LDS_Q + q * 1024);
      vita_lds_dma16(rd, off_do[q], base + LDS_DO + q * 1024);
    }
    // 64 lse (wave 0) / 64 delta (wave 1) of the tile's rows: lane -> row
    if (wave == 0) vita_lds_dma4(vita_make_rsrc_uniform(t.lp), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512);
    if (wave == 1) vita_lds_dma4(vita_make_rsrc_uniform(t.dlp), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512 + 256);
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
          t.jend = q_tiles_per_chunk;
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
  int n_tiles = 0;                                       // 0, or >= 2 G: the own chunk contributes at least two tiles
  for (int c = 0; c < p.n_q_chunks; ++c) {
    const int gq = p.q_gid[c];
    n_tiles += gq > gk ? q_tiles_per_chunk : (gq == gk ? q_tiles_per_chunk - j0 : 0);
  }
  n_tiles *= G;
  bf16_t* const out = ROLE_B ? p.dk + (int64_t)kvh * p.dk_hs : p.dv + (int64_t)kvh * p.dv_hs;
  const int64_t out_rs = ROLE_B ? p.dk_rs : p.dv_rs;
  if (n_tiles == 0) {                                    // context parallelism: a key chunk none of the local queries can see
    const u32x2 z = {0u, 0u};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16_t* op = out + (k_row0 + 32 * kb + l31) * out_rs;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x2*>(op + 8 * i + 4 * hi) = z;
    }
    return;
  }

  // ---- state -------------------------------------------------------------------------------------------------------------------------
  f32x16 o[2][4];                                        // A: dV^T, B: dK^T  [kb][db] (AGPRs)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[kb][db][r] = 0.f;
      asm volatile("" : "+a"(o[kb][db]));
    }
  f32x16 xb[2][2];                                       // A: S, B: dP of a half [parity][kb]: row 32 qh + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][2][4];                               // A: packed P, B: packed dS  [parity][kb][k-step t'][4 dwords]
  u32x4 pin[2][2];                                       // B: the pair's packed P of the current half, as A wrote it  [kb][t']
  float rstat[16];                                       // A: lse * log2e, B: delta * scale of the 16 rows of a half this lane sees
  const unsigned xaddr = lds0 + LDS_X + pair * 4096 + lane * 16;     // + parity * 8192 + (kb * 2 + t') * 1024

  auto load_stat = [&](unsigned st, int qh) __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 v4 = *(lds_f32x4*)(uintptr_t)(st + (ROLE_B ? 256 : 0) + (32 * qh + 8 * rg + 4 * hi) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) rstat[rg * 4 + j] = v4[j] * (ROLE_B ? scale : LOG2E);
    }
  };
  // A: P of pair e (kb = e >> 3, registers 2 (e & 7), +1) of buffer `par`, in place
  auto exp_pair = [&](int par, int e) __attribute__((always_inline)) {
    const int kb = e >> 3, r = 2 * (e & 7);
    xb[par][kb][r] = __builtin_amdgcn_exp2f(fmaf(xb[par][kb][r], scale_log2e, -rstat[r]));
    xb[par][kb][r + 1] = __builtin_amdgcn_exp2f(fmaf(xb[par][kb][r + 1], scale_log2e, -rstat[r + 1]));
  };
  auto pack_pair = [&](int par, int e) __attribute__((always_inline)) {          // A: bf16 pair of P
    const int kb = e >> 3, pr = e & 7, r = 2 * pr;
    pk[par][kb][pr >> 2][pr & 3] = pack_bf16x2(xb[par][kb][r], xb[par][kb][r + 1]);
    asm volatile("" :: "v"(pk[par][kb][pr >> 2][pr & 3]));                       // computed HERE (no sinking past the phase)
  };
  auto ds_pair = [&](int par, int e) __attribute__((always_inline)) {            // B: dS = P o (dP scale - delta scale), packed
    const int kb = e >> 3, pr = e & 7, r = 2 * pr;
    const unsigned w = pin[kb][pr >> 2][pr & 3];
    const float a = bf16lo_to_f32(w) * fmaf(xb[par][kb][r], scale, -rstat[r]);
    const float b = bf16hi_to_f32(w) * fmaf(xb[par][kb][r + 1], scale, -rstat[r + 1]);
    pk[par][kb][pr >> 2][pr & 3] = pack_bf16x2(a, b);
    asm volatile("" :: "v"(pk[par][kb][pr >> 2][pr & 3]));
  };
  auto hand_over = [&](int par) __attribute__((always_inline)) {                 // A: packed P of buffer `par` -> the pair's LDS slot
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const u32x4 w = {pk[par][kb][t2][0], pk[par][kb][t2][1], pk[par][kb][t2][2], pk[par][kb][t2][3]};
        *(lds_u32x4*)(uintptr_t)(xaddr + par * 8192 + (kb * 2 + t2) * 1024) = w;
      }
  };
  auto take_over = [&](int par) __attribute__((always_inline)) {                 // B: the pair's packed P of buffer `par`
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) pin[kb][t2] = *(lds_u32x4*)(uintptr_t)(xaddr + par * 8192 + (kb * 2 + t2) * 1024);
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
  // A: arithmetic causal mask of a half (q_off_h = first row of the half inside the chunk) in buffer `par`: key <= row is visible
  auto mask_half = [&](int par, int q_off_h) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int base = q_off_h + 4 * hi - (k_off + 32 * kb + l31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = (r & 3) + 8 * (r >> 2);
        const float pen = fminf((float)(base + rc), 0.0f);
        xb[par][kb][r] = fmaf(pen, 3.0e38f, xb[par][kb][r]);
      }
    }
  };
  // 16 slots: X(rows qh_n of the image at `img`) = rows x own fragments -> buffer par ^ 1   (A: S = Q K^T, B: dP = dO V^T);
  // FILL works on buffer `par`: B — dS / pack of its 16 pairs; A — exp2 of pairs 8 .. 15 (even slots) and their bf16 pack (odd slots)
  auto x_group = [&](int par, unsigned img, int qh_n, bool fill) __attribute__((always_inline)) {
    bf16x8 fr[4];
    fr[0] = frag(img, 0, qh_n); fr[1] = frag(img, 1, qh_n);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, kb = s & 1;
      if (kb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(img, ds + 2, qh_n);
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        xb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], wf[kb][ds], z, 0, 0, 0);
      } else {
        xb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], wf[kb][ds], xb[par ^ 1][kb], 0, 0, 0);
      }
      if (ROLE_B && fill) ds_pair(par, s);
      if (!ROLE_B && fill) {
        if ((s & 1) == 0) exp_pair(par, 8 + (s >> 1));
        else pack_pair(par, 8 + (s >> 1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // 16 slots: gradient^T += X^T(half qh of the image at `img`, transposed reads) packed(par)   (A: dV^T += dO^T P, B: dK^T += Q^T dS);
  // A with FILL: exp2 of pairs 0 .. 7 of buffer par ^ 1 (even slots) and their bf16 pack (odd slots: an exp2 result is never consumed
  // by the next instruction) behind the MFMAs
  auto g_group = [&](int par, unsigned img, int qh, bool fill) __attribute__((always_inline)) {
    bf16x8 tr[4];
    tr[0] = tr_frag(img, 2 * qh, 0); tr[1] = tr_frag(img, 2 * qh, 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int i = s >> 1, kb = s & 1, t2 = i >> 2, db = i & 3;
      if (kb == 0 && i + 2 < 8) tr[(i + 2) & 3] = tr_frag(img, 2 * qh + ((i + 2) >> 2), (i + 2) & 3);
      const u32x4 pw = {pk[par][kb][t2][0], pk[par][kb][t2][1], pk[par][kb][t2][2], pk[par][kb][t2][3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[kb][db]) : "v"(tr[i & 3]), "v"(pf));
      if (!ROLE_B && fill) {
        if ((s & 1) == 0) exp_pair(par ^ 1, s >> 1);
        else pack_pair(par ^ 1, s >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto needs_mask = [&](const QTileIt& t) __attribute__((always_inline)) { return t.diag && t.j * QT < k_off_wg + KWG; };
  // VALU result -> inline-asm MFMA operand: wait states the compiler does not know are needed, tied to the operands
  auto settle = [&](int par) __attribute__((always_inline)) {
    asm volatile("s_nop 4" : "+v"(pk[par][0][0][0]), "+v"(pk[par][0][0][1]), "+v"(pk[par][0][0][2]), "+v"(pk[par][0][0][3]),
                 "+v"(pk[par][1][0][0]), "+v"(pk[par][1][0][1]), "+v"(pk[par][1][0][2]), "+v"(pk[par][1][0][3]));
  };
  auto pair_barrier = [&]() __attribute__((always_inline)) {       // LDS hand-over visible to the partner (whole workgroup: one barrier kind)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: tiles 0 and 1 -> LDS; X of half 0; A: P(0) packed and handed over ---------------------------------------------------
  QTileIt cur;
  cur.hq = 0; cur.c = 0; cur.j = 0; cur.jend = 0; cur.diag = 0; cur.qp = qbase; cur.dop = dobase; cur.lp = (const char*)p.lse;
  cur.dlp = (const char*)p.delta;
  enter(cur);
  QTileIt nx1 = cur;
  advance(nx1);                                          // n_tiles >= 2
  QTileIt nx2 = nx1;
  if (n_tiles > 2) advance(nx2);
  dma_tile(cur, 0); dma_tile(nx1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned IMG_X = ROLE_B ? LDS_DO : LDS_Q;        // the image this role reads as fragments (dO for dP, Q for S)
  const unsigned IMG_G = ROLE_B ? LDS_Q : LDS_DO;        // ... and transposed (Q^T for dK, dO^T for dV)
  x_group(1, lds0 + IMG_X, 0, false);                    // -> buffers 0
  if (!ROLE_B) {
    if (needs_mask(cur)) mask_half(0, cur.j * QT);
    load_stat(lds0 + LDS_ST, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) exp_pair(0, e);
#pragma unroll
    for (int e = 0; e < 16; ++e) pack_pair(0, e);
    hand_over(0);
    settle(0);
    x_group(0, lds0 + IMG_X, 1, false);                  // A runs two halves ahead: S(1) -> buffers 1 (tile 0 always has both halves)
    if (needs_mask(cur)) mask_half(1, cur.j * QT + 32);
  }
  pair_barrier();

  // ---- main loop: one tile = two trips (half qh = buffer parity qh) ------------------------------------------------------------------------
  int s3 = 0, s3n = 1, s3nn = 2;                         // ring slots of tiles t, t+1, t+2
  // one trip: half u = (tile in ring slot `sl`, half qh) in buffers `par`; half u + 1 = (tile `tn` in slot `sln`, half qh_n) if has_next;
  // half u + 2 = (tile nx1 in slot s3n, half qh_2) if has_next2 (it always lies in tile t + 1)
  auto trip = [&](int par, int sl, int qh, const QTileIt& tn, int sln, int qh_n, const bool has_next, int qh_2, const bool has_next2)
      __attribute__((always_inline)) {
    const unsigned img_g = lds0 + IMG_G + sl * TILEB;
    if (ROLE_B) {
      take_over(par);                                    // P(u), written by A before the last barrier
      load_stat(lds0 + LDS_ST + sl * 512, qh);           // delta of half u
      if (has_next) {
        x_group(par, lds0 + IMG_X + sln * TILEB, qh_n, true);               // dP(u + 1)  ||  dS(u)
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) ds_pair(par, e);
      }
      settle(par);
      g_group(par, img_g, qh, false);                    // dK^T += Q^T dS(u)
    } else {
      // A holds P(u) packed in pk[par] and the raw (masked) S(u + 1) in buffers par ^ 1
      if (has_next) load_stat(lds0 + LDS_ST + sln * 512, qh_n);             // lse of half u + 1
      g_group(par, img_g, qh, has_next);                 // dV^T += dO^T P(u)  ||  exp2 / pack of pairs 0 .. 7 of S(u + 1)
      if (has_next) {
        if (has_next2) {
          x_group(par ^ 1, lds0 + IMG_X + s3n * TILEB, qh_2, true);        // S(u + 2) -> buffers par  ||  pairs 8 .. 15 of S(u + 1)
        } else {
#pragma unroll
          for (int e = 8; e < 16; ++e) exp_pair(par ^ 1, e);
#pragma unroll
          for (int e = 8; e < 16; ++e) pack_pair(par ^ 1, e);
        }
        hand_over(par ^ 1);                              // P(u + 1) -> the partner, before the barrier that ends this trip
        if (has_next2 && needs_mask(nx1)) mask_half(par, nx1.j * QT + 32 * qh_2);       // wave-uniform, diagonal tiles only
      }
    }
  };
  auto iteration = [&](const bool has1, const bool has2) __attribute__((always_inline)) {
    if (has2) dma_tile(nx2, s3nn);                       // that slot held tile t-1 (last read before the previous tile barrier)
    trip(0, s3, 0, cur, s3, 1, true, 0, has1);           // u = 2 t:     u + 1 = (tile t, rows 32 ..),    u + 2 = (tile t + 1, rows 0 ..)
    pair_barrier();
    trip(1, s3, 1, nx1, s3n, 0, has1, 1, has1);          // u = 2 t + 1: u + 1 = (tile t + 1, rows 0 ..), u + 2 = (tile t + 1, rows 32 ..)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur = nx1; nx1 = nx2;
    if (has2) advance(nx2);
    const int tmp = s3; s3 = s3n; s3n = s3nn; s3nn = tmp;
  };
  for (int t = 0; t + 2 < n_tiles; ++t) iteration(true, true);
  iteration(true, false);
  iteration(false, false);

  // ---- epilogue: dK / dV [key][d] -----------------------------------------------------------------------------------------------------------
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]),
               "+a"(o[1][2]), "+a"(o[1][3]));
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    bf16_t* op = out + (k_row0 + 32 * kb + l31) * out_rs;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * db + 8 * rg + 4 * hi;
        const u32x2 w = {pack_bf16x2(o[kb][db][rg * 4 + 0], o[kb][db][rg * 4 + 1]),
                         pack_bf16x2(o[kb][db][rg * 4 + 2], o[kb][db][rg * 4 + 3])};
        *reinterpret_cast<u32x2*>(op + d) = w;
      }
  }
}

__global__ __launch_bounds__(256, 1) void attn_bwd_kvp_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every wave of the workgroup executes the same number of barriers: the two roles differ only in what runs between them
  if (wave & 1) kvp_body<true>(p, lds0, wave, lane);
  else kvp_body<false>(p, lds0, wave, lane);
}

}  // namespace

// whole 128-key tiles, one chunk geometry per launch; packed samples stay on attn_bwd_kv64.hip's two launches
bool vita_attn_bwd_kvp_eligible(const BwdArgs& a) {
  if (a.chunk_len % KWG || a.chunk_len % QT) return false;
  if (a.seg_start) return false;
  if ((int64_t)QT * a.q_rs * 2 > 0x7fffffffLL || (int64_t)QT * a.do_rs * 2 > 0x7fffffffLL) return false;
  const char* e = vita_dev_getenv("VITA_ATTN_BWD_KVP");            // developer A/B switch: 0 = the two kv64 launches
  return !(e && e[0] == '0');
}

int vita_attn_bwd_kvp_launch(const BwdArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kvp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  });
  const int64_t n = (int64_t)a.n_kv_heads * a.n_kv_chunks * (a.chunk_len / KWG);
  if (n > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(attn_bwd_kvp_kernel, dim3((unsigned)n), dim3(256), LDS_TOTAL, st, a);
  return vita_check_launch();
}
