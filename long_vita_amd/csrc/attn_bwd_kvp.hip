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
// and dO transposed, B reads dO as fragments and Q transposed — 32 KB of LDS-DMA per 64-row query tile, rings of four (r05; three in r04).
// Pipeline per half u (32 query rows), 16-MFMA groups, one MFMA + its share of the arithmetic per slot:
//     B: [16 dP MFMAs of half u + 1 || dS(u), k-step 0 (8 pairs, every other slot)]  [16 dK MFMAs of half u || dS(u), k-step 1 under the first 8]
//     A: [16 dV MFMAs of half u || exp2 of elements 0 .. 15 of S(u + 1)]  [16 S MFMAs of half u + 2 || elements 16 .. 31]  hand-over of P(u + 1)
// A runs its S two halves ahead: the 32 exp2 of a half (the transcendental unit takes ~16 issue cycles each) then sit behind 32 MFMAs, ONE
// per slot — with S only one half ahead A's second group was VALU-bound and B idled at the barrier (5.75 ms at 16K), with two exp2 in
// every other slot 5.5 ms, one per slot 5.3-5.4 ms.  Every group starts from fragments its predecessor read at slot 12 and the lse / delta of
// a half are fetched a trip ahead, so no group waits for an LDS round trip before its first MFMA (that alone: +- 0: same-box A/B).
// Timing-only ablations (KVP_ABL, same box, whole backward 8.84 ms): no barriers 8.90, no exp2 8.38, no softmax / dS arithmetic at all 7.9 —
// the barriers are free, and what the arithmetic costs it costs in power, not in issue slots: 1.25 PFLOP/s executed without it is the
// chip's ceiling for an LDS-fed MFMA stream on random data (MI355X_MICROARCH.md, DVFS; the forward runs 1.20-1.24).
//
// r05 — the schedule across workgroups (what the XCD's L2 sees).  Every workgroup streams the Q / dO tiles its keys are visible to; in r04
// each of the 32 workgroups resident on an XCD walked its own (head, tile) sequence from its own first tile, the sequences drifted apart by
// two tiles per workgroup and per head, and the XCD's 4 MB L2 served half of the 21 GB of tile reads (10.2 GB fetched per 16K launch against
// 0.4 GB of distinct data: rocprofv3 FETCH_SIZE, profiles/r04_attn_bwd16k_pmc.txt).  Now:
//   * block b = one kv head (b % n_kv_heads: with 8 kv heads one head per XCD) and one PAIR of 128-key blocks of it, (w, NB - 1 - w) in
//     global key order: under the causal mask the pair sees NB + 1 tile columns whatever w is, so all workgroups are equally long
//     (16K: 512 workgroups = exactly two rounds of 256 CUs, no triangular tail);
//   * inside a block the tiles go tile-major with the group's query heads innermost, block w from the LAST query tile DOWN to its
//     diagonal, block NB - 1 - w from its diagonal UP to the last tile: at global step s every first-phase workgroup of the XCD reads
//     tile (last - s / G, head s % G) and every second-phase workgroup tile (s / G - 1, head s % G) — two streams per XCD instead of 32.
// Measured (16K, same box, dK + dV pass alone, best / median of 5): r04 5.20 / 5.45 ms -> 5.13 / 5.29 ms with 0.89 GB fetched per launch
// (11 x less); the ceiling of ANY locality scheme — timing ablation KVP_ABL = 8, every tile index taken mod 64 so that all reads hit the
// L2 — is 5.00 / 5.10 ms.  The 25 x re-reads of r04 cost 5 % of the pass, not more: the pass is bound by issue and power
// (MI355X_MICROARCH.md, DVFS: the fifth back-to-back launch of the same kernel runs 11 % slower than the first), as r04 concluded.
#include "attn_bwd_args.h"
#include <stdlib.h>

namespace {

constexpr int D = 128, QT = 64, KWG = 128, ROWB = D * 2, TILEB = QT * ROWB;             // 16 KiB per image of a 64-row query tile
constexpr int NRING = 4;                                // Q / dO tile images in LDS: tiles t, t + 1 (read), t + 2 (landed or landing), t + 3 (just issued)
constexpr int LDS_Q = 0, LDS_DO = NRING * TILEB, LDS_ST = 2 * NRING * TILEB, LDS_X = LDS_ST + NRING * 512, LDS_TOTAL = LDS_X + 2 * 2 * 4096;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr int PF_NONE = 0, PF_FRAG = 1, PF_TR = 2;
// timing-only ablations (WRONG results; never in libvita_hip.so): -DKVP_ABL=1 no barrier between the two trips of a tile, 2 no exp2,
// 3 no barrier at all, 5 no softmax / dS arithmetic at all (MFMAs + LDS traffic only), 6 = 5 + 3, 7 every tile read from the SAME
// addresses (tile 0 of head 0: every LDS-DMA after the first hits the XCD's L2 — the upper bound of what any re-ordering of the
// workgroups for L2 locality could give — but also the same operand bits in every MFMA: less switching power, higher clock), 8 every tile
// read from a 64-tile window of head 0 (tile index mod 64: 2 MB of DISTINCT random data that stays in the XCD's 4 MB L2)
#ifndef KVP_ABL
#define KVP_ABL 0
#endif
constexpr int ABL = KVP_ABL;
#ifndef KVP_HAND_SPREAD
#define KVP_HAND_SPREAD 1          // r06: three of the four hand-over stores of P behind MFMAs of the S group (0: all four in front of the pair barrier)
#endif
#ifndef KVP_DMA_SPREAD
#define KVP_DMA_SPREAD 1
#endif
constexpr bool DMA_SPREAD = KVP_DMA_SPREAD != 0;       // (0: the nine pieces of a tile in a burst at the top of the iteration, for A / B builds)

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) char lds_char;

struct QTileIt {
  int hq, c, j, jlo;    // query head of the group, query chunk, tile inside the chunk, first tile of the chunk the key block reaches
  int diag;             // chunk c is the key block's own chunk
  int stat;             // index of the tile's first row in lse / delta: head * n_q_rows + row
  const char* qp;       // first Q / dO row of the tile
  const char* dop;
};

template <bool ROLE_B>
__device__ __forceinline__ void kvp_body(const BwdArgs& p, const unsigned lds0, const int wave, const int lane) {
  const int hi = lane >> 5, l31 = lane & 31;
  const int pair = wave >> 1;
  const int G = p.n_q_heads / p.n_kv_heads;
  const int kt_per_chunk = p.chunk_len / KWG;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;   // bid: the pair (bid, NB - 1 - bid) of key blocks in global key order
  const int NB = p.n_kv_chunks * kt_per_chunk;
  const float scale_log2e = p.scale_log2e, scale = p.scale;

  // ---- per key block (set by the phase loop below) ---------------------------------------------------------------------------------------
  int gk = 0, k_off_wg = 0, k_off = 0, j0 = 0, dir = 1;  // gid of the block's chunk, first key of the block / of this PAIR inside the chunk,
  int64_t k_row0 = 0;                                    // first own-chunk tile that reaches the block, walking direction; row in K / V
  bf16x8 wf[2][8];                                       // the pair's own keys: A keeps K, B keeps V, as MFMA B operands (key k_off + 32 kb + l31,
                                                         // d = 16 ds + 8 hi .. + 7), pinned in AGPRs
  auto load_own_keys = [&]() __attribute__((always_inline)) {
    const bf16_t* wp = ROLE_B ? p.v + (int64_t)kvh * p.v_hs + (k_row0 + l31) * p.v_rs + hi * 8
                              : p.k + (int64_t)kvh * p.k_hs + (k_row0 + l31) * p.k_rs + hi * 8;
    const int64_t rs = ROLE_B ? p.v_rs : p.k_rs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) wf[kb][ds] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)32 * kb * rs + ds * 16);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) asm volatile("" : "+a"(wf[kb][ds]));      // consumed (loads waited for) and pinned in AGPRs here
  };

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
  const char* qbase = (const char*)(p.q + (int64_t)kvh * p.q_gs);
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
    if (wave == 0) vita_lds_dma4(vita_make_rsrc_uniform(p.lse + t.stat), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512);
    if (wave == 1) vita_lds_dma4(vita_make_rsrc_uniform(p.delta + t.stat), (unsigned)(lane * 4), lds0 + LDS_ST + slot3 * 512 + 256);
  };

  // the same tile one piece at a time (r05, KVP_DMA_SPREAD): piece i = 0 .. 3 Q line i, 4 .. 7 dO line i - 4, 8 the lse / delta row of waves
  // 0 / 1 — issued one per MFMA slot behind the first MFMAs of a trip instead of nine in a burst behind the barrier (attn64.hip r05)
  QTileIt it;                                            // the pipeline's cursor (described with the iteration below)
  int dma_slot3 = 0;
  auto dma_tile_piece = [&](const QTileIt& t, int i) __attribute__((always_inline)) {
    unsigned base = lds_w + dma_slot3 * TILEB;
    asm volatile("" : "+s"(base));
    if (i < 4) vita_lds_dma16(vita_make_rsrc_uniform(t.qp), off_q[i], base + LDS_Q + i * 1024);
    else if (i < 8) vita_lds_dma16(vita_make_rsrc_uniform(t.dop), off_do[i - 4], base + LDS_DO + (i - 4) * 1024);
    else {
      if (wave == 0) vita_lds_dma4(vita_make_rsrc_uniform(p.lse + t.stat), (unsigned)(lane * 4), lds0 + LDS_ST + dma_slot3 * 512);
      if (wave == 1) vita_lds_dma4(vita_make_rsrc_uniform(p.delta + t.stat), (unsigned)(lane * 4), lds0 + LDS_ST + dma_slot3 * 512 + 256);
    }
  };

  // ---- iteration space: (query chunks that see the key block) x (64-row tiles) x (query heads of the group, innermost) ------------------
  // walked upwards (dir = 1: from the block's diagonal to the last tile) or downwards (dir = -1: the reverse sequence)
  const int q_tiles_per_chunk = p.chunk_len / QT;
  // steps of the walk, signed by its direction (set per key block): one query head on, and one tile on while the head wraps around
  int64_t q_hstep = 0, do_hstep = 0, q_tstep = 0, do_tstep = 0;
  int st_hstep = 0, st_tstep = 0;
  auto setptr = [&](QTileIt& t) __attribute__((always_inline)) {    // chunk entry only: everything else is increments
    const int64_t row = ABL == 7 ? 0 : (ABL == 8 ? (int64_t)(t.j & 63) * QT : (int64_t)t.c * p.chunk_len + (int64_t)t.j * QT);
    const int hq = ABL == 7 || ABL == 8 ? 0 : t.hq;
    const int head = kvh * G + hq;
    t.qp = qbase + ((int64_t)hq * p.q_hs + row * p.q_rs) * 2;
    t.dop = (const char*)p.d_o + ((int64_t)head * p.do_hs + row * p.do_rs) * 2;
    t.stat = head * p.n_q_rows + (int)row;
  };
  auto enter = [&](QTileIt& t) __attribute__((always_inline)) {     // from chunk t.c on (in direction dir): the first tile of the next visible chunk
    while (t.c >= 0 && t.c < p.n_q_chunks) {
      const int gq = p.q_gid[t.c];
      if (gq >= gk) {
        t.diag = gq == gk;
        t.jlo = t.diag ? j0 : 0;
        t.j = dir > 0 ? t.jlo : q_tiles_per_chunk - 1;
        t.hq = dir > 0 ? 0 : G - 1;
        setptr(t);
        return;
      }
      t.c += dir;
    }
  };
  auto advance = [&](QTileIt& t) __attribute__((always_inline)) {
    t.hq += dir;
    if (t.hq >= 0 && t.hq < G) {                          // the next head of the same tile (G - 1 times out of G)
      if (ABL != 7 && ABL != 8) { t.qp += q_hstep; t.dop += do_hstep; t.stat += st_hstep; }
      return;
    }
    t.hq = dir > 0 ? 0 : G - 1;
    t.j += dir;
    if (t.j >= t.jlo && t.j < q_tiles_per_chunk) {
      if (ABL == 8) setptr(t);
      else if (ABL != 7) { t.qp += q_tstep; t.dop += do_tstep; t.stat += st_tstep; }
      return;
    }
    t.c += dir;
    enter(t);
  };
  // the causal mask of a tile: its first row inside the chunk when rows of the tile precede keys of the workgroup, else -1
  auto mask_row = [&](const QTileIt& t) __attribute__((always_inline)) { return t.diag && t.j * QT < k_off_wg + KWG ? t.j * QT : -1; };

  // ---- state -------------------------------------------------------------------------------------------------------------------------
  f32x16 o[2][4];                                        // A: dV^T, B: dK^T  [kb][db] (AGPRs); zeroed per key block
  f32x16 xb[2][2];                                       // A: S, B: dP of a half [parity][kb]: row 32 qh + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][2][4];                               // A: packed P, B: packed dS  [parity][kb][k-step t'][4 dwords]
  u32x4 pin[2][2];                                       // B: the pair's packed P of the current half, as A wrote it  [kb][t']
  float rstat[16];                                       // A: lse * log2e, B: delta * scale of the 16 rows of a half this lane sees
  f32x4 stat_raw[4];                                     // ... of the NEXT half, as read from LDS before the barrier (stat_finish scales it)
  bf16x8 fr_pre[2];                                      // the first two fragments of the NEXT 16-MFMA group, read under the current one
  const unsigned xaddr = lds0 + LDS_X + pair * 4096 + lane * 16;     // + parity * 8192 + (kb * 2 + t') * 1024

  auto stat_fetch = [&](unsigned st, int qh) __attribute__((always_inline)) {     // LDS reads only: issued a group ahead of their use
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) stat_raw[rg] = *(lds_f32x4*)(uintptr_t)(st + (ROLE_B ? 256 : 0) + (32 * qh + 8 * rg + 4 * hi) * 4);
  };
  auto stat_finish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < 4; ++j) rstat[rg * 4 + j] = stat_raw[rg][j] * (ROLE_B ? scale : LOG2E);
  };
  auto load_stat = [&](unsigned st, int qh) __attribute__((always_inline)) { stat_fetch(st, qh); stat_finish(); };
  // A: P of pair e (kb = e >> 3, registers 2 (e & 7), +1) of buffer `par`, in place
  auto exp_pair = [&](int par, int e) __attribute__((always_inline)) {
    if (ABL == 2 || ABL == 5 || ABL == 6) return;
    const int kb = e >> 3, r = 2 * (e & 7);
    xb[par][kb][r] = __builtin_amdgcn_exp2f(fmaf(xb[par][kb][r], scale_log2e, -rstat[r]));
    xb[par][kb][r + 1] = __builtin_amdgcn_exp2f(fmaf(xb[par][kb][r + 1], scale_log2e, -rstat[r + 1]));
  };
  // A: P of ONE element (el = 16 kb + r) — one transcendental per MFMA slot: two in one slot (r04 first form) took 40 cycles of issue
  // against the MFMA's 32 and left the next slot half empty (timing ablations: the softmax arithmetic cost 17 % of the kernel)
  auto exp_one = [&](int par, int el) __attribute__((always_inline)) {
    if (ABL == 2 || ABL == 5 || ABL == 6) return;
    const int kb = el >> 4, r = el & 15;
    xb[par][kb][r] = __builtin_amdgcn_exp2f(fmaf(xb[par][kb][r], scale_log2e, -rstat[r]));
  };
  auto pack_pair = [&](int par, int e) __attribute__((always_inline)) {          // A: bf16 pair of P
    const int kb = e >> 3, pr = e & 7, r = 2 * pr;
    if (ABL == 5 || ABL == 6) { pk[par][kb][pr >> 2][pr & 3] = __builtin_bit_cast(unsigned, xb[par][kb][r]); return; }
    pk[par][kb][pr >> 2][pr & 3] = pack_bf16x2(xb[par][kb][r], xb[par][kb][r + 1]);
    asm volatile("" :: "v"(pk[par][kb][pr >> 2][pr & 3]));                       // computed HERE (no sinking past the phase)
  };
  auto ds_pair = [&](int par, int e) __attribute__((always_inline)) {            // B: dS = P o (dP scale - delta scale), packed
    const int kb = e >> 3, pr = e & 7, r = 2 * pr;
    if (ABL == 5 || ABL == 6) { pk[par][kb][pr >> 2][pr & 3] = pin[kb][pr >> 2][pr & 3] ^ __builtin_bit_cast(unsigned, xb[par][kb][r]); return; }
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
  // r06 (KVP_HAND_SPREAD): one of the four 16-byte stores of a hand-over.  Issued back to back in front of the pair barrier, the four stores of
  // the two A waves fill the LDS data FIFO (SQ_LDS_DATA_FIFO_FULL 2.4e7 per 16K launch: a ds_write_b128 moves 5 source dwords at 2 cycles each)
  // and the wave stalls with the matrix pipe draining; three of them now go out behind MFMAs of the S group, as soon as their pairs are packed
  auto hand_over_part = [&](int par, int i) __attribute__((always_inline)) {
    const int kb = i >> 1, t2 = i & 1;
    const u32x4 w = {pk[par][kb][t2][0], pk[par][kb][t2][1], pk[par][kb][t2][2], pk[par][kb][t2][3]};
    *(lds_u32x4*)(uintptr_t)(xaddr + par * 8192 + (kb * 2 + t2) * 1024) = w;
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
  // PF_NONE | PF_FRAG (the next group is an x_group: rows pf_qh of the image at pf_img) | PF_TR (a g_group: half pf_qh of pf_img): the
  // group's own fragment reads end at slot 10; at slot 12 it reads the first two fragments of the NEXT group into fr_pre, so that the
  // group after a barrier (or after this one) starts its MFMAs at once instead of behind an LDS round trip.  use_pre: take them.
  auto prefetch = [&](int pf, unsigned pf_img, int pf_qh) __attribute__((always_inline)) {
    if (pf == PF_FRAG) { fr_pre[0] = frag(pf_img, 0, pf_qh); fr_pre[1] = frag(pf_img, 1, pf_qh); }
    if (pf == PF_TR) { fr_pre[0] = tr_frag(pf_img, 2 * pf_qh, 0); fr_pre[1] = tr_frag(pf_img, 2 * pf_qh, 1); }
  };
  auto x_group = [&](int par, unsigned img, int qh_n, bool fill, bool use_pre, int pf, unsigned pf_img, int pf_qh, bool pf_stat,
                     unsigned pf_st, int pf_st_qh, const bool dma = false, const bool hand = false) __attribute__((always_inline)) {
    bf16x8 fr[4];
    if (use_pre) { fr[0] = fr_pre[0]; fr[1] = fr_pre[1]; }
    else { fr[0] = frag(img, 0, qh_n); fr[1] = frag(img, 1, qh_n); }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ds = s >> 1, kb = s & 1;
      if (KVP_DMA_SPREAD == 1 && dma && s >= 1 && s <= 9) dma_tile_piece(it, s - 1);
      if (KVP_DMA_SPREAD == 2 && dma && (s & 1)) dma_tile_piece(it, s >> 1);
      if (KVP_DMA_SPREAD == 2 && dma && s == 14) dma_tile_piece(it, 8);
      if (kb == 0 && ds + 2 < 8) fr[(ds + 2) & 3] = frag(img, ds + 2, qh_n);
      if (s == 12) { prefetch(pf, pf_img, pf_qh); if (pf_stat) stat_fetch(pf_st, pf_st_qh); }
      if (ds == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        xb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], wf[kb][ds], z, 0, 0, 0);
      } else {
        xb[par ^ 1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ds & 3], wf[kb][ds], xb[par ^ 1][kb], 0, 0, 0);
      }
      if (ROLE_B && fill && s >= 2 && ((s & 1) == 0 || s == 15)) {
        // B: the eight dS pairs the dK group's FIRST eight MFMAs read (k-step 0: pairs 0 .. 3 of both key blocks), every other slot,
        // from slot 2 on (P was read after the barrier); the other eight are formed under those MFMAs (g_group)
        const int i = s == 15 ? 7 : (s - 2) >> 1;
        ds_pair(par, (i >> 2) * 8 + (i & 3));
      }
      if (!ROLE_B && fill) {                                  // A: elements 16 .. 31 of S(u + 1), one exp2 per slot; a pair is packed a slot late
        exp_one(par, 16 + s);
        if (s >= 2 && (s & 1) == 0) pack_pair(par, 8 + ((s - 2) >> 1));
        // pairs 0 .. 7 were packed under the previous group, pairs 8 .. 11 by slot 8
        if (KVP_HAND_SPREAD && hand) {
          if (s == 1) hand_over_part(par, 0);
          if (s == 5) hand_over_part(par, 1);
          if (s == 11) hand_over_part(par, 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!ROLE_B && fill) {
      pack_pair(par, 15);
      if (KVP_HAND_SPREAD && hand) hand_over_part(par, 3);
    }
  };
  // 16 slots: gradient^T += X^T(half qh of the image at `img`, transposed reads) packed(par)   (A: dV^T += dO^T P, B: dK^T += Q^T dS);
  // A with FILL: exp2 of pairs 0 .. 7 of buffer par ^ 1 (even slots) and their bf16 pack (odd slots: an exp2 result is never consumed
  // by the next instruction) behind the MFMAs
  auto g_group = [&](int par, unsigned img, int qh, bool fill, bool use_pre, int pf, unsigned pf_img, int pf_qh, bool pf_stat,
                     unsigned pf_st, int pf_st_qh, const bool dma = false) __attribute__((always_inline)) {
    bf16x8 tr[4];
    if (use_pre) { tr[0] = fr_pre[0]; tr[1] = fr_pre[1]; }
    else { tr[0] = tr_frag(img, 2 * qh, 0); tr[1] = tr_frag(img, 2 * qh, 1); }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int i = s >> 1, kb = s & 1, t2 = i >> 2, db = i & 3;
      if (KVP_DMA_SPREAD == 1 && dma && s >= 1 && s <= 9) dma_tile_piece(it, s - 1);
      if (KVP_DMA_SPREAD == 2 && dma && (s & 1)) dma_tile_piece(it, s >> 1);
      if (KVP_DMA_SPREAD == 2 && dma && s == 14) dma_tile_piece(it, 8);
      if (kb == 0 && i + 2 < 8) tr[(i + 2) & 3] = tr_frag(img, 2 * qh + ((i + 2) >> 2), (i + 2) & 3);
      if (s == 12) { prefetch(pf, pf_img, pf_qh); if (pf_stat) stat_fetch(pf_st, pf_st_qh); }
      const u32x4 pw = {pk[par][kb][t2][0], pk[par][kb][t2][1], pk[par][kb][t2][2], pk[par][kb][t2][3]};
      const bf16x8 pf8 = __builtin_bit_cast(bf16x8, pw);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[kb][db]) : "v"(tr[i & 3]), "v"(pf8));
      if (!ROLE_B && fill) {                                  // A: elements 0 .. 15 of S(u + 1)
        exp_one(par ^ 1, s);
        if (s >= 2 && (s & 1) == 0) pack_pair(par ^ 1, (s - 2) >> 1);
      }
      // B: dS pairs 4 .. 7 (k-step 1) of key block 0 under slots 0 .. 3, of key block 1 under slots 4 .. 7; first read by slots 8 / 9
      if (ROLE_B && fill && s < 8) ds_pair(par, (s >> 2) * 8 + 4 + (s & 3));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!ROLE_B && fill) pack_pair(par ^ 1, 7);
  };
  // VALU result -> inline-asm MFMA operand: wait states the compiler does not know are needed, tied to the operands
  auto settle = [&](int par) __attribute__((always_inline)) {
    asm volatile("s_nop 4" : "+v"(pk[par][0][0][0]), "+v"(pk[par][0][0][1]), "+v"(pk[par][0][0][2]), "+v"(pk[par][0][0][3]),
                 "+v"(pk[par][1][0][0]), "+v"(pk[par][1][0][1]), "+v"(pk[par][1][0][2]), "+v"(pk[par][1][0][3]));
  };
  auto pair_barrier = [&]() __attribute__((always_inline)) {       // LDS hand-over visible to the partner (whole workgroup: one barrier kind)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL == 3 || ABL == 6) return;
    __builtin_amdgcn_s_barrier();
  };

  // ---- the tile pipeline's cursor: ONE full iterator (two tiles ahead of the arithmetic: it feeds the LDS-DMA) + the mask rows of the
  // tiles in flight ------------------------------------------------------------------------------------------------------------------------
  int m_cur = -1, m_nx1 = -1, m_nx2 = -1;                // mask_row of tile t / t + 1 / t + 2
  int s3 = 0, s3n = 1, s3nn = 2, s3n3 = 3;               // ring slots of tiles t, t+1, t+2, t+3
  const unsigned IMG_X = ROLE_B ? LDS_DO : LDS_Q;        // the image this role reads as fragments (dO for dP, Q for S)
  const unsigned IMG_G = ROLE_B ? LDS_Q : LDS_DO;        // ... and transposed (Q^T for dK, dO^T for dV)
  // ---- one tile = two trips (half qh = buffer parity qh) ------------------------------------------------------------------------
  // Every 16-MFMA group starts from fragments its predecessor read at slot 12 (fr_pre) and the statistics of a half are fetched a trip
  // ahead (stat_raw), so that neither the group after a barrier nor the second group of a trip waits for an LDS round trip before its
  // first MFMA (r04 first form: 11 ds_reads and an s_waitcnt between s_barrier and the first MFMA of every trip).
  // The LDS-DMA of tile t + 3 is issued at the top of tile t and has to have LANDED at the end of tile t + 1 (tile t + 2 is read from the
  // start of tile t + 1 on: A runs two halves ahead): two tile times, ~4 us.  r04 kept a ring of three and waited `vmcnt(0)` at the end
  // of the tile that issued the DMA — one tile time, ~2 us.  Same-box A / B of the two rings: equal within the noise (5.09 vs 5.12 ms);
  // the longer ring is kept for the context-parallel case, where a tile of a remote chunk may come from further away.
  auto iteration = [&](const bool has1, const bool has3) __attribute__((always_inline)) {
    int m_nx3 = -1;
    if (DMA_SPREAD) dma_slot3 = s3n3;
    else if (has3) dma_tile(it, s3n3);                   // tile t + 3 -> the slot that held tile t - 1 (last read before the previous tile barrier)
    // the cursor's own step (compares, 64-bit adds, two branches: ~15 scalar instructions, more at a chunk boundary) goes BEHIND the first
    // 16-MFMA group of the trip, where the matrix pipe is still draining: at the top of the trip — straight after the barrier, in all four
    // waves at once — it sat on the critical path of every tile (r05 first form: + 5 % SQ_WAVE_CYCLES)
    auto step_cursor = [&]() __attribute__((always_inline)) {
      if (has3) { m_nx3 = mask_row(it); advance(it); }
    };
    const unsigned x_cur = lds0 + IMG_X + s3 * TILEB, x_nxt = lds0 + IMG_X + s3n * TILEB;
    const unsigned g_cur = lds0 + IMG_G + s3 * TILEB, g_nxt = lds0 + IMG_G + s3n * TILEB;
    const unsigned st_cur = lds0 + LDS_ST + s3 * 512, st_nxt = lds0 + LDS_ST + s3n * 512;
    // ---- trip 0: half u = (tile t, rows 0 ..) in buffers 0; u + 1 = (tile t, rows 32 ..); u + 2 = (tile t + 1, rows 0 ..) if has1 ---------
    if (ROLE_B) {
      take_over(0);                                      // P(u), written by A before the last barrier
      stat_finish();                                     // delta of half u (fetched under the previous trip's dK group)
      x_group(0, x_cur, 1, true, true, PF_TR, g_cur, 0, false, 0, 0, has3);                        // dP(u + 1)  ||  dS(u)  [+ the DMA of tile t + 3]
      step_cursor();
      settle(0);
      g_group(0, g_cur, 0, true, true, has1 ? PF_FRAG : PF_TR, has1 ? x_nxt : g_cur, has1 ? 0 : 1, true, st_cur, 1);   // dK^T += Q^T dS(u)  ||  dS(u), k-step 1
    } else {
      // A holds P(u) packed in pk[0] and the raw (masked) S(u + 1) in buffers 1
      stat_finish();                                     // lse of half u + 1
      g_group(0, g_cur, 0, true, true, has1 ? PF_FRAG : PF_TR, has1 ? x_nxt : g_cur, has1 ? 0 : 1, false, 0, 0, has3);   // dV^T += dO^T P(u) || exp2 0 .. 7 of S(u + 1)
      step_cursor();
      if (has1) {
        x_group(1, x_nxt, 0, true, true, PF_TR, g_cur, 1, true, st_nxt, 0, false, true);         // S(u + 2) -> buffers 0  ||  pairs 8 .. 15 of S(u + 1)  [+ P(u + 1) -> the partner]
      } else {
#pragma unroll
        for (int e = 8; e < 16; ++e) exp_pair(1, e);
#pragma unroll
        for (int e = 8; e < 16; ++e) pack_pair(1, e);
      }
      if (!(KVP_HAND_SPREAD && has1)) hand_over(1);      // P(u + 1) -> the partner, before the barrier that ends this trip
      if (has1 && m_nx1 >= 0) mask_half(0, m_nx1);                                       // wave-uniform, diagonal tiles only
    }
    if (ABL != 1) pair_barrier();
    // ---- trip 1: half u = (tile t, rows 32 ..) in buffers 1; u + 1, u + 2 = the two halves of tile t + 1 if has1 ---------------------------
    if (ROLE_B) {
      take_over(1);
      stat_finish();
      if (has1) {
        x_group(1, x_nxt, 0, true, true, PF_TR, g_cur, 1, false, 0, 0);                            // dP(u + 1)  ||  dS(u)
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) ds_pair(1, e);
      }
      settle(1);
      g_group(1, g_cur, 1, has1, true, has1 ? PF_FRAG : PF_NONE, x_nxt, 1, has1, st_nxt, 0);       // next: trip 0 of tile t + 1 (dO rows 32 .., delta of rows 0 ..)
    } else {
      if (has1) stat_finish();                           // lse of (tile t + 1, rows 0 ..)
      g_group(1, g_cur, 1, has1, true, has1 ? PF_FRAG : PF_NONE, x_nxt, 1, false, 0, 0);
      if (has1) {
        x_group(0, x_nxt, 1, true, true, PF_TR, g_nxt, 0, true, st_nxt, 1, false, true);         // S(u + 2) -> buffers 1; next: trip 0 of tile t + 1
        if (!KVP_HAND_SPREAD) hand_over(0);
        if (m_nx1 >= 0) mask_half(1, m_nx1 + 32);
      }
    }
    // tile t + 2 has landed (every wave waits for ITS pieces, then the barrier): all but the pieces issued at the top of this tile —
    // 8 per wave, + 1 (lse / delta) in waves 0 and 1
    if (has3) {
      if (wave < 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (ABL != 3 && ABL != 6) __syncthreads();
    m_cur = m_nx1; m_nx1 = m_nx2; m_nx2 = m_nx3;
    const int tmp = s3; s3 = s3n; s3n = s3nn; s3nn = s3n3; s3n3 = tmp;
  };

  // ---- the two key blocks of the pair: global block `bid` walked DOWN from the last query tile, block NB - 1 - bid walked UP to it ----------
  // (Measured null, same box: starting the 32 workgroups of an XCD 300 .. 4200 cycles apart, so that one of them misses and the others hit
  // instead of 32 hits-under-miss on the same lines, changes nothing: 5.13 - 5.25 ms at every spacing.)
#pragma nounroll
  for (int phase = 0; phase < 2; ++phase) {
    const int gb = phase == 0 ? bid : NB - 1 - bid;      // block index in global key order
    if (phase == 1 && gb <= bid) break;                  // NB odd: the middle block has no partner
    dir = phase == 0 ? -1 : 1;
    q_hstep = dir * p.q_hs * 2; do_hstep = dir * p.do_hs * 2; st_hstep = dir * p.n_q_rows;
    q_tstep = dir * ((int64_t)QT * p.q_rs - (G - 1) * p.q_hs) * 2;
    do_tstep = dir * ((int64_t)QT * p.do_rs - (G - 1) * p.do_hs) * 2;
    st_tstep = dir * (QT - (G - 1) * p.n_q_rows);
    int kc = 0;                                          // the chunk (buffer order) whose gid has rank gb / kt_per_chunk
    for (int c = 0; c < p.n_kv_chunks; ++c) {
      int rank = 0;
      for (int e = 0; e < p.n_kv_chunks; ++e) rank += p.kv_gid[e] < p.kv_gid[c];
      if (rank == gb / kt_per_chunk) kc = c;
    }
    gk = p.kv_gid[kc];
    k_off_wg = (gb % kt_per_chunk) * KWG;                // first key of the block inside its chunk
    k_off = k_off_wg + pair * 64;                        // this PAIR's first key inside the chunk
    k_row0 = p.kv_row[kc] + k_off;                       // its row in the K / V buffers
    j0 = k_off_wg / QT;                                  // first tile of the own chunk whose rows reach the block's keys
    int n_tiles = 0;                                     // 0, or >= 2 G: the own chunk contributes at least two tiles
    for (int c = 0; c < p.n_q_chunks; ++c) {
      const int gq = p.q_gid[c];
      n_tiles += gq > gk ? q_tiles_per_chunk : (gq == gk ? q_tiles_per_chunk - j0 : 0);
    }
    n_tiles *= G;
    bf16_t* const out = ROLE_B ? p.dk + (int64_t)kvh * p.dk_hs : p.dv + (int64_t)kvh * p.dv_hs;
    const int64_t out_rs = ROLE_B ? p.dk_rs : p.dv_rs;
    if (n_tiles == 0) {                                  // context parallelism: a key chunk none of the local queries can see
      const u32x2 z = {0u, 0u};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16_t* op = out + (k_row0 + 32 * kb + l31) * out_rs;
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x2*>(op + 8 * i + 4 * hi) = z;
      }
      continue;
    }
    load_own_keys();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[kb][db][r] = 0.f;
        asm volatile("" : "+a"(o[kb][db]));
      }

    // ---- prologue: tiles 0 and 1 -> LDS; X of half 0; A: P(0) packed and handed over -------------------------------------------------
    it.hq = 0; it.c = dir > 0 ? 0 : p.n_q_chunks - 1; it.j = 0; it.jlo = 0; it.diag = 0; it.stat = 0; it.qp = qbase; it.dop = qbase;
    enter(it);
    s3 = 0; s3n = 1; s3nn = 2; s3n3 = 3;
    m_cur = mask_row(it); dma_tile(it, 0); advance(it);
    m_nx1 = mask_row(it); dma_tile(it, 1); advance(it);  // n_tiles >= 2
    m_nx2 = -1;
    if (n_tiles > 2) {                                   // tile 2 stays in flight under the prologue; `it` then stands on tile 3 (or past the end)
      m_nx2 = mask_row(it); dma_tile(it, 2); advance(it);
      if (wave < 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // B: dP(0) -> buffers 0, and under it the first fragments of trip 0's x_group (dO rows 32 ..) + delta of half 0
    x_group(1, lds0 + IMG_X, 0, false, false, ROLE_B ? PF_FRAG : PF_NONE, lds0 + IMG_X, 1, ROLE_B, lds0 + LDS_ST, 0);
    if (!ROLE_B) {
      if (m_cur >= 0) mask_half(0, m_cur);
      load_stat(lds0 + LDS_ST, 0);
  #pragma unroll
      for (int e = 0; e < 16; ++e) exp_pair(0, e);
  #pragma unroll
      for (int e = 0; e < 16; ++e) pack_pair(0, e);
      hand_over(0);
      settle(0);
      // A runs two halves ahead: S(1) -> buffers 1 (tile 0 always has both halves); under it the first dO^T fragments of trip 0's g_group
      // and the lse of half 1
      x_group(0, lds0 + IMG_X, 1, false, false, PF_TR, lds0 + IMG_G, 0, true, lds0 + LDS_ST, 1);
      if (m_cur >= 0) mask_half(1, m_cur + 32);
    }
    pair_barrier();


    int t = 0;
    for (; t + 3 < n_tiles; ++t) iteration(true, true);
    for (; t + 1 < n_tiles; ++t) iteration(true, false);   // the last one or two tiles that still have a successor
    iteration(false, false);

    // ---- epilogue: dK / dV [key][d] -----------------------------------------------------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]),
                 "+a"(o[1][2]), "+a"(o[1][3]));
  #pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16_t* op = out + (k_row0 + 32 * kb + l31) * out_rs;
  #pragma unroll
      for (int db = 0; db < 4; ++db) store_row_block32(op + 32 * db, o[kb][db], 1.0f, hi);    // two 16-byte stores per block (r06)
    }
    // the counted `vmcnt` waits of the next key block assume that nothing but its own LDS-DMA is in flight: drain the stores above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // the LDS images and the hand-over buffers are free for the next key block
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
  const int64_t nb = (int64_t)a.n_kv_chunks * (a.chunk_len / KWG);         // 128-key blocks per kv head; one workgroup per PAIR of them
  const int64_t n = (int64_t)a.n_kv_heads * ((nb + 1) / 2);
  if (n > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(attn_bwd_kvp_kernel, dim3((unsigned)n), dim3(256), LDS_TOTAL, st, a);
  return vita_check_launch();
}
