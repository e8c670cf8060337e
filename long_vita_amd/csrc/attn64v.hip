// Flash attention forward, d = 64, non-causal, ragged rows and keys: 4 waves x 64 query rows, one wave per SIMD (gfx950 / MI355X) — r05.
// The vision towers' attention: InternViT-300M (16 heads x 64, 1025 tokens per frame; M/core/transformer/dot_product_attention.py:312-329
// calls flash_attn_func(q, k, v, causal = False) per batch of frames) and any other head-size-64 non-causal call of vita_flash_attn_fwd.
//
// attn64.hip's in-wave software pipeline (its header explains the structure: S^T = K Q^T in VGPR form, O^T pinned in AGPRs, P packed in
// the accumulator's own lane layout, lazy running maximum, K / V tiles by LDS-DMA into swizzled two-slot rings, the softmax dealt to the
// MFMA slots by weight) at head size 64, where a 64-key tile is 16 + 16 MFMAs per wave instead of 32 + 32 under the SAME softmax work:
// ~8.5 VALU instructions per MFMA instead of ~4.2, so the phases are VALU-paced and the deal puts two to three filler units behind every
// MFMA.  The r01 kernel this replaces for the ViT (attn.hip: 8 waves x 32 rows, two waves per SIMD, softmax between the MFMA clusters) ran
// 0.515 PFLOP/s on the 253-frame batch.
// What is new against attn64.hip:
//   * any number of key tiles >= 1 (17 for 1025 tokens: odd), the last one partial: its K / V rows beyond the sequence are never read —
//     the tile's buffer descriptor ends at the last valid row, the LDS-DMA returns zeros beyond it — and its scores are masked
//     arithmetically (the same min(lim - key, 0) * 3e38 penalty as attn64's diagonal);
//   * query rows beyond the sequence (the 1025th row opens a fifth 256-row workgroup per frame and head): loads clamp to the last
//     valid row, nothing is stored for them;
//   * a batch dimension (frames) in the work decomposition; no chunk tables (one chunk), no causal mask, no packed samples.
#include "attn_args.h"
#include <stdlib.h>

namespace {

constexpr int D = 64, KVT = 64, QTILE = 256, ROWB = D * 2, TILEB = KVT * ROWB;       // 8 KiB per K (or V) tile
constexpr int NSLOT = 4, LDS_K = 0, LDS_V = NSLOT * TILEB, LDS_TOTAL = 2 * NSLOT * TILEB;     // rings of four 8 KiB slots: tile j in slot j & 3
constexpr int NF2 = 3;                                                               // P fragments of tile t+1 done in phase 2
constexpr int THR = 8;                                                               // lazy running maximum, log2 units
constexpr int NS = 16;                                                               // MFMA slots per phase

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;

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

// ---- filler schedule: units dealt to the NS MFMA slots of a phase by cumulative weight (attn64.hip's, over 16 slots) ---------------
// phase 2 units: 0..31 max3 steps (weight 2), 32..33 running-maximum decision (weight 12), then 8 * NF2 exp half-units
// (half 0: two fma + two exp2: weight 8; half 1: two row-sum adds + one bf16 pack: weight 6); phase 1 units: 8 * (8 - NF2) exp half-units
struct SlotMap { int first[NS + 1]; };
constexpr int unit_w2(int u) { return u < 32 ? 2 : (u < 34 ? 12 : ((u & 1) ? 6 : 8)); }
constexpr SlotMap make_map2() {
  SlotMap m{};
  const int n = 34 + 8 * NF2;
  int tot = 0;
  for (int u = 0; u < n; ++u) tot += unit_w2(u);
  int acc = 0, u = 0;
  for (int s = 0; s < NS; ++s) {
    m.first[s] = u;
    const int lim = (tot * (s + 1) + NS - 1) / NS;
    while (u < n && acc + unit_w2(u) <= lim) { acc += unit_w2(u); ++u; }
  }
  m.first[NS] = n;
  return m;
}
constexpr SlotMap make_map1() {
  SlotMap m{};
  const int n = 8 * (8 - NF2);
  for (int s = 0; s <= NS; ++s) m.first[s] = (n * s) / NS;
  return m;
}

// NQB = 2: every wave works on its two 32-row blocks.  NQB = 1: only block 0 of every wave — the workgroup that holds the sequence's last
// <= 32 rows (the ViT's 1025th token: one workgroup in five).  Its waves 1 .. 3 own no valid row and wave 0 only its first block; they all
// take the same path (same barriers, same LDS-DMA shares) with half the MFMAs and half the softmax, and the workgroup leaves its CU in
// ~55 % of the time.
// NW = 4: 64 rows per wave (NQB = 2), one wave per SIMD — the shipped form.  NW = 8 (NQB = 1 only; developer switch VITA_ATTN64V=8):
// eight waves x 32 rows, two waves per SIMD in 252 registers — built to test whether a second wave's issue slots help where a single wave
// cannot issue its ~13 instructions per MFMA inside the MFMA's 32 cycles.  They do not: 0.521 vs 0.477 ms.  SQ counters of the 4-wave
// form at 253 frames (profiles/r05_vit_attn_pmc.txt): 11.1 VALU instructions per MFMA (attn64: 5.7), the wave issues 66 % of its cycles,
// VALU active 53 %, matrix pipe busy 32 %, s_waitcnt / s_barrier 19.5 % — a deeper DMA ring (four slots, counted vmcnt: kept) did not
// move the last figure, so it is barrier skew between the waves, not load latency.  At head size 64 the softmax (one exp2, one fma, one
// add, half a max3 and half a pack per score) costs more issue time than the 8 MFMAs per 32 x 32 score tile take: the kernel is bound by
// VALU issue, and 0.85 PFLOP/s of MFMA work is out of reach for an fp32 softmax on this chip (16-lane SIMDs: 4 cycles per VALU
// instruction; the r01 kernel, the 8-wave form and this one all land within 12 % of each other).
template <int NQB, int NW>
__device__ __forceinline__ void fwd64v_body(const AttnArgs& p, const unsigned lds0, const int wave, const int lane, const int b, const int kvh,
                                            const int hq, const int qti) {
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- work decomposition: kv head innermost (the 256-row tiles of one frame and head are 16 x k blocks apart: the same XCD, close in
  // time — they share that head's K / V through the XCD's L2) ---------------------------------------------------------------------------
  const int G = p.n_q_heads / p.n_kv_heads;
  const int head = kvh * G + hq;
  constexpr int PW = 8 / NW;                        // 1 KiB LDS-DMA pieces of a K (or V) tile per wave
  const int q_off = qti * QTILE + wave * (QTILE / NW);   // this wave's first row
  const int q_last = p.q_valid - 1;
  const float scale_log2e = p.scale_log2e;

  // ---- Q fragments (B operand of S^T = K Q^T): block qb, k-step ds: row q_off + 32 qb + l31 (clamped), d = 16 ds + 8 hi .. + 7 --------
  bf16x8 qf[2][4];
  {
    const bf16_t* qb0 = p.q + (int64_t)b * p.q_bs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = min(q_off + 32 * qb + l31, q_last);
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) qf[qb][ds] = *reinterpret_cast<const bf16x8*>(qb0 + (int64_t)row * p.q_rs + ds * 16);
    }
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) asm volatile("" : "+a"(qf[qb][ds]));      // live in AGPRs from here on

  // ---- LDS fragment offsets (attn.hip's d = 64 layouts: K slot ^ ((row >> 1) & 7), V 32-byte chunk ^ (row & 2)) -----------------------
  unsigned koff[4], voff[2];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) koff[ds] = l31 * ROWB + (((2 * ds + hi) ^ ((l31 >> 1) & 7)) << 4);    // + 32 kb rows: immediate
  {
    const int g16 = lane >> 4, i16 = lane & 15, key_l = 4 * (g16 >> 1) + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
      voff[db] = key_l * ROWB + (((col >> 4) ^ (key_l & 2)) << 5) + (col & 15) * 2;
    }
  }
  // ---- LDS-DMA: a tile = 8 pieces of 1 KiB = 8 rows each; wave w moves pieces 2w, 2w + 1 of K and of V; swizzle on the SOURCE ----------
  unsigned dk_off[PW], dv_off[PW];
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const int row = (wave * PW + q) * 8 + (lane >> 3), ps = lane & 7;
    const int ks = ps ^ ((row >> 1) & 7);
    const int vs = (((ps >> 1) ^ (row & 2)) << 1) | (ps & 1);
    dk_off[q] = (unsigned)((row * p.k_rs + ks * 8) * 2);     // bytes inside the tile (64 rows x row stride < 2^31: checked on the host)
    dv_off[q] = (unsigned)((row * p.v_rs + vs * 8) * 2);
  }
  const char* kbase = (const char*)(p.k + (int64_t)b * p.k_bs + (int64_t)kvh * p.k_hs);
  const char* vbase = (const char*)(p.v + (int64_t)b * p.v_bs + (int64_t)kvh * p.v_hs);
  const int k_tile_bytes = (int)(p.k_rs * 2 * KVT), v_tile_bytes = (int)(p.v_rs * 2 * KVT);
  const int n_tiles = (p.kv_valid + KVT - 1) / KVT;
  const int last_rows = p.kv_valid - (n_tiles - 1) * KVT;            // valid keys of the last tile, 1 .. 64
  const unsigned lds_kw = lds0 + LDS_K + wave * (PW * 1024), lds_vw = lds0 + LDS_V + wave * (PW * 1024);
  // the descriptor is re-based on the tile's first row and ENDS behind its last valid row (num_records = rows x row stride): rows of the
  // last tile beyond the sequence are out of range for the buffer unit and land as zeros — no read past the tensor, no clamped copies
  auto rsrc_rows = [&](const char* base, int rows, int64_t rs) __attribute__((always_inline)) {
    vita_rsrc_t r = vita_make_rsrc_uniform(base);
    r[2] = (unsigned)(rows * rs * 2);
    return r;
  };
  auto dma_k = [&](int j, int slot) __attribute__((always_inline)) {
    const vita_rsrc_t r = rsrc_rows(kbase + (int64_t)j * k_tile_bytes, j == n_tiles - 1 ? last_rows : KVT, p.k_rs);
    unsigned base = lds_kw;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < PW; ++q) vita_lds_dma16(r, dk_off[q], base + slot * TILEB + q * 1024);
  };
  auto dma_v = [&](int j, int slot) __attribute__((always_inline)) {
    const vita_rsrc_t r = rsrc_rows(vbase + (int64_t)j * v_tile_bytes, j == n_tiles - 1 ? last_rows : KVT, p.v_rs);
    unsigned base = lds_vw;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int q = 0; q < PW; ++q) vita_lds_dma16(r, dv_off[q], base + slot * TILEB + q * 1024);
  };

  // ---- state ----------------------------------------------------------------------------------------------------------------------------
  f32x16 o[2][2];                                    // O^T[qb][db]: d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi, row 32 qb + l31 (AGPRs)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
      asm volatile("" : "+a"(o[qb][db]));
    }
  f32x16 sb[2][2][2];                                // S^T[parity][qb][kb]: key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi
  unsigned pk[2][2][4][4];                           // packed P^T[parity][qb][frag f][4 dwords]; frag f = regs 8 (f & 1) .. of kb = f >> 1
  float m_run[2] = {-1.0e30f, -1.0e30f}, l_run[2] = {0.f, 0.f}, m_neg[2], alpha[2] = {1.f, 1.f}, mxc[4];
  float l_tile[2] = {0.f, 0.f};
  float ea = 0.f, eb = 0.f, mx0_keep = 0.f;

  constexpr SlotMap MAP1 = make_map1(), MAP2 = make_map2();

  // exp half-units (64 per tile): h -> fragment g = h >> 3 (need order of P V: g = 2 f + qb), element pair pr = (h >> 1) & 3;
  // half 0: the two fma + exp2 of the pair, half 1: row sum, bf16 pack (an exp2 result is never consumed by the next instruction)
  auto exp_half = [&](int par, int h) __attribute__((always_inline)) {
    const int g = h >> 3, pr = (h >> 1) & 3, qb = g & 1, f = g >> 1, kb = f >> 1, r = 8 * (f & 1) + 2 * pr;
    if (qb >= NQB) return;
    if ((h & 1) == 0) {
      ea = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][kb][r], scale_log2e, m_neg[qb]));
      eb = __builtin_amdgcn_exp2f(fmaf(sb[par][qb][kb][r + 1], scale_log2e, m_neg[qb]));
    } else {
      l_tile[qb] += ea;
      l_tile[qb] += eb;
      pk[par][qb][f][pr] = pack_bf16x2(ea, eb);
      if (h == 55 || h == 63) {                      // the last half-unit of block qb (fragments g = 6 / 7): fold the tile's sum
        l_run[qb] += l_tile[qb];
        l_tile[qb] = 0.f;
      }
      asm volatile("" :: "v"(pk[par][qb][f][pr]), "v"(l_tile[qb]), "v"(l_run[qb]));            // computed HERE (no sinking past the phase)
    }
  };
  // the running-maximum decision of a tile: unit 32 keeps block 0's maximum, unit 33 decides for both blocks with ONE
  // wave-uniform flag (grow = some row exceeds its running maximum by more than 2^THR; the first tile always grows)
  auto max_unit = [&](int par, int u) __attribute__((always_inline)) {
    if (u < 32) {                                    // max3 steps: four chains (qb, kb): chain c = u & 3, step u >> 2
      const int c = u & 3, st = u >> 2, qb2 = c >> 1, kb2 = c & 1, r = 2 * st;
      if (qb2 >= NQB) return;
      const float a = sb[par][qb2][kb2][r], bb = sb[par][qb2][kb2][r + 1];
      mxc[c] = st == 0 ? fmaxf(a, bb) : fmaxf(fmaxf(a, bb), mxc[c]);
    } else if (u == 32) {
      mx0_keep = swap32_max(fmaxf(mxc[0], mxc[1])) * scale_log2e;
    } else {
      const float mx1 = NQB > 1 ? swap32_max(fmaxf(mxc[2], mxc[3])) * scale_log2e : -1.0e30f;
      const bool grow = __any((NQB > 1 && mx1 > m_run[1] + (float)THR) || (mx0_keep > m_run[0] + (float)THR));
#pragma unroll
      for (int qb2 = 0; qb2 < NQB; ++qb2) {
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
  auto v_frag = [&](unsigned vslot, int i) __attribute__((always_inline)) {        // i = 2 t + db: keys 16 t .., d block db
    const unsigned va = vslot + voff[i & 1] + 16 * (i >> 1) * ROWB;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va));
    const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(va + 8 * ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 ac = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, ac);
  };
  // S(buffer `dst`) = K(kslot) Q^T: slot = 4 ds + 2 kb + qb; K fragment (kb, ds) read two fragments ahead (ring of four);
  // FILL: the exp half-units 8 NF2 .. 63 of tile `par` go behind the MFMAs
  auto qk_phase = [&](int dst, unsigned kslot, bool fill, int par) __attribute__((always_inline)) {
    bf16x8 kr[4];
    kr[0] = k_frag(kslot, 0); kr[1] = k_frag(kslot, 1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int i = s >> 1, qb = s & 1, ds = i >> 1, kb = i & 1;
      if (qb == 0 && i + 2 < 8) kr[(i + 2) & 3] = k_frag(kslot, i + 2);
      if (qb < NQB) {
        if (ds == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sb[dst][qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[qb][ds], z, 0, 0, 0);
        } else {
          sb[dst][qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i & 3], qf[qb][ds], sb[dst][qb][kb], 0, 0, 0);
        }
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
    for (int s = 0; s < NS; ++s) {
      const int i = s >> 1, qb = s & 1, t = i >> 1, db = i & 1;
      if (qb == 0 && i + 2 < 8) vr[(i + 2) & 3] = v_frag(vslot, i + 2);
      const u32x4 pw = {pk[par][qb][t][0], pk[par][qb][t][1], pk[par][qb][t][2], pk[par][qb][t][3]};
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      if (qb < NQB) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(vr[i & 3]), "v"(pf));
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
  // the last tile's keys beyond the sequence (key >= last_rows): s += min(lim - key, 0) * 3e38 — exp2 of it is 0, no maximum sees it
  auto mask_last = [&](int par) __attribute__((always_inline)) {
    const int base = last_rows - 1 - 4 * hi;                                      // key <= last_rows - 1 visible; key = const(kb, r) + 4 hi
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = 32 * kb + (r & 3) + 8 * (r >> 2);
          const float pen = fminf((float)(base - kc), 0.0f);
          sb[par][qb][kb][r] = fmaf(pen, 3.0e38f, sb[par][qb][kb][r]);
        }
  };
  // O *= alpha (rare: only when a running maximum moved); every P V MFMA that precedes it has been issued
  auto rescale_o = [&]() __attribute__((always_inline)) {
    if (!__all(alpha[0] == 1.0f && (NQB == 1 || alpha[1] == 1.0f))) {
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));      // asm MFMA -> accumulator read
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha[qb];
          asm volatile("" : "+a"(o[qb][db]));
        }
      asm volatile("s_nop 7" ::: "memory");                                       // accumulator write -> asm MFMA read
    }
  };
  const bool ragged = last_rows < KVT;
  auto masks = [&](int j, int par) __attribute__((always_inline)) {              // wave-uniform condition
    if (ragged && j == n_tiles - 1) mask_last(par);
  };

  // ---- prologue: K(0), V(0), K(1) | V(1), K(2) -> LDS; S(0); the start of its softmax ---------------------------------------------------
  // A 64-key tile is only ~1.5 us of work at this head size — less than an L2 / HBM round trip under load — so the LDS-DMA runs TWO tiles
  // ahead (rings of four slots) and the end of a tile waits, with a counted vmcnt, for everything but the pieces issued during it
  // (first form: rings of two, vmcnt(0) per tile: the waves sat 19.5 % of their cycles in s_waitcnt / s_barrier).
  auto kslot = [&](int j) __attribute__((always_inline)) { return lds0 + LDS_K + (j & 3) * TILEB; };
  auto vslot = [&](int j) __attribute__((always_inline)) { return lds0 + LDS_V + (j & 3) * TILEB; };
  auto wait_all_but_last_tile = [&]() __attribute__((always_inline)) {            // 2 PW pieces per tile and wave
    if constexpr (PW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  dma_k(0, 0); dma_v(0, 0);
  if (n_tiles > 1) dma_k(1, 1);
  if (n_tiles > 2) {
    dma_v(1, 1); dma_k(2, 2);
    wait_all_but_last_tile();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  qk_phase(0, kslot(0), false, 0);
  masks(0, 0);
#pragma unroll
  for (int u = 0; u < 34 + 8 * NF2; ++u) {
    if (u < 34) max_unit(0, u);
    else exp_half(0, u - 34);
  }
  // (O is zero: no rescale for tile 0)

  // ---- main loop: full(par) — tile t sits in S / P buffer par = t & 1 (a compile-time constant): K(t + 3) and V(t + 2) -> their ring slots
  // (those of K(t - 1), V(t - 2): last read before the previous barrier), S(t + 1) -> buffer par ^ 1 under the rest of tile t's softmax, then
  // O += V(t)^T P(t)^T under the start of tile t + 1's softmax.  Tiles 0 .. n - 2 go through full(); the last tile finishes alone. ----------
  // (r05 null: issuing these pieces one behind every fourth MFMA of the S = K Q^T phase, which pays 1.5 - 3 % in attn64.hip, costs here —
  // new / r01 kernel on the same box 0.88 - 0.89 with the burst, 0.95 - 0.98 spread: at this head size the phase's slots are full of VALU.)
  int t = 0;
  auto full = [&](int par) __attribute__((always_inline)) {
    const bool more = t + 3 < n_tiles;                 // (n_tiles <= 2: V(1) is still to come)
    if (more) { dma_k(t + 3, (t + 3) & 3); dma_v(t + 2, (t + 2) & 3); }
    else if (t + 2 < n_tiles) dma_v(t + 2, (t + 2) & 3);
    else if (n_tiles == 2 && t == 0) dma_v(1, 1);
    qk_phase(par ^ 1, kslot(t + 1), true, par);
    masks(t + 1, par ^ 1);
    pv_phase(par, vslot(t), true);
    rescale_o();
    if (more) wait_all_but_last_tile();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++t;
  };
  while (t + 2 < n_tiles) {                            // at least two more tiles follow tile t
    full(0);
    full(1);
  }
  // asm MFMA -> accumulator read: the wait sits INSIDE each branch, straight behind its last MFMA — where the two branches join, the
  // compiler reconciles their register assignments with v_accvgpr_mov copies of the accumulators, and it does not know that the asm
  // statements in front of them are matrix instructions whose results are 16 passes away (first form: the copies read o[1][1] early)
  auto settle_o = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));
  };
  // ... and the same in front of a branch's first MFMA: the copies INTO the branch's registers are VALU writes of its accumulator operands
  auto settle_in = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 7" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));
  };
  if (t + 1 < n_tiles) {                               // one more follows: tile t (parity 0), the last one has parity 1
    full(0);
    finish_sm(1);
    settle_in();
    pv_phase(1, vslot(t), false);
    settle_o();
  } else {                                             // tile t (parity 0) is the last
    finish_sm(0);
    settle_in();
    pv_phase(0, vslot(t), false);
    settle_o();
  }

  // ---- epilogue: O[row][head][d] = O^T / l, lse; rows beyond the sequence are dropped -----------------------------------------------------
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const float l_tot = swap32_sum(l_run[qb]);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int orow = q_off + 32 * qb + l31;
    if (orow > q_last) continue;
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)orow * p.o_rs + (int64_t)kvh * p.o_gs + (int64_t)hq * p.o_hs;
#pragma unroll
    for (int db = 0; db < 2; ++db) store_row_block32(op + 32 * db, o[qb][db], inv, hi);       // two 16-byte stores per block (r06)
    if (p.lse && hi == 0) {
      const float lse = l_tot > 0.f ? (m_run[qb] + log2f(l_tot)) * 0.69314718055994530942f : -INFINITY;
      p.lse[((int64_t)b * p.n_q_heads + head) * p.n_q_rows + orow] = lse;
    }
  }
}

template <int NW>
__device__ __forceinline__ void fwd64v_entry(const AttnArgs& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = p.n_q_heads / p.n_kv_heads;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int hq = bid % G; bid /= G;
  const int n_qt = (p.q_valid + QTILE - 1) / QTILE;   // (one chunk: rows 0 .. q_valid - 1)
  const int qti = bid % n_qt;
  const int b = bid / n_qt;
  if constexpr (NW == 8) {
    fwd64v_body<1, 8>(p, lds0, wave, lane, b, kvh, hq, qti);
  } else {
    if (p.q_valid - qti * QTILE <= 32) fwd64v_body<1, 4>(p, lds0, wave, lane, b, kvh, hq, qti);
    else fwd64v_body<2, 4>(p, lds0, wave, lane, b, kvh, hq, qti);
  }
}

__global__ __launch_bounds__(256, 1) void flash_fwd64v_kernel(AttnArgs p) { fwd64v_entry<4>(p); }
__global__ __launch_bounds__(512, 2) void flash_fwd64v8_kernel(AttnArgs p) { fwd64v_entry<8>(p); }

}  // namespace

// head size 64, non-causal, one chunk (any number of rows / keys, any batch); everything else stays on attn.hip
bool vita_attn64v_eligible(const AttnArgs& a, int head_dim, bool causal) {
  if (head_dim != 64 || causal || a.seg_start) return false;
  if (a.n_q_chunks != 1 || a.n_kv_chunks != 1 || a.kv_row[0] != 0) return false;
  if (((uintptr_t)a.o & 15) || (a.o_rs & 7) || (a.o_hs & 7) || (a.o_gs & 7) || (a.o_bs & 7)) return false;       // 16-byte output stores (r06)
  // a tile's 64 rows x row stride is the buffer descriptor's 32-bit extent and the lanes' 32-bit offsets
  if (a.k_rs * 2 * KVT >= (1ll << 31) || a.v_rs * 2 * KVT >= (1ll << 31)) return false;
  const char* e = vita_dev_getenv("VITA_ATTN64V");                                 // developer A / B switch: 0 = the r01 kernel (attn.hip)
  return !(e && e[0] == '0');
}

int vita_attn64v_launch(const AttnArgs& a, hipStream_t st) {
  static std::atomic<unsigned long long> attr_set{0};
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64v_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd64v8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  });
  const int64_t nblocks = (int64_t)a.batch * a.n_q_heads * ((a.q_valid + QTILE - 1) / QTILE);
  if (nblocks > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  // measured (64 frames x 1025 tokens, 16 heads, same box): 4 waves x 64 rows 0.477 ms, 8 waves x 32 rows 0.521 ms, r01 kernel 0.543 ms
  const char* e = vita_dev_getenv("VITA_ATTN64V");                                 // developer A / B switch: 8 = the 8-wave x 32-row form
  if (e && e[0] == '8') hipLaunchKernelGGL(flash_fwd64v8_kernel, dim3((unsigned)nblocks), dim3(512), LDS_TOTAL, st, a);
  else hipLaunchKernelGGL(flash_fwd64v_kernel, dim3((unsigned)nblocks), dim3(256), LDS_TOTAL, st, a);
  return vita_check_launch();
}
