// Flash attention backward for gfx950 (head_dim 128, causal GQA with the zig-zag chunk geometry of
// the forward kernel).  Two kernels, both recomputing P from (Q, K, LSE) — no atomics, deterministic:
//
//   attn_bwd_dq_kernel   one workgroup = 128 query rows of one query head (4 waves x 32 rows), loops
//                        over the visible 64-key tiles:        dQ = sum_keys dS K
//   attn_bwd_dkv_kernel  one workgroup = 128 keys of one kv head (4 waves x 32 keys), loops over the
//                        G query heads of the group and their visible 32-row query tiles:
//                                                               dK = sum_q dS^T Q,  dV = sum_q P^T dO
//   with  P = exp(S*scale - LSE),  dP = dO V^T,  dS = P o (dP - D) * scale,  D = rowsum(dO o O).
//
// Replaces the autograd of flash-attn / TransformerEngine attention under the reference's training
// step (M/pretrain_long_vita.py:841-869 -> Megatron core attention backward; CP: the ring's dK/dV
// exchange becomes one reduce-scatter of the gathered-layout dK/dV buffer).
//
// One wave per SIMD (launch_bounds 256,1: 512 registers) — correctness-first structure; MFMA operand
// conventions are those of attn.hip (v_mfma_f32_32x32x16_bf16, any consistent k-slot assignment).
// LDS images: "frag" layout (16-byte slot ^ (row & 15), ds_read_b128 fragments) and "tr" layout
// (32-byte chunk ^ 2*(row & 3), ds_read_b64_tr_b16 transposed fragments); tiles are staged with
// the LDS-DMA, swizzles applied on the source address.
#include "attn_bwd_args.h"
#include <stdlib.h>

namespace {

constexpr int kMaxChunks = kBwdMaxChunks;
// r04: both kernels are templates on the head size HD = 128 (the decoder) or 64 (InternViT's attention: the ViT's backward through its
// native head size instead of zero-padded d = 128 copies, stage 2 trains the encoder).  What depends on HD: bytes per row (ROWB), the
// k-steps over d (HD / 16) and output blocks (HD / 32), rows per 1-KiB DMA piece, and the LDS swizzles below (a 128-byte row is half a
// bank period, so the row bits that select the slot move up by one).

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int HD> __device__ __forceinline__ int frag_key(int row) { return HD == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int HD> __device__ __forceinline__ int tr_key(int row) { return HD == 128 ? ((row & 3) << 1) : (((row >> 1) & 1) << 1); }
// r05: HD = 96 (SigLIP's 72, zero-padded to 96 instead of 128).  A 192-byte row is 3/4 of a bank period and has 12 slots, so the XOR keys do
// not apply: the fragment layout ROTATES the slot by (row >> 2) & 3 (mod 12), the transposed layout needs no swizzle (derivation: attn.hip).
template <int HD> __device__ __forceinline__ int frag_slot(int row, int slot) {          // logical -> physical 16-byte slot
  return HD == 96 ? (slot + ((row >> 2) & 3)) % 12 : (slot ^ frag_key<HD>(row));
}
template <int HD> __device__ __forceinline__ int frag_slot_inv(int row, int ps) {        // physical -> logical (the XOR forms are involutions)
  return HD == 96 ? (ps + 12 - ((row >> 2) & 3)) % 12 : (ps ^ frag_key<HD>(row));
}
template <int HD> __device__ __forceinline__ int tr_chunk(int row, int chunk) { return HD == 96 ? chunk : (chunk ^ tr_key<HD>(row)); }
template <int HD> __device__ __forceinline__ int frag_off(int row, int slot) { return row * (2 * HD) + (frag_slot<HD>(row, slot) << 4); }
template <int HD> __device__ __forceinline__ int tr_off(int row, int chunk, int b) { return row * (2 * HD) + (tr_chunk<HD>(row, chunk) << 5) + b; }

// DMA one 1-KiB piece (4 rows x 256 B) of a [rows][128] bf16 tile into LDS; tile = descriptor of its first row (wave-uniform),
// rows_valid = rows that exist (later ones are clamped, masked afterwards); lane -> (row piece*4 + lane/16, physical 16-B slot
// lane%16); `tr` selects the layout (source-side swizzle).  Issued from inline asm (vita_lds_dma16, see vita_common.h): through the
// builtin hipcc put an s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 of every tile — the prefetch of the next tile was
// a blocking load, the reason the round-1 kernels ran at 0.39 / 0.52 PFLOP/s.
template <int HD>
__device__ __forceinline__ unsigned dma_piece_voff(int64_t rs, int rows_valid, int piece, int lane, bool tr) {   // the lane's source offset (bytes)
  constexpr int LPR = HD / 8;                              // lanes (16-byte slots) per row: a 1-KiB piece is 64 / LPR rows (HD = 96: 5 1/3)
  const int unit = piece * 64 + lane;                      // 16-byte unit of the image this lane fills
  const int row = unit / LPR;
  const int ps = unit % LPR;
  const int ls = tr ? ((tr_chunk<HD>(row, ps >> 1) << 1) | (ps & 1)) : frag_slot_inv<HD>(row, ps);
  const int r = row < rows_valid ? row : rows_valid - 1;
  return (unsigned)(r * rs * 2 + ls * 16);
}
template <int HD>
__device__ __forceinline__ void dma_piece(vita_rsrc_t tile, int64_t rs, int rows_valid, int piece, int lane, bool tr,
                                          unsigned lds_dst) {
  vita_lds_dma16(tile, dma_piece_voff<HD>(rs, rows_valid, piece, lane, tr), (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_dst + piece * 1024)));
}
// 64 floats / ints: lane -> element lane % 32 of a wave-uniform array (both halves of the wave fetch the same 32)
__device__ __forceinline__ void dma_words32(vita_rsrc_t base, int lane, unsigned lds_dst) {
  vita_lds_dma4(base, (unsigned)((lane & 31) * 4), (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst));
}

template <int HD>
__device__ __forceinline__ bf16x8 read_frag(unsigned tile, int row, int slot) {
  return *(lds_bf16x8*)(uintptr_t)(tile + frag_off<HD>(row, slot));
}
// transposed fragment: rows r0 + {0..3} and r0 + 8 + {0..3} (r0 already includes 4*(lane>>5) and the
// lane's row inside its 16-lane group), 32 columns starting at 32*db
template <int HD>
__device__ __forceinline__ bf16x8 read_tr(unsigned tile, int lane, int row_base, int db) {
  const int g16 = lane >> 4, i16 = lane & 15;
  const int row = row_base + 4 * (g16 >> 1) + (i16 >> 2);
  const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(tile + tr_off<HD>(row, col >> 4, (col & 15) * 2)));
  const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(tile + tr_off<HD>(row + 8, col >> 4, (col & 15) * 2)));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)s[base + j];
  return r;
}

// ================================================================================================
// dQ kernel
// ================================================================================================
constexpr int QT_DQ = 128;      // query rows per workgroup
constexpr int KT_DQ = 64;       // keys per tile
// LDS per stage: K frag (16 KiB) | K tr (16 KiB) | V frag (16 KiB); a ring of three stages, tiles fetched two ahead (a tile of
// 48 MFMAs per wave is ~0.7 us of work against 1-2 us of memory latency: with one tile in flight the kernel waited on every tile)
constexpr int DQ_NSTAGE = 3;

template <int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(BwdArgs p) {
  constexpr int ROWB = 2 * HD, NDS = HD / 16, NDB = HD / 32;
  constexpr int DQ_STAGE = 3 * KT_DQ * ROWB;
  constexpr int PPW = KT_DQ * ROWB / 1024 / 4;             // 1-KiB pieces of a 64-row image per wave (4 / 2)
  constexpr int DQ_DMA_PER_STAGE = 3 * PPW;                // LDS-DMA instructions a wave issues per stage (3 images)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int G = p.n_q_heads / p.n_kv_heads;
  const int tiles_per_chunk = p.chunk_len / QT_DQ;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int hq = bid % G; bid /= G;
  const int qt = bid;                                   // 0 .. n_q_chunks*tiles_per_chunk-1, heavy first
  const int n_qt = p.n_q_chunks * tiles_per_chunk;
  const int qt_rev = n_qt - 1 - qt;
  const int qc = qt_rev / tiles_per_chunk;
  const int qti = qt_rev % tiles_per_chunk;
  const int gq = p.q_gid[qc];
  const int head = kvh * G + hq;
  const int q_off = qti * QT_DQ + wave * 32;            // wave's first row inside its chunk
  const int my_q = q_off + l31;
  const int64_t q_row = (int64_t)qc * p.chunk_len + my_q;
  const int q_last_wg = qti * QT_DQ + QT_DQ - 1;
  // packed sequences: keys before the row's segment are invisible; the workgroup starts at its first row's segment
  const int my_start = p.seg_start ? p.seg_start[q_row] : 0;
  const int wg_first_start = p.seg_start ? p.seg_start[(int64_t)qc * p.chunk_len + qti * QT_DQ] : 0;
  const int wg_last_start = p.seg_start ? p.seg_start[(int64_t)qc * p.chunk_len + q_last_wg] : 0;

  bf16x8 qf[NDS], dof[NDS];
  {
    const bf16_t* qp = p.q + q_row * p.q_rs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
    const bf16_t* dp = p.d_o + q_row * p.do_rs + (int64_t)head * p.do_hs + hi * 8;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
      dof[ds] = *reinterpret_cast<const bf16x8*>(dp + ds * 16);
    }
  }
  const float lse2 = p.lse[(int64_t)head * p.n_q_rows + q_row] * 1.44269504088896340736f;
  const float dlt = p.delta[(int64_t)head * p.n_q_rows + q_row];
  // consume the loads here: the compiler then waits for them HERE and not at their first use inside the loop, where its
  // s_waitcnt vmcnt() would also wait for the LDS-DMA pieces in flight (issued from asm, it does not know about them)
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) asm volatile("" :: "v"(qf[ds]), "v"(dof[ds]));
  asm volatile("" :: "v"(lse2), "v"(dlt));

  f32x16 dq_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

  const bf16_t* kbase = p.k + (int64_t)kvh * p.k_hs;
  const bf16_t* vbase = p.v + (int64_t)kvh * p.v_hs;

  auto chunk_tiles = [&](int c) __attribute__((always_inline)) -> int {
    const int all = p.chunk_len / KT_DQ;
    const int gk = p.kv_gid[c];
    if (gk < gq) return all;
    if (gk > gq) return 0;
    return min(all, q_last_wg / KT_DQ + 1);
  };
  // HD = 96: the lane's source offsets hold divisions by 12 — computed once (every tile is whole: chunk_len % 128 == 0)
  unsigned pre_kf[PPW], pre_kt[PPW], pre_vf[PPW];
  if constexpr (HD == 96) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      pre_kf[q] = dma_piece_voff<HD>(p.k_rs, KT_DQ, wave * PPW + q, lane, false);
      pre_kt[q] = dma_piece_voff<HD>(p.k_rs, KT_DQ, wave * PPW + q, lane, true);
      pre_vf[q] = dma_piece_voff<HD>(p.v_rs, KT_DQ, wave * PPW + q, lane, false);
    }
  }
  auto stage_tile = [&](int c, int j, unsigned sl) __attribute__((always_inline)) {
    const int64_t row0 = p.kv_row[c] + (int64_t)j * KT_DQ;
    const int valid = p.chunk_len - j * KT_DQ;
    const vita_rsrc_t kt0 = vita_make_rsrc(kbase + row0 * p.k_rs);
    const vita_rsrc_t vt0 = vita_make_rsrc(vbase + row0 * p.v_rs);
    // 16 (HD = 96: 12, HD = 64: 8) pieces per 64-row image; 4 waves -> PPW pieces each per image
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave * PPW + q;
      if constexpr (HD == 96) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sl + piece * 1024));
        vita_lds_dma16(kt0, pre_kf[q], dst);
        vita_lds_dma16(kt0, pre_kt[q], dst + KT_DQ * ROWB);
        vita_lds_dma16(vt0, pre_vf[q], dst + 2 * KT_DQ * ROWB);
      } else {
        dma_piece<HD>(kt0, p.k_rs, valid, piece, lane, false, sl);
        dma_piece<HD>(kt0, p.k_rs, valid, piece, lane, true, sl + KT_DQ * ROWB);
        dma_piece<HD>(vt0, p.v_rs, valid, piece, lane, false, sl + 2 * KT_DQ * ROWB);
      }
    }
  };

  // two iterators over the same tile sequence: `f` = the tile fetched next (two ahead), `cur` = the tile computed
  auto advance = [&](int& c, int& j, int& n) __attribute__((always_inline)) {
    if (++j == n) {
      j = 0;
      ++c;
      while (c < p.n_kv_chunks && (n = chunk_tiles(c)) == 0) ++c;
    }
  };
  int c_cur = 0, j_cur = wg_first_start / KT_DQ, n_cur = 0;
  while (c_cur < p.n_kv_chunks && (n_cur = chunk_tiles(c_cur)) == 0) ++c_cur;
  int c_f = c_cur, j_f = j_cur, n_f = n_cur;
  int fstage = 0;                                        // ring slot the next fetched tile goes to
  auto fetch_next = [&]() __attribute__((always_inline)) -> bool {
    if (c_f >= p.n_kv_chunks) return false;
    stage_tile(c_f, j_f, lds0 + fstage * DQ_STAGE);
    fstage = fstage + 1 == DQ_NSTAGE ? 0 : fstage + 1;
    advance(c_f, j_f, n_f);
    return true;
  };
  fetch_next();
  if (fetch_next()) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DQ_DMA_PER_STAGE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int stage = 0;
  while (c_cur < p.n_kv_chunks) {
    const bool issued = fetch_next();                    // tile t+2 -> the slot tile t-1 was read from (barrier at the end of t-1)

    const unsigned kf = lds0 + stage * DQ_STAGE, kt = kf + KT_DQ * ROWB, vf = kf + 2 * KT_DQ * ROWB;
    const int kv_off = j_cur * KT_DQ;
    const bool diag = p.kv_gid[c_cur] == gq;
    const bool skip = diag && kv_off > q_off + 31;
    if (!skip) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                      // two 32-key halves
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
          const bf16x8 ka = read_frag<HD>(kf, 32 * h + l31, 2 * ds + hi);
          const bf16x8 va = read_frag<HD>(vf, 32 * h + l31, 2 * ds + hi);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ds], s, 0, 0, 0);       // S^T[key, q]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ds], dp, 0, 0, 0);    // dP^T[key, q]
        }
        const bool need_mask = diag && kv_off + 32 * h + 31 > q_off;
        const bool seg_mask = kv_off + 32 * h < wg_last_start;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -lse2));      // P^T
        if (need_mask || seg_mask) {                     // wave-uniform: only tiles on the diagonal / at a sample boundary
          const int lim_hi = need_mask ? my_q : 0x7fffffff, lim_lo = seg_mask ? my_start : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv_off + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * hi;
            s[r] = ((key > lim_hi) | (key < lim_lo)) ? 0.f : s[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s[r] * (dp[r] - dlt) * p.scale;                                 // dS^T
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 dsf = pack8(s, 8 * t);
          // VALU result -> inline-asm MFMA operand: the compiler does not know the asm is an MFMA and inserts no wait states
          asm volatile("s_nop 4" : "+v"(dsf));
#pragma unroll
          for (int db = 0; db < NDB; ++db) {
            const bf16x8 ktf = read_tr<HD>(kt, lane, 32 * h + 16 * t, db);           // K^T[d, keys]
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dq_acc[db]) : "v"(ktf), "v"(dsf));
          }
        }
      }
    }
    // tile t+1 must have landed; the pieces of tile t+2 (issued above) may stay in flight
    if (issued) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DQ_DMA_PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage = stage + 1 == DQ_NSTAGE ? 0 : stage + 1;
    advance(c_cur, j_cur, n_cur);
  }

  // the inline-asm MFMAs are invisible to the compiler's hazard tracking: wait for the matrix pipe, accumulators tied to the wait
  if constexpr (NDB == 4) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dq_acc[0]), "+a"(dq_acc[1]), "+a"(dq_acc[2]), "+a"(dq_acc[3]));
  else if constexpr (NDB == 3) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dq_acc[0]), "+a"(dq_acc[1]), "+a"(dq_acc[2]));
  else asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dq_acc[0]), "+a"(dq_acc[1]));
  bf16_t* op = p.dq + q_row * p.dq_rs + (int64_t)kvh * p.dq_gs + (int64_t)hq * p.dq_hs;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 32 * db + 8 * rg + 4 * hi;
      u32x2 w = {pack_bf16x2(dq_acc[db][rg * 4 + 0], dq_acc[db][rg * 4 + 1]),
                 pack_bf16x2(dq_acc[db][rg * 4 + 2], dq_acc[db][rg * 4 + 3])};
      *reinterpret_cast<u32x2*>(op + d) = w;
    }
}

// ================================================================================================
// dK / dV kernel
// ================================================================================================
constexpr int KT_KV = 128;      // keys per workgroup (4 waves x 32)
constexpr int QT_KV = 32;       // query rows per step
// The workgroup's K / V fragments live in registers (they are the same for every step).  LDS: a ring of four stages of
// [Q frag 8 KiB | Q tr 8 | dO frag 8 | dO tr 8 | lse, delta, segment starts of the 32 rows (2 KiB)], steps fetched three ahead:
// a 32-row step is 32 MFMAs per wave (~0.45 us) against 1-2 us of memory latency.
constexpr int KV_NSTAGE = 4;

template <int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(BwdArgs p) {
  constexpr int ROWB = 2 * HD, NDS = HD / 16, NDB = HD / 32;
  constexpr int KV_STAGE = 4 * QT_KV * ROWB + 2048;
  constexpr int NPI = QT_KV * ROWB / 1024;                 // 1-KiB pieces of a 32-row image (8 / 6 / 4)
  constexpr bool WAVE_IMAGE = NPI % 4 != 0;                // HD = 96: six pieces do not deal to four waves — wave w stages image w whole
  constexpr int PPW = NPI / 4;                             // pieces of an image per wave otherwise (2 / 1)
  constexpr int KV_DMA_PER_STAGE = (WAVE_IMAGE ? NPI : 4 * PPW) + 2;   // LDS-DMA instructions a wave issues per stage (+ two statistics pieces)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int G = p.n_q_heads / p.n_kv_heads;
  const int kt_per_chunk = p.chunk_len / KT_KV;
  int bid = blockIdx.x;
  const int kvh = bid % p.n_kv_heads; bid /= p.n_kv_heads;
  const int kc = bid / kt_per_chunk;                    // kv chunk (buffer order)
  const int kti = bid % kt_per_chunk;
  const int gk = p.kv_gid[kc];
  const int k_off = kti * KT_KV;                        // first key of this workgroup inside its chunk
  const int my_key = k_off + wave * 32 + l31;           // this lane's key (column of S)
  const int64_t k_row0 = p.kv_row[kc] + k_off;

  // K / V fragments of this wave's 32 keys (MFMA B operand: lane -> key l31, k slot 2 ds + hi), straight from global memory
  bf16x8 kfr[NDS], vfr[NDS];
  {
    const bf16_t* kb = p.k + (int64_t)kvh * p.k_hs + (k_row0 + wave * 32 + l31) * p.k_rs + hi * 8;
    const bf16_t* vb = p.v + (int64_t)kvh * p.v_hs + (k_row0 + wave * 32 + l31) * p.v_rs + hi * 8;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      kfr[ds] = *reinterpret_cast<const bf16x8*>(kb + ds * 16);
      vfr[ds] = *reinterpret_cast<const bf16x8*>(vb + ds * 16);
    }
    // consume the loads here (see the dQ kernel): no compiler-placed vmcnt wait inside the loop
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) asm volatile("" :: "v"(kfr[ds]), "v"(vfr[ds]));
  }

  f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

  // ---- iteration space: (query head of the group, local query chunk, 32-row tile) ----------------
  // packed sequences: the last key of the workgroup bounds the query rows that can see any of its keys
  const int qt_per_chunk = p.seg_end ? (p.seg_end[k_row0 + KT_KV - 1] + QT_KV - 1) / QT_KV : p.chunk_len / QT_KV;
  auto first_tile = [&](int qc) __attribute__((always_inline)) -> int {   // first visible tile of chunk qc, or qt_per_chunk
    const int gq = p.q_gid[qc];
    if (gq > gk) return 0;
    if (gq < gk) return qt_per_chunk;
    return k_off / QT_KV;                               // rows >= first key of the workgroup
  };
  // HD = 96 (wave w stages image w): the lane's source offsets hold divisions by 12 — computed once
  unsigned pre_img[WAVE_IMAGE ? NPI : 1];
  if constexpr (WAVE_IMAGE) {
#pragma unroll
    for (int q = 0; q < NPI; ++q) pre_img[q] = dma_piece_voff<HD>(wave < 2 ? p.q_rs : p.do_rs, QT_KV, q, lane, (wave & 1) != 0);
  }
  auto stage_q = [&](int hq, int qc, int qt, unsigned sl) __attribute__((always_inline)) {
    const int head = kvh * G + hq;
    const int64_t row0 = (int64_t)qc * p.chunk_len + qt * QT_KV;
    const vita_rsrc_t qb = vita_make_rsrc(p.q + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + row0 * p.q_rs);
    const vita_rsrc_t db = vita_make_rsrc(p.d_o + (int64_t)head * p.do_hs + row0 * p.do_rs);
    if constexpr (WAVE_IMAGE) {                          // images in LDS order: Q frag, Q tr, dO frag, dO tr = waves 0 .. 3
      const vita_rsrc_t src = wave < 2 ? qb : db;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sl + wave * QT_KV * ROWB));
#pragma unroll
      for (int q = 0; q < NPI; ++q) vita_lds_dma16(src, pre_img[q], dst + q * 1024);
    } else {
      // 8 (HD = 64: 4) pieces per 32-row image: 4 waves x PPW
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int piece = wave * PPW + q;
        dma_piece<HD>(qb, p.q_rs, QT_KV, piece, lane, false, sl);
        dma_piece<HD>(qb, p.q_rs, QT_KV, piece, lane, true, sl + QT_KV * ROWB);
        dma_piece<HD>(db, p.do_rs, QT_KV, piece, lane, false, sl + 2 * QT_KV * ROWB);
        dma_piece<HD>(db, p.do_rs, QT_KV, piece, lane, true, sl + 3 * QT_KV * ROWB);
      }
    }
    // statistics of the 32 rows, by DMA as well (no register round trip, so nothing waits on it).  Every wave issues two pieces,
    // so that one counted vmcnt serves all waves: wave 0 the ones that are read (lse at +0, delta at +256), wave 1 the segment starts
    // of packed sequences (+512) and a spare, waves 2 / 3 spares
    const int64_t sidx = (int64_t)head * p.n_q_rows + row0;
    const unsigned stat = sl + 4 * QT_KV * ROWB;
    const void* first = (wave == 1 && p.seg_start) ? (const void*)(p.seg_start + row0) : (const void*)(p.lse + sidx);
    dma_words32(vita_make_rsrc(first), lane, stat + wave * 512);
    dma_words32(vita_make_rsrc(p.delta + sidx), lane, stat + wave * 512 + 256);
  };

  int hq_c = 0, qc_c = 0, qt_c = 0;
  auto normalize = [&](int& hq, int& qc, int& qt) __attribute__((always_inline)) {   // skip to a valid tile or hq == G
    while (hq < G) {
      while (qc < p.n_q_chunks) {
        if (qt < qt_per_chunk) return;
        ++qc;
        if (qc < p.n_q_chunks) qt = first_tile(qc);
      }
      ++hq; qc = 0; qt = first_tile(0);
    }
  };
  qt_c = first_tile(0);
  normalize(hq_c, qc_c, qt_c);
  // two iterators over the same step sequence: `f` = the step fetched next (three ahead), `c` = the step computed
  int hq_f = hq_c, qc_f = qc_c, qt_f = qt_c;
  int fstage = 0, in_flight = 0;                         // ring slot of the next fetch; steps issued and not yet consumed
  const unsigned st0 = lds0;
  auto fetch_next = [&]() __attribute__((always_inline)) {
    if (hq_f >= G) return;
    stage_q(hq_f, qc_f, qt_f, st0 + fstage * KV_STAGE);
    fstage = fstage + 1 == KV_NSTAGE ? 0 : fstage + 1;
    ++in_flight;
    ++qt_f;
    normalize(hq_f, qc_f, qt_f);
  };
  // wait until all but the `n` most recently issued steps have landed
  auto wait_keep = [&](int n) __attribute__((always_inline)) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * KV_DMA_PER_STAGE) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KV_DMA_PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  fetch_next(); fetch_next(); fetch_next();
  wait_keep(in_flight - 1);
  __syncthreads();

  const int krow = wave * 32 + l31;                     // this lane's key inside the workgroup's 128
  (void)krow;
  int stage = 0;
  while (hq_c < G) {
    fetch_next();                                        // step t+3 -> the slot step t-1 was read from

    const unsigned sl = st0 + stage * KV_STAGE;
    const unsigned qfr = sl, qtr = sl + QT_KV * ROWB, dofr = sl + 2 * QT_KV * ROWB, dotr = sl + 3 * QT_KV * ROWB;
    const unsigned stat = sl + 4 * QT_KV * ROWB;
    const int q_off = qt_c * QT_KV;                     // tile offset inside its chunk
    const bool diag = p.q_gid[qc_c] == gk;
    // rows of this tile are all before this wave's keys -> nothing visible
    const bool skip = diag && q_off + QT_KV - 1 < k_off + wave * 32;
    if (!skip) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        const bf16x8 qa = read_frag<HD>(qfr, l31, 2 * ds + hi);
        const bf16x8 da = read_frag<HD>(dofr, l31, 2 * ds + hi);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kfr[ds], s, 0, 0, 0);       // S[q, key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vfr[ds], dp, 0, 0, 0);     // dP[q, key]
      }
      const bool need_mask = diag && q_off < k_off + wave * 32 + 31;
      f32x16 pr;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 l4 = *(lds_f32x4*)(uintptr_t)(stat + (8 * rg + 4 * hi) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pr[rg * 4 + j] = __builtin_amdgcn_exp2f(fmaf(s[rg * 4 + j], p.scale_log2e, -l4[j] * 1.44269504088896340736f));
      }
      if (need_mask || p.seg_start) {                    // wave-uniform: only steps on the diagonal, or packed sequences
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          typedef __attribute__((ext_vector_type(4))) int i32x4;
          i32x4 st4 = {0, 0, 0, 0};
          if (p.seg_start) st4 = *(__attribute__((address_space(3))) const i32x4*)(uintptr_t)(stat + 512 + (8 * rg + 4 * hi) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int qrow = need_mask ? q_off + 8 * rg + 4 * hi + j : 0x7fffffff;
            pr[rg * 4 + j] = ((my_key > qrow) | (my_key < st4[j])) ? 0.f : pr[rg * 4 + j];
          }
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 d4 = *(lds_f32x4*)(uintptr_t)(stat + 256 + (8 * rg + 4 * hi) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[rg * 4 + j] = pr[rg * 4 + j] * (dp[rg * 4 + j] - d4[j]) * p.scale;      // dS[q, key]
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 pf = pack8(pr, 8 * t);
        bf16x8 dsf = pack8(s, 8 * t);
        // VALU result -> inline-asm MFMA operand: the compiler does not know the asm is an MFMA and inserts no wait states
        asm volatile("s_nop 4" : "+v"(pf), "+v"(dsf));
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const bf16x8 dot = read_tr<HD>(dotr, lane, 16 * t, db);                    // dO^T[d, q]
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dv_acc[db]) : "v"(dot), "v"(pf));
          const bf16x8 qtf = read_tr<HD>(qtr, lane, 16 * t, db);                     // Q^T[d, q]
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dk_acc[db]) : "v"(qtf), "v"(dsf));
        }
      }
    }
    // step t+1 must have landed; steps t+2 and t+3 may stay in flight
    --in_flight;
    wait_keep(in_flight - 1);
    __syncthreads();
    stage = stage + 1 == KV_NSTAGE ? 0 : stage + 1;
    ++qt_c;
    normalize(hq_c, qc_c, qt_c);
  }

  if constexpr (NDB == 4)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk_acc[0]), "+a"(dk_acc[1]), "+a"(dk_acc[2]), "+a"(dk_acc[3]), "+a"(dv_acc[0]),
                 "+a"(dv_acc[1]), "+a"(dv_acc[2]), "+a"(dv_acc[3]));
  else if constexpr (NDB == 3)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk_acc[0]), "+a"(dk_acc[1]), "+a"(dk_acc[2]), "+a"(dv_acc[0]), "+a"(dv_acc[1]), "+a"(dv_acc[2]));
  else asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk_acc[0]), "+a"(dk_acc[1]), "+a"(dv_acc[0]), "+a"(dv_acc[1]));
  const int64_t orow = k_row0 + wave * 32 + l31;
  bf16_t* okp = p.dk + orow * p.dk_rs + (int64_t)kvh * p.dk_hs;
  bf16_t* ovp = p.dv + orow * p.dv_rs + (int64_t)kvh * p.dv_hs;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 32 * db + 8 * rg + 4 * hi;
      u32x2 wk = {pack_bf16x2(dk_acc[db][rg * 4 + 0], dk_acc[db][rg * 4 + 1]),
                  pack_bf16x2(dk_acc[db][rg * 4 + 2], dk_acc[db][rg * 4 + 3])};
      u32x2 wv = {pack_bf16x2(dv_acc[db][rg * 4 + 0], dv_acc[db][rg * 4 + 1]),
                  pack_bf16x2(dv_acc[db][rg * 4 + 2], dv_acc[db][rg * 4 + 3])};
      *reinterpret_cast<u32x2*>(okp + d) = wk;
      *reinterpret_cast<u32x2*>(ovp + d) = wv;
    }
}

}  // namespace

extern "C" int vita_flash_attn_bwd_parts(const vita_attn_bwd_params* p, int parts, void* stream);

extern "C" int vita_flash_attn_bwd(const vita_attn_bwd_params* p, void* stream) {
  return vita_flash_attn_bwd_parts(p, VITA_ATTN_BWD_DQ | VITA_ATTN_BWD_DKV, stream);
}

// parts: VITA_ATTN_BWD_DKV (the dK + dV pass) and / or VITA_ATTN_BWD_DQ (the dQ pass); when both are asked for, dK / dV are launched
// FIRST so that a context-parallel caller can start the reduce-scatter of dK / dV while the dQ pass runs (cp_attn.hip)
extern "C" int vita_flash_attn_bwd_parts(const vita_attn_bwd_params* p, int parts, void* stream) {
  if (!(parts & (VITA_ATTN_BWD_DQ | VITA_ATTN_BWD_DKV))) return VITA_ERR_INVALID_ARG;
  if (!p || !p->q || !p->k || !p->v || !p->d_o || !p->lse || !p->delta) return VITA_ERR_INVALID_ARG;
  if (((parts & VITA_ATTN_BWD_DQ) && !p->dq) || ((parts & VITA_ATTN_BWD_DKV) && (!p->dk || !p->dv))) return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 128 && p->head_dim != 96 && p->head_dim != 64) return VITA_ERR_UNSUPPORTED;
  if (p->n_q_heads <= 0 || p->n_kv_heads <= 0 || p->n_q_heads % p->n_kv_heads) return VITA_ERR_INVALID_ARG;
  if (p->n_q_chunks <= 0 || p->n_kv_chunks <= 0 || p->n_q_chunks > kMaxChunks || p->n_kv_chunks > kMaxChunks)
    return VITA_ERR_UNSUPPORTED;
  if (p->chunk_len <= 0 || p->chunk_len % 128 || p->chunk_len > 0x7fffff00LL) return VITA_ERR_UNSUPPORTED;
  if (!p->q_chunk_gid || !p->kv_chunk_gid || !p->kv_chunk_row) return VITA_ERR_INVALID_ARG;
  const int64_t strides[] = {p->q_row_stride, p->q_head_stride, p->q_group_stride, p->k_row_stride,
                             p->k_head_stride, p->v_row_stride, p->v_head_stride, p->do_row_stride,
                             p->do_head_stride};
  for (int64_t s : strides)
    if (s & 7) return VITA_ERR_UNSUPPORTED;
  BwdArgs a;
  const int G = p->n_q_heads / p->n_kv_heads;
  a.q = (const bf16_t*)p->q; a.q_rs = p->q_row_stride; a.q_hs = p->q_head_stride;
  a.q_gs = p->q_group_stride ? p->q_group_stride : p->q_head_stride * G;
  a.k = (const bf16_t*)p->k; a.k_rs = p->k_row_stride; a.k_hs = p->k_head_stride;
  a.v = (const bf16_t*)p->v; a.v_rs = p->v_row_stride; a.v_hs = p->v_head_stride;
  a.d_o = (const bf16_t*)p->d_o; a.do_rs = p->do_row_stride; a.do_hs = p->do_head_stride;
  a.lse = p->lse; a.delta = p->delta;
  a.dq = (bf16_t*)p->dq; a.dq_rs = p->dq_row_stride; a.dq_hs = p->dq_head_stride;
  a.dq_gs = p->dq_group_stride ? p->dq_group_stride : p->dq_head_stride * G;
  a.dk = (bf16_t*)p->dk; a.dk_rs = p->dk_row_stride; a.dk_hs = p->dk_head_stride;
  a.dv = (bf16_t*)p->dv; a.dv_rs = p->dv_row_stride; a.dv_hs = p->dv_head_stride;
  a.n_q_heads = p->n_q_heads; a.n_kv_heads = p->n_kv_heads; a.head_dim = p->head_dim;
  a.chunk_len = (int)p->chunk_len; a.n_q_chunks = p->n_q_chunks; a.n_kv_chunks = p->n_kv_chunks;
  a.n_q_rows = (int)(p->n_q_chunks * p->chunk_len);
  a.scale = p->softmax_scale; a.scale_log2e = p->softmax_scale * 1.44269504088896340736f;
  a.seg_start = p->q_seg_start; a.seg_end = p->k_seg_end;
  if ((p->q_seg_start != nullptr) != (p->k_seg_end != nullptr)) return VITA_ERR_INVALID_ARG;
  if (p->q_seg_start && (p->n_q_chunks != 1 || p->n_kv_chunks != 1)) return VITA_ERR_UNSUPPORTED;
  for (int i = 0; i < p->n_q_chunks; ++i) a.q_gid[i] = p->q_chunk_gid[i];
  for (int i = 0; i < p->n_kv_chunks; ++i) { a.kv_gid[i] = p->kv_chunk_gid[i]; a.kv_row[i] = p->kv_chunk_row[i]; }

  hipStream_t st = (hipStream_t)stream;
  static std::atomic<unsigned long long> attr_set{0};
  constexpr int lds_dq = DQ_NSTAGE * 3 * KT_DQ * 256, lds_kv = KV_NSTAGE * (4 * QT_KV * 256 + 2048);        // the HD = 128 sizes (HD = 64 needs less)
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
  });
  const int64_t n_dq = (int64_t)p->n_q_heads * p->n_q_chunks * (p->chunk_len / QT_DQ);
  const int64_t n_kv = (int64_t)p->n_kv_heads * p->n_kv_chunks * (p->chunk_len / KT_KV);
  if (n_dq > 0x7fffffff || n_kv > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  const char* only = vita_dev_getenv("VITA_ATTN_BWD_ONLY");     // developer measurement aid: "dq" / "dkv" launch one of the two kernels
  if ((parts & VITA_ATTN_BWD_DKV) && (!only || only[1] == 'k')) {
    if (vita_attn_bwd_kv64_eligible(a) && vita_attn_bwd_kvp_eligible(a)) {      // r04: both gradients in one launch, S computed once
      const int rc = vita_attn_bwd_kvp_launch(a, st);
      if (rc != VITA_OK) return rc;
    } else if (vita_attn_bwd_kv64_eligible(a)) {            // whole 256-key tiles: 64 keys per wave, a dK and a dV launch (attn_bwd_kv64.hip)
      const int rc = vita_attn_bwd_kv64_launch(a, st);
      if (rc != VITA_OK) return rc;
    } else if (p->head_dim == 64) {
      hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, dim3((unsigned)n_kv), dim3(256), lds_kv, st, a);
    } else if (p->head_dim == 96) {
      hipLaunchKernelGGL(attn_bwd_dkv_kernel<96>, dim3((unsigned)n_kv), dim3(256), lds_kv, st, a);
    } else {
      hipLaunchKernelGGL(attn_bwd_dkv_kernel<128>, dim3((unsigned)n_kv), dim3(256), lds_kv, st, a);
    }
  }
  if ((parts & VITA_ATTN_BWD_DQ) && (!only || only[1] == 'q')) {
    if (vita_attn_bwd_dq64_eligible(a)) {            // causal whole 256-row tiles: 64 query rows per wave (attn_bwd64.hip)
      const int rc = vita_attn_bwd_dq64_launch(a, st);
      if (rc != VITA_OK) return rc;
    } else if (p->head_dim == 64) {
      hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, dim3((unsigned)n_dq), dim3(256), lds_dq, st, a);
    } else if (p->head_dim == 96) {
      hipLaunchKernelGGL(attn_bwd_dq_kernel<96>, dim3((unsigned)n_dq), dim3(256), lds_dq, st, a);
    } else {
      hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, dim3((unsigned)n_dq), dim3(256), lds_dq, st, a);
    }
  }
  return vita_check_launch();
}
