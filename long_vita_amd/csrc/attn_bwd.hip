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
#include "vita_common.h"

namespace {

constexpr int kMaxChunks = 32;
constexpr int D = 128, ROWB = 256;     // bytes per row

struct BwdArgs {
  const bf16_t* q; int64_t q_rs, q_hs, q_gs;        // query rows (rotated), grouped head addressing
  const bf16_t* k; int64_t k_rs, k_hs;
  const bf16_t* v; int64_t v_rs, v_hs;
  const bf16_t* d_o; int64_t do_rs, do_hs;           // [rows, Hq, 128]
  const float* lse;                                   // [Hq, n_q_rows] natural log
  const float* delta;                                 // [Hq, n_q_rows]
  bf16_t* dq; int64_t dq_rs, dq_hs, dq_gs;
  bf16_t* dk; int64_t dk_rs, dk_hs;                   // same row space as k / v
  bf16_t* dv; int64_t dv_rs, dv_hs;
  int n_q_heads, n_kv_heads;
  int chunk_len, n_q_chunks, n_kv_chunks, n_q_rows;
  float scale, scale_log2e;
  const int* seg_start;   // packed sequences (single chunk): first row of each query row's segment, or null
  const int* seg_end;     // one past the last row of each key row's segment, or null
  int q_gid[kMaxChunks];
  int kv_gid[kMaxChunks];
  int64_t kv_row[kMaxChunks];
};

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ int frag_off(int row, int slot) { return row * ROWB + ((slot ^ (row & 15)) << 4); }
__device__ __forceinline__ int tr_off(int row, int chunk, int b) { return row * ROWB + ((chunk ^ ((row & 3) << 1)) << 5) + b; }

// DMA one 1-KiB piece (4 rows x 256 B) of a [rows][128] bf16 matrix into LDS; lane -> (row piece*4 +
// lane/16, physical 16-B slot lane%16); `tr` selects the layout (source-side swizzle).
__device__ __forceinline__ void dma_piece(const bf16_t* base, int64_t rs, int row_lo, int row_hi, int piece,
                                          int lane, bool tr, unsigned lds_dst) {
  int row = piece * 4 + (lane >> 4);
  const int ps = lane & 15;
  const int ls = tr ? ((((ps >> 1) ^ ((row & 3) << 1)) << 1) | (ps & 1)) : (ps ^ (row & 15));
  int grow = row_lo + row;
  grow = grow < row_hi ? grow : row_hi - 1;            // clamp (masked later)
  __builtin_amdgcn_global_load_lds((gvoid*)(base + (int64_t)grow * rs + ls * 8),
                                   (lvoid*)(uintptr_t)(lds_dst + piece * 1024), 16, 0, 0);
}

__device__ __forceinline__ bf16x8 read_frag(unsigned tile, int row, int slot) {
  return *(lds_bf16x8*)(uintptr_t)(tile + frag_off(row, slot));
}
// transposed fragment: rows r0 + {0..3} and r0 + 8 + {0..3} (r0 already includes 4*(lane>>5) and the
// lane's row inside its 16-lane group), 32 columns starting at 32*db
__device__ __forceinline__ bf16x8 read_tr(unsigned tile, int lane, int row_base, int db) {
  const int g16 = lane >> 4, i16 = lane & 15;
  const int row = row_base + 4 * (g16 >> 1) + (i16 >> 2);
  const int col = 32 * db + 16 * (g16 & 1) + 4 * (i16 & 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(tile + tr_off(row, col >> 4, (col & 15) * 2)));
  const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(tile + tr_off(row + 8, col >> 4, (col & 15) * 2)));
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
// LDS per stage: K frag (16 KiB) | K tr (16 KiB) | V frag (16 KiB)
constexpr int DQ_STAGE = 3 * KT_DQ * ROWB;

__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(BwdArgs p) {
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

  bf16x8 qf[8], dof[8];
  {
    const bf16_t* qp = p.q + q_row * p.q_rs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
    const bf16_t* dp = p.d_o + q_row * p.do_rs + (int64_t)head * p.do_hs + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
      dof[ds] = *reinterpret_cast<const bf16x8*>(dp + ds * 16);
    }
  }
  const float lse2 = p.lse[(int64_t)head * p.n_q_rows + q_row] * 1.44269504088896340736f;
  const float dlt = p.delta[(int64_t)head * p.n_q_rows + q_row];

  f32x16 dq_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
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
  auto stage_tile = [&](int c, int j, unsigned sl) __attribute__((always_inline)) {
    const int64_t crow = p.kv_row[c];
    const int lo = j * KT_DQ;
    // 16 pieces per 64-row image; 4 waves -> 4 pieces each per image
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = wave * 4 + q;
      dma_piece(kbase + crow * p.k_rs, p.k_rs, lo, p.chunk_len, piece, lane, false, sl);
      dma_piece(kbase + crow * p.k_rs, p.k_rs, lo, p.chunk_len, piece, lane, true, sl + KT_DQ * ROWB);
      dma_piece(vbase + crow * p.v_rs, p.v_rs, lo, p.chunk_len, piece, lane, false, sl + 2 * KT_DQ * ROWB);
    }
  };

  int c_cur = 0, j_cur = wg_first_start / KT_DQ, n_cur = 0;
  while (c_cur < p.n_kv_chunks && (n_cur = chunk_tiles(c_cur)) == 0) ++c_cur;
  if (c_cur < p.n_kv_chunks) stage_tile(c_cur, j_cur, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int stage = 0;
  while (c_cur < p.n_kv_chunks) {
    int c_n = c_cur, j_n = j_cur + 1, n_n = n_cur;
    if (j_n == n_cur) {
      j_n = 0;
      ++c_n;
      while (c_n < p.n_kv_chunks && (n_n = chunk_tiles(c_n)) == 0) ++c_n;
    }
    if (c_n < p.n_kv_chunks) stage_tile(c_n, j_n, lds0 + (stage ^ 1) * DQ_STAGE);

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
        for (int ds = 0; ds < 8; ++ds) {
          const bf16x8 ka = read_frag(kf, 32 * h + l31, 2 * ds + hi);
          const bf16x8 va = read_frag(vf, 32 * h + l31, 2 * ds + hi);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ds], s, 0, 0, 0);       // S^T[key, q]
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ds], dp, 0, 0, 0);    // dP^T[key, q]
        }
        const bool need_mask = diag && kv_off + 32 * h + 31 > q_off;
        const bool seg_mask = kv_off + 32 * h < wg_last_start;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv_off + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * hi;
          float pr = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -lse2));
          if ((need_mask && key > my_q) || (seg_mask && key < my_start)) pr = 0.f;
          s[r] = pr * (dp[r] - dlt) * p.scale;           // dS^T
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 dsf = pack8(s, 8 * t);
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const bf16x8 ktf = read_tr(kt, lane, 32 * h + 16 * t, db);               // K^T[d, keys]
            dq_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq_acc[db], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage ^= 1;
    c_cur = c_n; j_cur = j_n; n_cur = n_n;
  }

  bf16_t* op = p.dq + q_row * p.dq_rs + (int64_t)kvh * p.dq_gs + (int64_t)hq * p.dq_hs;
#pragma unroll
  for (int db = 0; db < 4; ++db)
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
// LDS: K frag (32 KiB) | V frag (32 KiB) | 2 stages x [Q frag 8 | Q tr 8 | dO frag 8 | dO tr 8 | lse 128 B | delta 128 B]
constexpr int KV_FIXED = 2 * KT_KV * ROWB;
constexpr int KV_STAGE = 4 * QT_KV * ROWB + 384;   // + lse, delta, segment start of the 32 rows

__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(BwdArgs p) {
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

  // K / V of this workgroup: frag layout, once
  {
    const bf16_t* kb = p.k + (int64_t)kvh * p.k_hs + k_row0 * p.k_rs;
    const bf16_t* vb = p.v + (int64_t)kvh * p.v_hs + k_row0 * p.v_rs;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                        // 32 pieces per 128-row image, 8 per wave
      const int piece = wave * 8 + q;
      dma_piece(kb, p.k_rs, 0, KT_KV, piece, lane, false, lds0);
      dma_piece(vb, p.v_rs, 0, KT_KV, piece, lane, false, lds0 + KT_KV * ROWB);
    }
  }

  f32x16 dk_acc[4], dv_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
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
  auto stage_q = [&](int hq, int qc, int qt, unsigned sl) __attribute__((always_inline)) {
    const int head = kvh * G + hq;
    const int64_t row0 = (int64_t)qc * p.chunk_len + qt * QT_KV;
    const bf16_t* qb = p.q + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + row0 * p.q_rs;
    const bf16_t* db = p.d_o + (int64_t)head * p.do_hs + row0 * p.do_rs;
    // 8 pieces per 32-row image: 4 waves x 2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int piece = wave * 2 + q;
      dma_piece(qb, p.q_rs, 0, QT_KV, piece, lane, false, sl);
      dma_piece(qb, p.q_rs, 0, QT_KV, piece, lane, true, sl + QT_KV * ROWB);
      dma_piece(db, p.do_rs, 0, QT_KV, piece, lane, false, sl + 2 * QT_KV * ROWB);
      dma_piece(db, p.do_rs, 0, QT_KV, piece, lane, true, sl + 3 * QT_KV * ROWB);
    }
    if (wave == 0) {                                     // lse (x log2e) and delta of the 32 rows
      const int64_t sidx = (int64_t)head * p.n_q_rows + row0 + l31;
      const float val = hi == 0 ? p.lse[sidx] * 1.44269504088896340736f : p.delta[sidx];
      *(__attribute__((address_space(3))) float*)(uintptr_t)(sl + 4 * QT_KV * ROWB + hi * 128 + l31 * 4) = val;
    }
    if (wave == 1 && hi == 0)
      *(__attribute__((address_space(3))) int*)(uintptr_t)(sl + 4 * QT_KV * ROWB + 256 + l31 * 4) =
          p.seg_start ? p.seg_start[row0 + l31] : 0;
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
  const unsigned st0 = lds0 + KV_FIXED;
  if (hq_c < G) stage_q(hq_c, qc_c, qt_c, st0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned kfrag = lds0, vfrag = lds0 + KT_KV * ROWB;
  const int krow = wave * 32 + l31;                     // this lane's key row inside the K/V images
  int stage = 0;
  while (hq_c < G) {
    int hq_n = hq_c, qc_n = qc_c, qt_n = qt_c + 1;
    normalize(hq_n, qc_n, qt_n);
    if (hq_n < G) stage_q(hq_n, qc_n, qt_n, st0 + (stage ^ 1) * KV_STAGE);

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
      for (int ds = 0; ds < 8; ++ds) {
        const bf16x8 qa = read_frag(qfr, l31, 2 * ds + hi);
        const bf16x8 kb = read_frag(kfrag, krow, 2 * ds + hi);
        const bf16x8 da = read_frag(dofr, l31, 2 * ds + hi);
        const bf16x8 vb = read_frag(vfrag, krow, 2 * ds + hi);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kb, s, 0, 0, 0);            // S[q, key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vb, dp, 0, 0, 0);          // dP[q, key]
      }
      const bool need_mask = diag && q_off < k_off + wave * 32 + 31;
      f32x16 pr;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 l4 = *(lds_f32x4*)(uintptr_t)(stat + (8 * rg + 4 * hi) * 4);
        const f32x4 d4 = *(lds_f32x4*)(uintptr_t)(stat + 128 + (8 * rg + 4 * hi) * 4);
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 st4 = *(__attribute__((address_space(3))) const i32x4*)(uintptr_t)(stat + 256 + (8 * rg + 4 * hi) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = rg * 4 + j;
          const int qrow = q_off + 8 * rg + 4 * hi + j;
          float e = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -l4[j]));
          if ((need_mask && my_key > qrow) || my_key < st4[j]) e = 0.f;
          pr[r] = e;
          s[r] = e * (dp[r] - d4[j]) * p.scale;          // dS[q, key]
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 pf = pack8(pr, 8 * t);
        const bf16x8 dsf = pack8(s, 8 * t);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 dot = read_tr(dotr, lane, 16 * t, db);                        // dO^T[d, q]
          dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dv_acc[db], 0, 0, 0);
          const bf16x8 qtf = read_tr(qtr, lane, 16 * t, db);                         // Q^T[d, q]
          dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk_acc[db], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    stage ^= 1;
    hq_c = hq_n; qc_c = qc_n; qt_c = qt_n;
  }

  const int64_t orow = k_row0 + wave * 32 + l31;
  bf16_t* okp = p.dk + orow * p.dk_rs + (int64_t)kvh * p.dk_hs;
  bf16_t* ovp = p.dv + orow * p.dv_rs + (int64_t)kvh * p.dv_hs;
#pragma unroll
  for (int db = 0; db < 4; ++db)
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

extern "C" int vita_flash_attn_bwd(const vita_attn_bwd_params* p, void* stream) {
  if (!p || !p->q || !p->k || !p->v || !p->d_o || !p->lse || !p->delta || !p->dq || !p->dk || !p->dv)
    return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 128) return VITA_ERR_UNSUPPORTED;
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
  a.n_q_heads = p->n_q_heads; a.n_kv_heads = p->n_kv_heads;
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
  constexpr int lds_dq = 2 * DQ_STAGE, lds_kv = KV_FIXED + 2 * KV_STAGE;
  vita_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
  });
  const int64_t n_dq = (int64_t)p->n_q_heads * p->n_q_chunks * (p->chunk_len / QT_DQ);
  const int64_t n_kv = (int64_t)p->n_kv_heads * p->n_kv_chunks * (p->chunk_len / KT_KV);
  if (n_dq > 0x7fffffff || n_kv > 0x7fffffff) return VITA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)n_dq), dim3(256), lds_dq, st, a);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)n_kv), dim3(256), lds_kv, st, a);
  return vita_check_launch();
}
