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
#include "vita_common.h"

namespace {

constexpr int kMaxChunks = 32;
constexpr int QTILE = 256;   // query rows per workgroup (8 waves x 32)
constexpr int KVT = 64;      // keys per tile

struct AttnArgs {
  const bf16_t* q; int64_t q_bs, q_rs, q_hs, q_gs;   // q_gs: stride between kv groups' first query head
  const bf16_t* k; int64_t k_bs, k_rs, k_hs;
  const bf16_t* v; int64_t v_bs, v_rs, v_hs;
  bf16_t* o; int64_t o_bs, o_rs, o_hs, o_gs;
  float* lse;
  int batch, n_q_heads, n_kv_heads;
  int chunk_len, q_valid, kv_valid;      // rows; *_valid apply to the last chunk
  int n_q_chunks, n_kv_chunks;
  int tiles_per_q_chunk;                 // ceil(chunk_len / 256)
  int n_q_rows;                          // total local q rows (for lse indexing)
  float scale_log2e;                     // softmax_scale * log2(e)
  int q_order[kMaxChunks];               // q chunks sorted by gid descending
  int q_gid[kMaxChunks];
  int kv_gid[kMaxChunks];
  int64_t kv_row[kMaxChunks];
};

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
  else return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}
// LDS byte offset of V (row, 32-byte chunk c, byte b within chunk)
template <int D>
__device__ __forceinline__ int v_lds_off(int row, int chunk, int b) {
  if (D == 128) return row * 256 + ((chunk ^ ((row & 3) << 1)) << 5) + b;
  else return row * 128 + ((chunk ^ (row & 2)) << 5) + b;
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void flash_fwd_kernel(AttnArgs p) {
  constexpr int DS = D / 16;              // QK^T k-steps
  constexpr int DB = D / 32;              // O^T row blocks
  constexpr int ROWB = D * 2;             // bytes per K/V row
  constexpr int TILEB = KVT * ROWB;       // bytes per K (or V) tile
  constexpr int SLOTS = ROWB / 16;        // 16-byte slots per row
  constexpr int LD_PER_THR = (KVT * SLOTS) / 512;  // 16-byte loads per thread per operand

  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K tile | V tile]

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

  // ---- Q fragments (B operand of S^T = K Q^T): lane = (query row l31, k-slot half hi) --------
  bf16x8 qf[DS];
  {
    const bf16_t* qp = p.q + (int64_t)b * p.q_bs + q_local_row * p.q_rs + (int64_t)kvh * p.q_gs + (int64_t)hq * p.q_hs + hi * 8;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  f32x16 o_acc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const bf16_t* kbase = p.k + (int64_t)b * p.k_bs + (int64_t)kvh * p.k_hs;
  const bf16_t* vbase = p.v + (int64_t)b * p.v_bs + (int64_t)kvh * p.v_hs;

  // number of tiles of kv chunk c this workgroup must visit
  auto chunk_tiles = [&](int c) -> int {
    const int rows = (c == p.n_kv_chunks - 1) ? p.kv_valid : p.chunk_len;
    const int all = (rows + KVT - 1) / KVT;
    if (!CAUSAL) return all;
    const int gk = p.kv_gid[c];
    if (gk < gq) return all;
    if (gk > gq) return 0;
    return min(all, q_last_wg / KVT + 1);
  };

  // staging registers for the next tile
  u32x4 kreg[LD_PER_THR], vreg[LD_PER_THR];
  auto issue_loads = [&](int c, int j) {
    const int rows = (c == p.n_kv_chunks - 1) ? p.kv_valid : p.chunk_len;
    const int64_t row0 = p.kv_row[c] + (int64_t)j * KVT;
#pragma unroll
    for (int it = 0; it < LD_PER_THR; ++it) {
      const int e = tid + it * 512;
      const int row = e / SLOTS, slot = e % SLOTS;
      int rr = j * KVT + row;
      rr = rr < rows ? rr : rows - 1;               // clamp padded tail rows (masked later)
      const int64_t grow = row0 + (rr - j * KVT);
      kreg[it] = *reinterpret_cast<const u32x4*>(kbase + grow * p.k_rs + slot * 8);
      vreg[it] = *reinterpret_cast<const u32x4*>(vbase + grow * p.v_rs + slot * 8);
    }
  };
  auto write_lds = [&](char* stage) {
#pragma unroll
    for (int it = 0; it < LD_PER_THR; ++it) {
      const int e = tid + it * 512;
      const int row = e / SLOTS, slot = e % SLOTS;
      *reinterpret_cast<u32x4*>(stage + k_lds_off<D>(row, slot)) = kreg[it];
      *reinterpret_cast<u32x4*>(stage + TILEB + v_lds_off<D>(row, slot >> 1, (slot & 1) << 4)) = vreg[it];
    }
  };

  // ---- tile iterator over (chunk, tile) pairs -------------------------------------------------
  int c_cur = 0, j_cur = 0, n_cur = 0;
  while (c_cur < p.n_kv_chunks && (n_cur = chunk_tiles(c_cur)) == 0) ++c_cur;
  const bool any = c_cur < p.n_kv_chunks;
  if (any) {
    issue_loads(c_cur, 0);
    write_lds(smem);
  }
  __syncthreads();

  int stage = 0;
  while (c_cur < p.n_kv_chunks) {
    // next tile
    int c_nxt = c_cur, j_nxt = j_cur + 1, n_nxt = n_cur;
    if (j_nxt == n_cur) {
      j_nxt = 0;
      ++c_nxt;
      while (c_nxt < p.n_kv_chunks && (n_nxt = chunk_tiles(c_nxt)) == 0) ++c_nxt;
    }
    const bool has_next = c_nxt < p.n_kv_chunks;
    if (has_next) issue_loads(c_nxt, j_nxt);

    const char* kt = smem + stage * (2 * TILEB);
    const char* vt = kt + TILEB;
    const int kv_off = j_cur * KVT;                  // tile offset inside its chunk
    const bool diag = CAUSAL && p.kv_gid[c_cur] == gq;
    const int kv_rows = (c_cur == p.n_kv_chunks - 1) ? p.kv_valid : p.chunk_len;
    const bool tail = kv_off + KVT > kv_rows;
    // wave-uniform skip: the whole tile lies after this wave's last query row
    const bool skip = diag && kv_off > q_off + 31;

    if (!skip) {
      // ---- S^T = K Q^T : two 32(key) x 32(query) blocks --------------------------------------
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kt + k_lds_off<D>(l31, 2 * ds + hi));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kt + k_lds_off<D>(32 + l31, 2 * ds + hi));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ds], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ds], s1, 0, 0, 0);
      }
      // key index (inside the tile) of accumulator register r: (r&3) + 8*(r>>2) + 4*hi (+32 for s1)
      if (diag || tail) {
        const int lim_c = diag ? (my_q - kv_off) : 0x7fffffff;        // key <= lim_c visible
        const int lim_t = kv_rows - kv_off - 1;                        // key <= lim_t valid
        const int lim = min(lim_c, lim_t);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key > lim) s0[r] = -INFINITY;
          if (key + 32 > lim) s1[r] = -INFINITY;
        }
      }
      // ---- online softmax (log2 domain) --------------------------------------------------------
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
      mx = swap32_max(mx);
      const float m_new = fmaxf(m_run, mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], p.scale_log2e, -m_new));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], p.scale_log2e, -m_new));
        psum += s0[r] + s1[r];
      }
      l_run = l_run * alpha + psum;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
      }
      // ---- P^T operand: step t uses registers 8*(t&1).. of block t>>1 ---------------------------
      bf16x8 pf[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pf[0][j] = (__bf16)s0[j];
        pf[1][j] = (__bf16)s0[8 + j];
        pf[2][j] = (__bf16)s1[j];
        pf[3][j] = (__bf16)s1[8 + j];
      }
      // ---- O^T += V^T P^T ------------------------------------------------------------------------
      // V^T A-operand for step t, row block db: lane (d = 32*db + l31, k-half hi) needs keys
      // 16t + 4hi + {0..3} and 16t + 8 + 4hi + {0..3}: two transpose reads.  Inside a 16-lane
      // group, lane i supplies the 8-byte piece (row i>>2, columns 4*(i&3)..) of a 4x16 block.
      const int g16 = lane >> 4, i16 = lane & 15;
      const int v_key_l = 4 * (g16 >> 1) + (i16 >> 2);          // + 16 t (+8)
      const int v_col_l = 16 * (g16 & 1) + 4 * (i16 & 3);       // + 32 db
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int col = 32 * db + v_col_l;
          const int r0 = 16 * t + v_key_l, r1 = r0 + 8;
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vt + v_lds_off<D>(r0, col >> 4, (col & 15) * 2)));
          const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vt + v_lds_off<D>(r1, col >> 4, (col & 15) * 2)));
          union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
          u.s.lo = a; u.s.hi = c;
          o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, pf[t], o_acc[db], 0, 0, 0);
        }
      }
    }

    if (has_next) write_lds(smem + (stage ^ 1) * (2 * TILEB));
    __syncthreads();
    stage ^= 1;
    c_cur = c_nxt; j_cur = j_nxt; n_cur = n_nxt;
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

template <int D, bool CAUSAL>
int launch_attn(const AttnArgs& a, int64_t nblocks, hipStream_t st) {
  constexpr int lds = 2 * 2 * KVT * D * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_kernel<D, CAUSAL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((flash_fwd_kernel<D, CAUSAL>), dim3((unsigned)nblocks), dim3(512), lds, st, a);
  return vita_check_launch();
}

}  // namespace

extern "C" int vita_flash_attn_fwd(const vita_attn_params* p, void* stream) {
  if (!p || !p->q || !p->k || !p->v || !p->o) return VITA_ERR_INVALID_ARG;
  if (p->batch <= 0 || p->n_q_heads <= 0 || p->n_kv_heads <= 0 || p->chunk_len <= 0 ||
      p->n_q_chunks <= 0 || p->n_kv_chunks <= 0 || !p->q_chunk_gid || !p->kv_chunk_gid ||
      !p->kv_chunk_row)
    return VITA_ERR_INVALID_ARG;
  if (p->head_dim != 64 && p->head_dim != 128) return VITA_ERR_UNSUPPORTED;
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
  if (p->head_dim == 128) return p->causal ? launch_attn<128, true>(a, nblocks, st) : launch_attn<128, false>(a, nblocks, st);
  return p->causal ? launch_attn<64, true>(a, nblocks, st) : launch_attn<64, false>(a, nblocks, st);
}
