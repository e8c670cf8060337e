// Launch arguments of the flash-attention forward kernels (attn.hip: 8 waves x 32 rows, every geometry; attn64.hip: 4 waves x
// 64 rows, d = 128 causal with whole 256-row / 64-key tiles).
#pragma once
#include "vita_common.h"

constexpr int kMaxChunks = 32;

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
  const int* seg_start;                  // packed sequences: first key row of each query row's segment (or null)
  int q_order[kMaxChunks];               // q chunks sorted by gid descending
  int q_gid[kMaxChunks];
  int kv_gid[kMaxChunks];
  int64_t kv_row[kMaxChunks];
};

// attn64.hip: true when the geometry qualifies (head_dim 128, causal, no packed segments, chunk_len % 256 == 0, whole chunks)
bool vita_attn64_eligible(const AttnArgs& a, int head_dim, bool causal);
int vita_attn64_launch(const AttnArgs& a, int64_t nblocks, hipStream_t st);
// attn64v.hip (r05): head_dim 64, non-causal, one chunk, ragged rows / keys, any batch — the vision towers' attention
bool vita_attn64v_eligible(const AttnArgs& a, int head_dim, bool causal);
int vita_attn64v_launch(const AttnArgs& a, hipStream_t st);
