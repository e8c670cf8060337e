// Shared between attn_bwd.hip (the general backward kernels and the C entry point) and attn_bwd64.hip (the 64-rows-per-wave dQ kernel).
#pragma once
#include "vita_common.h"

constexpr int kBwdMaxChunks = 32;

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
  int n_q_heads, n_kv_heads, head_dim;
  int chunk_len, n_q_chunks, n_kv_chunks, n_q_rows;
  float scale, scale_log2e;
  const int* seg_start;   // packed sequences (single chunk): first row of each query row's segment, or null
  const int* seg_end;     // one past the last row of each key row's segment, or null
  int q_gid[kBwdMaxChunks];
  int kv_gid[kBwdMaxChunks];
  int64_t kv_row[kBwdMaxChunks];
};

// attn_bwd64.hip: dQ with 64 query rows per wave (causal, whole 256-row tiles, every query chunk also a key chunk)
bool vita_attn_bwd_dq64_eligible(const BwdArgs& a);
int vita_attn_bwd_dq64_launch(const BwdArgs& a, hipStream_t st);
// attn_bwd_kv64.hip: dK and dV with 64 keys per wave, as two launches (causal, whole 256-key tiles)
bool vita_attn_bwd_kv64_eligible(const BwdArgs& a);
int vita_attn_bwd_kv64_launch(const BwdArgs& a, hipStream_t st);
// attn_bwd_kvp.hip (r04): dK AND dV in ONE launch — wave pairs share 64 keys, one wave computes S / P / dV, its partner dP / dS / dK (4 GEMM
// units instead of kv64's 5); causal, whole 128-key tiles, no packed samples
bool vita_attn_bwd_kvp_eligible(const BwdArgs& a);
int vita_attn_bwd_kvp_launch(const BwdArgs& a, hipStream_t st);
