/*
 * vita_hip.h — C ABI of libvita_hip.so, the MI355X (gfx950) hot path of Long-VITA prefill.
 *
 * The reference (VITA-MLLM/Long-VITA) contains no native code: every kernel on this path is
 * reached from Python through a third-party package (flash_attn, transformer_engine, apex,
 * cuBLAS via torch.matmul).  Each entry point below therefore cites the *Python call site* of
 * the reference that it replaces (R/ = /root/reference, M/ = R/long_vita_megatron,
 * H/ = R/long_vita).  A reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless noted;
 *   - activations / weights are bf16 (uint16 bit pattern), indices int64, statistics fp32;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); nothing is
 *     allocated, retained or freed by the library;
 *   - return value: VITA_OK (0) or a negative VITA_ERR_* code; no C++ exception crosses the ABI.
 */
#ifndef VITA_HIP_H
#define VITA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITA_OK 0
#define VITA_ERR_INVALID_ARG (-1)   /* null pointer, non-positive size, bad enum                */
#define VITA_ERR_UNSUPPORTED (-2)   /* shape outside what the gfx950 kernels are built for      */
#define VITA_ERR_LAUNCH (-3)        /* hipGetLastError() != hipSuccess after the launch          */

/* ABI version; bump on any signature change. */
int vita_abi_version(void);
/* Human-readable text for a VITA_ERR_* code (static storage). */
const char* vita_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * RMSNorm forward.   y = (x.float() * rsqrt(mean(x^2) + eps)).to(bf16) * w        [rows, cols]
 * Replaces M/core/transformer/custom_layers/transformer_engine.py:74-79 (RMSNorm._norm/forward;
 * the TE-fused equivalent on GPU, M/core/models/gpt/gpt_layer_specs.py:39).
 * rstd_out (fp32 [rows]) may be NULL; when given it holds rsqrt(mean(x^2)+eps) per row.
 * ------------------------------------------------------------------------------------------- */
int vita_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd_out,
                     int64_t rows, int cols, float eps, void* stream);

/* LayerNorm forward (biased variance, fp32 statistics), bf16 in/out.
 * Replaces torch.nn.LayerNorm at M/core/models/vision/vit_layer_specs.py (ViT block norms,
 * eps 1e-6) and M/pretrain_long_vita.py:443-446 (projector pre-norm, eps 1e-5). */
int vita_layernorm_fwd(const void* x, const void* w, const void* b, void* y,
                       int64_t rows, int cols, float eps, void* stream);

/* Logit post-processing, in place on the bf16 logits [rows, cols] of the selected rows (row stride ld elements):
 *   logits *= multiplier_scale (skipped when 0);  logits = tanh(logits / softcapping) * softcapping (skipped when 0)
 * with the reference's bf16 rounding after every step.  Replaces M/core/models/multimodal/gpt_vl_model.py:349-355
 * (args.output_multiplier_scale, args.output_logit_softcapping).  _bwd: grad *= scale * (1 - (y / cap)^2) from the
 * stored output y (autograd of the same expressions). */
int vita_logit_postprocess(void* logits, int64_t ld, int64_t rows, int64_t cols, float multiplier_scale,
                           float softcapping, void* stream);
int vita_logit_postprocess_bwd(const void* y, int64_t ldy, void* grad, int64_t ldg, int64_t rows, int64_t cols,
                               float multiplier_scale, float softcapping, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RoPE.  Replaces M/core/models/common/embeddings/rotary_pos_embedding.py:84-122 (table),
 * :181-204 (apply_rotary_pos_emb_bshd) and apex fused_apply_rotary_pos_emb (:248-252).
 *
 * vita_rope_table: cos/sin tables, bf16 [n, dim/2], for integer positions pos[n] (int64):
 *   freqs = float(pos) * inv_freq[i]  (fp32, like torch.outer),  cos/sin in fp32, cast to bf16
 *   (rotary_pos_embedding.py:200-201 casts cos_/sin_ to t.dtype).  Positions are the *global*
 *   token positions of the local rows, i.e. the zig-zag slice of :36-47 is an index choice.
 * ------------------------------------------------------------------------------------------- */
int vita_rope_table(const int64_t* pos, const float* inv_freq, void* cos_out, void* sin_out,
                    int64_t n, int half_dim, void* stream);

/* cos/sin tables, bf16 [n, half_dim], from the fp32 angles Megatron's RotaryEmbedding.forward returns
 * (`freqs` [s, 1, 1, dim] = cat(freqs, freqs), rotary_pos_embedding.py:106-108; row_stride = dim): the argument
 * apply_rotary_pos_emb(t, freqs, config, cu_seqlens) receives at :232-259.  cos_ = cos(freqs).to(bf16) as :200-201. */
int vita_rope_cos_sin(const float* freqs, int64_t row_stride, void* cos_out, void* sin_out,
                      int64_t n, int half_dim, void* stream);

/* In-place RoPE on a strided [rows, heads, head_dim] bf16 view (non-interleaved halves):
 *   t = bf16(bf16(t*cos) + bf16(rotate_half(t)*sin))   — the rounding chain of :203.
 * row_stride / head_stride in elements.  sign = +1 forward, -1 backward (transpose rotation).
 * Replaces apply_rotary_pos_emb_bshd (M/core/models/common/embeddings/rotary_pos_embedding.py:181-204) on q / k views. */
int vita_rope_apply(void* t, int64_t rows, int heads, int head_dim,
                    int64_t row_stride, int64_t head_stride,
                    const void* cos_tab, const void* sin_tab, int sign, void* stream);

/* Fused RoPE over Megatron's mixed QKV activation [rows, groups, (qpg + 2) * d]
 * (layout: L/core/models/vision/intern_vit_model.py:145-197; weights R/tools/hf2mcore_long_vita.py:597-609):
 * rotates the qpg query heads and the key head of every group in place and, when kv_out != NULL,
 * also packs rotated K and V into kv_out = [kv_split][2][rows][groups / kv_split][d]: kv_split
 * contiguous all-gather messages of the context-parallel attention (split by kv head so that the
 * gather of split j+1 overlaps the attention over split j; kv_split = 1: one message). */
int vita_rope_qkv_fwd(void* mixed_qkv, int64_t rows, int groups, int q_per_group, int head_dim,
                      const void* cos_tab, const void* sin_tab, void* kv_out, int kv_split,
                      void* stream);
/* Backward of the rotation on the gradient of the mixed QKV activation (in place): applies the
 * transpose rotation to the dQ and dK heads, leaves dV untouched.
 * The autograd of apply_rotary_pos_emb_bshd (rotary_pos_embedding.py:181-204) on the mixed QKV gradient. */
int vita_rope_qkv_bwd(void* d_mixed_qkv, int64_t rows, int groups, int q_per_group, int head_dim,
                      const void* cos_tab, const void* sin_tab, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row gather / scatter with int64 indices (bit-exact data movement).
 * Replaces: VocabParallelEmbedding lookup M/core/tensor_parallel/layers.py:216-232;
 *           visual-token scatter M/core/models/common/embeddings/language_model_embedding.py:123,126,131;
 *           masked_select / masked_scatter of the logits-masked head M/core/tensor_parallel/layers.py:348,407,451,455.
 *   gather : dst[i, :] = src[idx[i], :]            i in [0, n)
 *   scatter: dst[dst_idx[i], :] = src[src_idx ? src_idx[i] : i, :]
 * Out-of-range indices set *err_flag (int32 device word, may be NULL) to 1 and are skipped.
 * ------------------------------------------------------------------------------------------- */
int vita_row_gather(const void* src, int64_t src_rows, const int64_t* idx, void* dst,
                    int64_t n, int cols, int elem_bytes, int* err_flag, void* stream);
int vita_row_scatter(const void* src, int64_t src_rows, const int64_t* src_idx,
                     void* dst, int64_t dst_rows, const int64_t* dst_idx,
                     int64_t n, int cols, int elem_bytes, int* err_flag, void* stream);

/* Ordered compaction of a boolean mask: idx_out[k] = position of the k-th true entry,
 * *count_out = number of true entries (device int64).  This is the index form of
 * torch.masked_select (layers.py:348,407).  n <= 2^31. */
int vita_mask_to_index(const uint8_t* mask, int64_t n, int64_t* idx_out, int64_t* count_out,
                       void* stream);

/* Zig-zag context-parallel index remap, the integer arithmetic of
 * M/training/utils.py:279-325,347-350 (get_batch_on_this_cp_rank + index_of_a_in_b):
 *   indices_s [n_img, tok_per_img] int64 global positions -> per element
 *   hit[n_img*tok] (uint8: position owned by this rank) and local[n_img*tok] (int64 local
 *   position after the zig-zag slice, or -1).  seq_len % (2*cp) == 0. */
int vita_cp_index_remap(const int64_t* indices_s, int64_t n, int64_t seq_len, int cp_size,
                        int cp_rank, uint8_t* hit, int64_t* local_pos, void* stream);
/* out[r] = any(mask[r, :])  — `selected_i = torch.any(mask, dim=1)` (M/training/utils.py:284,297). */
int vita_rows_any(const uint8_t* mask, int64_t rows, int cols, uint8_t* out, void* stream);
/* inv[idx[k]] = k  — renumbering of the selected images (utils.py:300 `arange(num_images)`). */
int vita_index_inverse(const int64_t* idx, int64_t n, int64_t* inv, void* stream);
/* For the k-th hit f = hit_idx[k] (flat index into [n_img, tok_per_img]):
 *   src_b = img_rank[f / tok], src_s = f % tok   (utils.py:300-304)
 *   tgt_b = indices_b[f],       tgt_s = local_pos[f]   (utils.py:306-309). */
int vita_cp_src_tgt(const int64_t* hit_idx, int64_t n_hit, int tok_per_img,
                    const int64_t* img_rank, const int64_t* indices_b, const int64_t* local_pos,
                    int64_t* src_b, int64_t* src_s, int64_t* tgt_b, int64_t* tgt_s, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 GEMM on MFMA:   C[M,N] = epilogue(A[M,K] @ W[N,K]^T)        fp32 accumulation.
 * W is row-major [N, K] exactly as torch.nn.Linear / Megatron store it, so this replaces
 * torch.matmul(total_input, weight.t()) (M/core/tensor_parallel/layers.py:270,409), the TE
 * linears of M/core/models/gpt/gpt_layer_specs.py:39-49, the ViT linears and the projector MLP
 * (M/core/models/vision/multimodal_projector.py:53-69).
 *   lda / ldw / ldc / ldr : row strides in elements.  K % 64 == 0, K >= 64; M, N arbitrary.
 * Epilogues (v = fp32 accumulator; every arrow rounds to bf16 like the unfused reference chain):
 *   VITA_EPI_NONE              C = bf16(v)
 *   VITA_EPI_BIAS              C = bf16(v + bias[n])
 *   VITA_EPI_BIAS_GELU         C = bf16(gelu_erf(bf16(v + bias[n])))              (bias may be NULL)
 *   VITA_EPI_RESIDUAL          C = bf16(R[m,n] + bf16(v + bias[n]))               (bias may be NULL)
 *   VITA_EPI_BIAS_SCALE_RES    C = bf16(R[m,n] + bf16(bf16(v + bias[n]) * scale[n]))   (LayerScale,
 *                              M/core/models/vision/intern_vit_model.py:63,77)
 *   VITA_EPI_BIAS2_GELU_TANH   C = bf16(gelu_tanh(bf16(bf16(v) + bias[n])))     (SigLIP MLP: linear_fc1 returns its
 *                              bias unfused (skip_bias_add), M/core/models/vision/siglip_vit_model.py:66-69)
 *   VITA_EPI_BIAS2_RES         C = bf16(R[m,n] + bf16(bf16(v) + bias[n]))        (SigLIP linear_proj / linear_fc2, :50-52,69-72)
 *   VITA_EPI_BIAS2_GELU        C = bf16(gelu_erf(bf16(bf16(v) + bias[n])))      (InternViT MLP under Megatron's local spec: linear_fc1 is
 *                              skip_bias_add and `bias_activation_fusion` is off, M/pretrain_long_vita.py:213 — MLP.forward adds
 *                              the bias to the bf16 product as an op of its own before F.gelu)
 *   VITA_EPI_SWIGLU            W holds [gate(N) ; up(N)] = 2N rows (fc1 = cat[gate, up],
 *                              R/tools/hf2mcore_long_vita.py:612); C[M,N] = bf16(bf16(silu(bf16(g))) * bf16(u))
 * ------------------------------------------------------------------------------------------- */
#define VITA_EPI_NONE 0
#define VITA_EPI_BIAS 1
#define VITA_EPI_BIAS_GELU 2
#define VITA_EPI_RESIDUAL 3
#define VITA_EPI_BIAS_SCALE_RES 4
#define VITA_EPI_SWIGLU 5
#define VITA_EPI_BIAS2_GELU_TANH 6
#define VITA_EPI_BIAS2_RES 7
#define VITA_EPI_BIAS2_GELU 8

int vita_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, int epilogue, const void* bias,
                   const void* scale, const void* R, int64_t ldr, void* stream);

/* The weight-gradient GEMM without transposes (r03): C[M, N] = A_t^T W_t, BOTH operands contraction-major — A_t [K, M] (row stride
 * lda), W_t [K, N] (ldw).  grad_weight = grad_output.t().matmul(total_input) (M/core/tensor_parallel/layers.py:522-523) is exactly this
 * with A_t = grad_output [tokens, out], W_t = total_input [tokens, in] as the forward left them: no vita_transpose_bf16 pass over
 * either.  M, N multiples of 256, K of 64 (VITA_ERR_UNSUPPORTED otherwise: the caller falls back to vita_transpose_bf16 + vita_gemm_bf16). */
int vita_gemm_bf16_tn(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                      int64_t K, void* stream);

/* Skinny-M GEMM for the logits-masked LM head (n_sel rows, M <= 16):
 *   logits[M, N] fp32-accumulated, stored bf16 (out_f32 == 0) or fp32 (out_f32 != 0).
 * Replaces torch.matmul on the masked rows, M/core/tensor_parallel/layers.py:402-409. */
/* ABI 16: C[M, N] = A W with A [M, K] row-major and W [K, N] contraction-major — the input-gradient GEMM
 * `grad_input = grad_output.matmul(weight)` (M/core/tensor_parallel/layers.py:444,453) with the weight [out, in] exactly as the forward holds
 * it: no transposed copy.  M, N multiples of 256, K a multiple of 64, rows 16-byte aligned; other shapes: VITA_ERR_UNSUPPORTED (the caller
 * transposes and uses vita_gemm_bf16). */
int vita_gemm_bf16_nn(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      void* stream);

/* ABI 15: vita_gemm_bf16_tn with the contraction cut into `splits` ranges — for weight gradients whose output is only a few 256 x 256
 * tiles over a long token contraction (the ViT's linears; the decoder's narrow qkv / proj gradients).  Every (split, tile) workgroup
 * writes an fp32 partial to `workspace` ([splits][M][N] floats, vita_gemm_tn_splitk_workspace_bytes), a second kernel sums the splits and
 * rounds once to bf16.  splits <= 1: vita_gemm_bf16_tn.  K / 64 K-tiles are dealt ceil(K / 64 / splits) per split. */
size_t vita_gemm_tn_splitk_workspace_bytes(int64_t M, int64_t N, int splits);
int vita_gemm_bf16_tn_splitk(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                             int64_t K, int splits, void* workspace, void* stream);
/* ABI 15: out[c] += sum over rows of float(x[r][c]) — grad_bias = grad_output.sum(dim=0) (M/core/tensor_parallel/layers.py:524) in one
 * pass over grad_output.  out: fp32 [cols], zeroed by the caller; cols % 4 == 0. */
int vita_colsum_bf16(const void* x, int64_t ldx, float* out, int64_t rows, int cols, void* stream);
/* ABI 18: the same sums added in a FIXED order — bit-reproducible from run to run, as the reference's grad_output.sum(dim=0) is (the form
 * above adds its row blocks with fp32 atomics, in arrival order).  workspace: vita_colsum_workspace_bytes(rows, cols) bytes, 16-byte aligned;
 * out: fp32 [cols], 16-byte aligned, zeroed (or holding a running sum) on entry. */
size_t vita_colsum_workspace_bytes(int64_t rows, int cols);
int vita_colsum_bf16_ordered(const void* x, int64_t ldx, float* out, int64_t rows, int cols, void* workspace, void* stream);

int vita_gemm_skinny_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C,
                          int64_t ldc, int M, int64_t N, int64_t K, int out_f32, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flash attention forward (online softmax, fp32 statistics, bf16 I/O), head_dim 64, 96 or 128 (96: SigLIP's 72 zero-padded by the
 * caller, M/core/models/vision/siglip_vit_model.py:29-86; anything else: VITA_ERR_UNSUPPORTED).
 * Replaces flash_attn_func (ViT, M/core/transformer/dot_product_attention.py:318-326),
 * transformers._flash_attention_forward (LLM CP=1, :374-390) and TransformerEngine's
 * AttnFuncWithCP ring (LLM CP>1, M/core/models/gpt/gpt_layer_specs.py:40).
 *
 * Sequence geometry is described by *chunks* so that one kernel serves plain causal, non-causal
 * and zig-zag context-parallel attention:
 *   - the local Q rows are n_q_chunks chunks of chunk_len rows, chunk i carrying the global
 *     chunk id q_chunk_gid[i];
 *   - the K/V rows visible to this rank are n_kv_chunks chunks of chunk_len rows; chunk j starts
 *     at row kv_chunk_row[j] of the K / V buffers and carries global chunk id kv_chunk_gid[j];
 *   - causal != 0: query (gq, i) attends key (gk, j) iff gk < gq or (gk == gq and j <= i);
 *     causal == 0: everything is visible; kv_valid (<= chunk_len) masks the padded tail of a
 *     single-chunk sequence (ViT: 1025 tokens).
 * q/k/v/o strides are in elements; `batch` replicates the geometry (ViT frames).
 * GQA: query head h uses kv head h / (n_q_heads / n_kv_heads); with G = n_q_heads / n_kv_heads the
 * address of query head (g, j) is q + g * q_group_stride + j * q_head_stride (q_group_stride == 0
 * means G * q_head_stride) so Q can be read in place from Megatron's mixed QKV activation;
 * o_group_stride likewise.
 * lse (fp32 [batch, n_q_heads, n_q_rows], natural log) may be NULL.
 * All tables are HOST pointers (<= 64 chunks); they are copied into kernel arguments.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; int64_t q_batch_stride, q_row_stride, q_head_stride, q_group_stride;
  const void* k; int64_t k_batch_stride, k_row_stride, k_head_stride;
  const void* v; int64_t v_batch_stride, v_row_stride, v_head_stride;
  void* o;       int64_t o_batch_stride, o_row_stride, o_head_stride, o_group_stride;
  float* lse;
  int batch, n_q_heads, n_kv_heads, head_dim;
  int64_t chunk_len;        /* rows per chunk                                             */
  int64_t q_valid;          /* valid rows of the last q chunk  (<= chunk_len)             */
  int64_t kv_valid;         /* valid rows of the last kv chunk (<= chunk_len)             */
  int n_q_chunks, n_kv_chunks;
  const int32_t* q_chunk_gid;      /* host [n_q_chunks]  */
  const int32_t* kv_chunk_gid;     /* host [n_kv_chunks] */
  const int64_t* kv_chunk_row;     /* host [n_kv_chunks] */
  int causal;
  float softmax_scale;
  const int32_t* q_seg_start;      /* DEVICE [rows] or NULL: packed sequences (flash_attn_varlen_func with
                                      cu_seqlens, M/core/transformer/dot_product_attention.py:334-367): first row of
                                      the segment each query row belongs to; keys before it are masked.  Needs
                                      batch 1, causal, single-chunk geometry. */
} vita_attn_params;

int vita_flash_attn_fwd(const vita_attn_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Context-parallel core attention for callers without PyTorch (cp_attn.hip): the K/V exchange of one decoder layer over
 * RCCL + the zig-zag chunk-table attention.  Replaces TransformerEngine's AttnFuncWithCP (the core_attention of
 * M/core/models/gpt/gpt_layer_specs.py:40 under --context-parallel-size > 1: a CP-1-step P2P ring) with one all-gather per
 * kv-head split issued up front on a communication stream — xGMI is point to point, every peer pushes over its own link —
 * and, in the backward, one reduce-scatter of dK / dV per split.  Rank r owns zig-zag chunks {r, 2 CP - 1 - r} of the
 * sequence (M/training/utils.py:329-341); s_local = the two chunks together.
 *
 * vita_cp_unique_id / vita_cp_init / vita_cp_destroy: the RCCL communicator of the context-parallel group (ncclGetUniqueId on
 *   one rank, shipped to the others by any channel, then ncclCommInitRank), the communication stream and its events.  The only
 *   persistent state of the library (SURVEY.md §8b "Ownership"); RCCL is resolved at run time, so nothing here is needed —
 *   or loaded — on the single-GPU path.  cp_size = 1 is valid (the exchange degenerates to a copy through RCCL).
 *   unique_id128 = NULL creates a context WITHOUT a communicator ("external exchange"): the host moves the shards itself (its own
 *   transport — e.g. peer copies on the copy engines — or one process simulating the ranks): vita_cp_attn_fwd then expects
 *   `workspace` to hold the gathered K/V [n_split][cp_size][2][s_local][n_kv_heads / n_split][head_dim] (ordered before `stream`;
 *   its own slot may be left unwritten when scratch enables own-chunks-first) and vita_cp_attn_bwd leaves the reduction of
 *   p->dkv_workspace over the ranks to the caller.
 * vita_cp_attn_fwd: q = this rank's rotated queries as a grouped view (head h of kv group g at q + row * q_row_stride +
 *   g * q_group_stride + h * q_head_stride); kv_packed = its rotated K and V packed [n_split][2 (K | V)][s_local][n_kv_heads /
 *   n_split][head_dim] (what vita_rope_qkv_fwd writes); out [s_local][n_q_heads][head_dim] via out_row/head_stride; lse
 *   (optional) [n_q_heads][s_local].  workspace = vita_cp_attn_workspace_bytes(...) of device memory, caller-owned: the gathered
 *   K/V; it is the `k`/`v` of the backward.  Everything is stream-ordered behind `stream`.
 *   Own chunks first (what TransformerEngine's ring does with its local block): gathers 1.. run under the attention of
 *   the split before them, gather 0 has nothing in front of it.  With p->scratch = vita_cp_attn_scratch_bytes(...) of device
 *   memory (one split's output + two lse vectors) and cp_size > 1, split 0 attends to the rank's OWN two chunks straight from
 *   kv_packed while gather 0 is in flight, then to the 2 CP - 2 remote chunks, and vita_attn_merge joins the two partials;
 *   scratch = NULL keeps one launch per split behind its gather.
 *   Choosing n_split: every split is one attention launch over n_q_heads / n_split heads x s_local / 256 row tiles, one workgroup
 *   per CU, and short launches waste their last round: at s_local = 16384, 40 : 8 heads (2560 workgroups per layer) 1 / 2 / 4 splits
 *   take 18.0-18.3 / 18.8-19.0 / 18.9-19.9 ms per layer (profiles/r03_cp8_rank_attn.jsonl); the host mirror keeps >= 2560
 *   workgroups per launch (long_vita_amd.ops.cp_kv_split).
 * vita_cp_attn_bwd: d_out like out; lse from the forward; delta from vita_attn_delta; dq like q; dkv_packed like kv_packed
 *   receives this rank's dK / dV; p->dkv_workspace (same size as workspace) takes the gathered-layout dK / dV. */
typedef struct vita_cp_context vita_cp_context;
typedef struct {
  const void* q; int64_t q_row_stride, q_head_stride, q_group_stride;
  const void* kv_packed;
  void* out; int64_t out_row_stride, out_head_stride;
  float* lse;
  int64_t s_local;
  int n_q_heads, n_kv_heads, head_dim, n_split;
  float softmax_scale;
  void* workspace; size_t workspace_bytes;
  void* dkv_workspace;
  void* scratch; size_t scratch_bytes;      /* optional (ABI 13): vita_cp_attn_scratch_bytes(...) enables "own chunks first" */
} vita_cp_attn_params;
int vita_cp_unique_id(void* id_out128);
int vita_cp_init(vita_cp_context** ctx, int cp_size, int cp_rank, const void* unique_id128);
int vita_cp_destroy(vita_cp_context* ctx);
size_t vita_cp_attn_workspace_bytes(int cp_size, int64_t s_local, int n_kv_heads, int head_dim);
size_t vita_cp_attn_scratch_bytes(int64_t s_local, int n_q_heads, int n_split, int head_dim);
int vita_cp_attn_fwd(vita_cp_context* ctx, const vita_cp_attn_params* p, void* stream);
int vita_cp_attn_bwd(vita_cp_context* ctx, const vita_cp_attn_params* p, const void* d_out, const float* lse,
                     const float* delta, void* dq, void* dkv_packed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ViT front / back ends.
 * vita_patchify14: im2col of the 14x14 / stride-14 patch conv (M/core/models/vision/intern_vit_model.py:139-145,203-205):
 *   images [n, 3, H, W] bf16 -> patches [n * (H/14) * (W/14), k_pad] bf16, column = c*196 + dy*14 + dx
 *   (the flattening of Conv2d.weight [out, 3, 14, 14]); columns 588..k_pad-1 are zero.
 * vita_vit_assemble: x[n, 1 + P, h] = cat(cls, patch_embeds[n, P, h]) + pos[1 + P, h]   (:207-216),
 *   rounding bf16 after the add.
 * vita_pixel_shuffle_ln: drop cls, pixel-shuffle x0.5 and LayerNorm(4*h) in one pass
 *   (M/pretrain_long_vita.py:467-483,572-582 + :443-446):
 *   x [n, 1 + g*g, h] -> y [n, (g/2)*(g/2), 4h].
 * ------------------------------------------------------------------------------------------- */
int vita_patchify14(const void* images, void* patches, int64_t n, int H, int W, int k_pad,
                    void* stream);
int vita_vit_assemble(const void* patch_embeds, const void* cls_token, const void* pos_emb,
                      void* x, int64_t n, int n_patches, int hidden, int has_cls, void* stream);
int vita_pixel_shuffle_ln(const void* x, const void* w, const void* b, void* y, int64_t n,
                          int grid, int hidden, int has_cls, float eps, void* stream);

/* ABI 15 — the same front / back end under Megatron's module seams (InternViTModel.forward / SigLIPViTModel.forward,
 * M/core/models/vision/intern_vit_model.py:190-261, siglip_vit_model.py:165-228; MegatronVisionModel.forward_downsample /
 * forward_projection, M/pretrain_long_vita.py:436-483), forward AND backward (stage 2 trains the encoder, the projector always trains):
 *   vita_patchify14_ex    token_major != 0: patch row = patch * n + image — Megatron's [s, b, h] order, so the conv GEMM's output IS
 *                         rows 1.. of the encoder input and the backward's gradient rows line up with the patch rows (no transposes).
 *   vita_vit_assemble_ex  token_major as above for patch_embeds and x; pos_row0 = first row of the position table that is looked up
 *                         (InternViTModel without a class token uses rows 1 .. seq, :141-143).
 *   vita_vit_assemble_bwd d_pos[seq, h] = bf16(sum over images of dx[., s, :]) (fp32 sums) — the position table's gradient; its row 0
 *                         is also the class token's (x[:, 0] = cls + pos[0]).
 *   vita_pixel_shuffle_ln_ex   x addressed as x[image * img_stride + token * tok_stride + c] (elements, multiples of 8): the
 *                         contiguous [n, seq, h] tensor or the permuted view of the encoder's [s, b, h] output (no copy);
 *                         norm == 0: the pure permutation (forward_downsample alone; w / b unused).
 *   vita_pixel_shuffle_ln_bwd  dx (x's strides; NULL = frozen encoder; the class token's row receives zeros) and, norm != 0,
 *                         dgamma / dbeta (fp32 [4 h], accumulated: caller zeroes).  4 h <= 4096. */
int vita_patchify14_ex(const void* images, void* patches, int64_t n, int H, int W, int k_pad, int token_major, void* stream);
int vita_vit_assemble_ex(const void* patch_embeds, const void* cls_token, const void* pos_emb, void* x, int64_t n, int n_patches,
                         int hidden, int has_cls, int pos_row0, int token_major, void* stream);
int vita_vit_assemble_bwd(const void* dx, void* d_pos, int64_t n, int seq, int hidden, int token_major, void* stream);
int vita_pixel_shuffle_ln_ex(const void* x, const void* w, const void* b, void* y, int64_t n, int grid, int hidden, int has_cls,
                             float eps, int norm, int64_t img_stride, int64_t tok_stride, void* stream);
int vita_pixel_shuffle_ln_bwd(const void* dy, const void* x, const void* w, void* dx, float* dgamma, float* dbeta, int64_t n,
                              int grid, int hidden, int has_cls, float eps, int norm, int64_t img_stride, int64_t tok_stride,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pass (training step, reference: torch autograd over the forward modules; the only
 * first-party backward code is M/core/tensor_parallel/layers.py:416-534).
 * ------------------------------------------------------------------------------------------- */

/* dst[c][r] = src[r][c] (bf16).  Used to feed dgrad / wgrad to vita_gemm_bf16, whose operands are
 * both contraction-contiguous:  dX = dY @ W  = gemm(dY, W^T),   dW = dY^T @ X = gemm(dY^T, X^T)
 * (grad_input = grad_output.matmul(weight), grad_weight = grad_output.t().matmul(total_input):
 * layers.py:444, :522-523). */
int vita_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows,
                        int64_t cols, void* stream);

/* RMSNorm backward of vita_rmsnorm_fwd's expression: dx bf16 (+ `res` when not NULL: the gradient
 * arriving through the residual connection, dx = bf16(res + bf16(dx_norm))); dw_acc fp32 [cols] is
 * ACCUMULATED (caller zeroes it; may be NULL).
 * The autograd of RMSNorm._norm / forward (M/core/transformer/custom_layers/transformer_engine.py:74-79). */
int vita_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* res, void* dx,
                     float* dw_acc, int64_t rows, int cols, float eps, void* stream);

/* SwiGLU on the unfused fc1 output y = [gate | up] ([rows, 2*ffn]):  a = bf16(bf16(silu(g)) * u);
 * backward writes dy = [dgate | dup].
 * Megatron MLP's `silu(gate) * up` under `--swiglu` (the MLP the spec of M/core/models/gpt/gpt_layer_specs.py:93-104 builds) and its autograd. */
int vita_swiglu_fwd(const void* y, void* a, int64_t rows, int ffn, void* stream);
int vita_swiglu_bwd(const void* y, const void* da, void* dy, int64_t rows, int ffn, void* stream);

/* dx = bf16(dy * gelu_erf'(x)) over n elements (projector MLP, multimodal_projector.py:53-69). */
int vita_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);

/* LayerNorm parameter gradients (projector pre-norm, M/pretrain_long_vita.py:443-446; its input is the
 * frozen ViT's output, so no dx):  dgamma += sum_rows dy*xhat, dbeta += sum_rows dy (fp32, accumulated).
 * prenormalized != 0: x already holds xhat (e.g. vita_pixel_shuffle_ln run with w = 1, b = 0). */
int vita_layernorm_param_grad(const void* dy, const void* x, float* dgamma, float* dbeta,
                              int64_t rows, int cols, float eps, int prenormalized, void* stream);

/* ViT training (reference stage 2 trains the encoder: R/scripts/megatron/qwen25/finetune_..._stage2.sh has no --vision-model-freeze).
 * vita_layernorm_bwd: backward of torch.nn.LayerNorm / TENorm (the block norms of InternViTTransformerLayer,
 *   M/core/models/vision/intern_vit_model.py:46,72): dx, and dgamma / dbeta accumulated into fp32 [cols] (caller zeroes).
 * vita_gelu_fwd: a = bf16(gelu(x)) as a kernel of its own (training keeps the pre-activation for vita_gelu_bwd);
 *   tanh_form != 0: the tanh approximation (SigLIP).
 * vita_bias_scale_res_fwd / _bwd: `hidden = residual + (out + bias) * ls` of InternViTTransformerLayer.forward (:60-66, :79-82)
 *   with a bf16 rounding after each of the three ops, bias / scale optional (NULL); backward: dx = bf16(g * scale),
 *   d_bias += sum_rows dx, d_scale += sum_rows g * bf16(x + bias) (fp32 [cols], caller zeroes; either may be NULL);
 *   d(residual) = g. */
int vita_layernorm_bwd(const void* dy, const void* x, const void* w, void* dx, float* dgamma, float* dbeta,
                       int64_t rows, int cols, float eps, void* stream);
int vita_gelu_fwd(const void* x, void* a, int64_t n, int tanh_form, void* stream);
/* ABI 14: dx = bf16(dy * gelu_tanh'(x)) — the backward of `partial(F.gelu, approximate="tanh")`, SigLIP's activation
 * (M/pretrain_long_vita.py:290; SigLIPViTTransformerLayer's MLP, M/core/models/vision/siglip_vit_model.py:29-86). */
int vita_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int vita_bias_scale_res_fwd(const void* x, const void* bias, const void* scale, const void* residual, void* out,
                            int64_t rows, int cols, void* stream);
int vita_bias_scale_res_bwd(const void* g, const void* x, const void* bias, const void* scale, void* dx, float* d_bias,
                            float* d_scale, int64_t rows, int cols, void* stream);

/* Vocabulary cross-entropy of the selected rows (TP = 1 form of Megatron's vocab-parallel CE, called at
 * M/core/models/multimodal/gpt_vl_model.py:414):  loss[i] = logsumexp(float(logits[i])) - logits[i, label[i]];
 * when dlogits != NULL also dlogits[i] = bf16((softmax - onehot) * grad_scale[i]) (grad_scale NULL = 1).
 * A label outside [0, vocab) (the datasets pad with IGNORE_TOKEN_ID = -100, M/pretrain_long_vita.py:751) is treated as Megatron's
 * vocab_parallel_cross_entropy treats it — masked target: loss[i] = log sum exp(logits[i] - max), no one-hot term in dlogits[i] — and
 * additionally sets *err_flag when err_flag != NULL (ABI 17; before: the row was skipped). */
int vita_ce_loss(const void* logits, int64_t ld, const int64_t* labels, float* loss, void* dlogits,
                 int64_t ld_d, const float* grad_scale, int64_t rows, int vocab, int* err_flag,
                 void* stream);
/* ABI 15: the same on fp32 logits with fp32 dlogits — what `tensor_parallel.vocab_parallel_cross_entropy(logits.float(), labels)`
 * receives (megatron LanguageModule.compute_language_model_loss, called at gpt_vl_model.py:414); loss may be NULL when only
 * dlogits is wanted (the autograd backward re-runs the row pass with grad_scale = the incoming gradient). */
int vita_ce_loss_f32(const float* logits, int64_t ld, const int64_t* labels, float* loss, float* dlogits,
                     int64_t ld_d, const float* grad_scale, int64_t rows, int vocab, int* err_flag, void* stream);

/* ABI 17: the vocabulary-PARALLEL cross entropy (TP > 1) — megatron.core.tensor_parallel.cross_entropy.vocab_parallel_cross_entropy
 * as LanguageModule.compute_language_model_loss calls it at M/core/models/multimodal/gpt_vl_model.py:414.  Each rank keeps its
 * [rows, vocab_local] shard (logits bf16, or fp32 when is_f32) and its labels in GLOBAL vocabulary ids:
 *   vita_ce_vp_stats   stats[row] = {max, sum exp(l - max), predicted raw logit or 0, 1 if vocab_start <= label < vocab_start + vocab_local};
 *   (caller: all-gather the [rows, 4] fp32 records over the tensor-parallel group -> stats_all [tp, rows, 4], rank-major)
 *   vita_ce_vp_finish  loss[row] = log(sum exp) - (predicted - max), or log(sum exp) when no shard holds the label (masked target);
 *                      row_stat[row] = {global max, global sum exp} for the backward; loss may be NULL;
 *   vita_ce_vp_grad    dlogits[row, v] = (exp(l - max) / sumexp - [v == label - vocab_start]) * grad_scale[row]  (NULL = 1), shard only. */
int vita_ce_vp_stats(const void* logits, int is_f32, int64_t ld, const int64_t* labels, int64_t vocab_start, float* stats,
                     int64_t rows, int vocab_local, void* stream);
int vita_ce_vp_finish(const float* stats_all, int tp, int64_t rows, float* loss, float* row_stat, void* stream);
int vita_ce_vp_grad(const void* logits, int is_f32, int64_t ld, const int64_t* labels, int64_t vocab_start, const float* row_stat,
                    const float* grad_scale, void* dlogits, int64_t ld_d, int64_t rows, int vocab_local, void* stream);

/* dst_f32[idx[i], :] += float(src_bf16[i, :]) — word-embedding weight gradient (the backward of
 * M/core/tensor_parallel/layers.py:216-232); idx[i] < 0 skips row i (visual-token positions). */
int vita_row_scatter_add_f32(const void* src, const int64_t* idx, float* dst, int64_t dst_rows,
                             int64_t n, int cols, int* err_flag, void* stream);

/* delta[h, row] = sum_d float(dO[row,h,d]) * float(O[row,h,d])  (attention backward pre-pass).
 * First kernel of the flash-attention backward that stands behind M/core/transformer/dot_product_attention.py:374-390. */
int vita_attn_delta(const void* o, const void* d_o, float* delta, int64_t rows, int heads,
                    int head_dim, int64_t o_row_stride, int64_t o_head_stride, int64_t do_row_stride,
                    int64_t do_head_stride, void* stream);

/* Flash attention backward (head_dim 128 | 96 | 64, causal through the chunk tables, batch 1) over the same chunk geometry as
 * vita_flash_attn_fwd (chunk_len % 128 == 0).  q/k/v are the forward's (rotated) inputs, lse its
 * log-sum-exp output [n_q_heads, n_q_rows], delta from vita_attn_delta.  dk/dv are written for EVERY
 * key row of the visible K/V buffer (under context parallelism: the gathered layout, to be
 * reduce-scattered by the caller). */
typedef struct {
  const void* q; int64_t q_row_stride, q_head_stride, q_group_stride;
  const void* k; int64_t k_row_stride, k_head_stride;
  const void* v; int64_t v_row_stride, v_head_stride;
  const void* d_o; int64_t do_row_stride, do_head_stride;
  const float* lse;
  const float* delta;
  void* dq; int64_t dq_row_stride, dq_head_stride, dq_group_stride;
  void* dk; int64_t dk_row_stride, dk_head_stride;
  void* dv; int64_t dv_row_stride, dv_head_stride;
  int n_q_heads, n_kv_heads, head_dim;
  int64_t chunk_len;
  int n_q_chunks, n_kv_chunks;
  const int32_t* q_chunk_gid;      /* host */
  const int32_t* kv_chunk_gid;     /* host */
  const int64_t* kv_chunk_row;     /* host */
  float softmax_scale;
  const int32_t* q_seg_start;      /* DEVICE [rows] or NULL: packed sequences, as in vita_attn_params */
  const int32_t* k_seg_end;        /* DEVICE [rows] or NULL: one past the last row of each key row's segment */
} vita_attn_bwd_params;

int vita_flash_attn_bwd(const vita_attn_bwd_params* p, void* stream);
/* ABI 15: the two passes separately — VITA_ATTN_BWD_DKV (dK + dV of every key row: needs p->dk, p->dv) and VITA_ATTN_BWD_DQ
 * (needs p->dq); with both bits dK / dV are launched first.  A context-parallel caller runs DKV, starts the reduce-scatter of
 * dK / dV on its communication stream, then runs DQ under it (vita_cp_attn_bwd; TE's ring does the same with its P2P steps). */
#define VITA_ATTN_BWD_DQ 1
#define VITA_ATTN_BWD_DKV 2
int vita_flash_attn_bwd_parts(const vita_attn_bwd_params* p, int parts, void* stream);

/* Merge of two attention partials over DISJOINT key sets (their natural-log lse from vita_flash_attn_fwd), in place into the first:
 *   lse = log(exp(lse_a) + exp(lse_b)),  O_a = O_a exp(lse_a - lse) + O_b exp(lse_b - lse)      (a part that saw no key: lse = -inf).
 * o_a / o_b [rows, heads, 128] bf16 with row / head strides in elements, lse_a / lse_b [heads, rows] fp32.
 * Context parallelism: a rank attends to its own zig-zag chunks while the K/V all-gather of the first kv-head split is in flight,
 * then to the remote chunks, and merges — replaces the "local block first" step of TransformerEngine's AttnFuncWithCP ring
 * (reached from M/core/models/gpt/gpt_layer_specs.py:40); SURVEY.md 8(e). */
int vita_attn_merge(void* o_a, int64_t oa_row_stride, int64_t oa_head_stride, float* lse_a, const void* o_b,
                    int64_t ob_row_stride, int64_t ob_head_stride, const float* lse_b, int64_t rows, int heads, int head_dim,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Single-token decode against a sequence-sharded KV cache.
 * The decode loop M/inference/text_generation/generation.py:123-205 with --use-kv-cache feeds
 * tokens[:, prev:ctx] (one token) through the model; under context parallelism the reference turns
 * the cache off (server_cp .sh:184) and re-prefills.  These three kernels are the per-token path:
 *
 * vita_gemv_bf16: y[N] = epilogue(W[N,K] . x[K]) for one token (M = 1) — the linears of
 *   M/core/tensor_parallel/layers.py:825 at sq = 1.  Epilogues NONE / BIAS / RESIDUAL / SWIGLU
 *   (W = cat[gate, up], N = ffn) with the rounding chain of vita_gemm_bf16.
 * vita_decode_attn_partial: the new token's rotated query (qpg heads per kv group, strided view)
 *   against `len` cached rows k_cache/v_cache[len][groups][128] (row / group strides in elements;
 *   len_dev != NULL: the row count is min(len, *len_dev), an int32 in device memory, so that a
 *   captured hipGraph of the token step can be replayed as the cache grows);
 *   n_splits x groups workgroups each write an un-normalised partial: max and sum in the log2
 *   domain part_m, part_l [n_splits][heads] and part_o [n_splits][heads][128] (fp32).
 * vita_decode_attn_merge: merges nparts partials per head (part p at part_m/l + p*part_ml_stride,
 *   part_o + p*part_o_stride, in floats).  out_bf16 != NULL: normalised context
 *   [heads][128] bf16; else the merged partial (out_m, out_l [heads], out_o [heads][128]) that the
 *   ranks exchange (one all-gather of (128+2)*heads floats per layer) and merge again.
 * Replaces the inference branch of Megatron's Attention.forward (_adjust_key_value_for_inference +
 * core attention at sq = 1), which M/core/transformer/dot_product_attention.py:153 is called from. */
int vita_gemv_bf16(const void* x, const void* W, int64_t ldw, void* y, int64_t N, int64_t K,
                   int epilogue, const void* bias, const void* R, void* stream);
int vita_decode_attn_partial(const void* q, int64_t q_group_stride, int64_t q_head_stride,
                             const void* k_cache, const void* v_cache, int64_t kv_row_stride,
                             int64_t kv_group_stride, int len, const void* len_dev, int n_splits,
                             int groups, int qpg, int head_dim, float softmax_scale, void* part_m,
                             void* part_l, void* part_o, void* stream);
int vita_decode_attn_merge(const void* part_m, const void* part_l, const void* part_o, int nparts,
                           int64_t part_ml_stride, int64_t part_o_stride, int heads, int head_dim, void* out_m, void* out_l, void* out_o,
                           void* out_bf16, void* stream);

/* out[n] = bf16(a[n] + b[n]) (n % 8 == 0): the residual add that follows the tensor-parallel all-reduce of a
 * row-parallel linear's output (RowParallelLinear + bias_dropout_add, TP > 1 only; at TP = 1 the add is the GEMM's
 * RESIDUAL epilogue).
 * RowParallelLinear.forward + bias_dropout_add (M/core/tensor_parallel/layers.py:1059-1115). */
int vita_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* One decoder layer for one token, launched from C: vita_decode_layer_attn = RMSNorm (fused into the
 * GEMV) + QKV GEMV + bias -> RoPE + K/V append -> decode attention partial -> merge; vita_decode_layer_mlp
 * = [merge of the CP ranks' partials] -> o-proj GEMV + residual -> RMSNorm + fc1 GEMV + SwiGLU -> fc2 GEMV
 * + residual.  Same kernels and rounding chains as the separate entry points; the split is where the
 * context-parallel all-gather of the packed partial (msg: heads*head_dim + 2*heads floats) sits.
 * Replaces TransformerLayer.forward at sq = 1 (the TE layer spec of M/core/models/gpt/gpt_layer_specs.py:35-49
 * under the decode loop M/inference/text_generation/generation.py:127-131). */
typedef struct {
  const void *ln1, *qkv_w, *qkv_b, *o_w, *ln2, *fc1_w, *fc2_w; /* one layer, Megatron layout, bf16 */
  int hidden, heads, kv_groups, head_dim, ffn;
  float eps, softmax_scale;
  void* h;                       /* [hidden] residual stream, updated in place */
  const void *cos, *sin;         /* [head_dim / 2] bf16 for this token's position */
  void *k_cache, *v_cache;       /* this rank's shard [capacity][kv_groups][head_dim] (strides in elements) */
  int64_t kv_row_stride, kv_group_stride;
  int capacity;
  int append_row;                /* >= 0: row that receives this token's K / V; < 0: another rank owns the token */
  int len;                       /* rows attended to (including the appended one) */
  int n_splits;
  void *qkv, *ctx, *act;         /* scratch: [(heads + 2 kv_groups) head_dim], [heads head_dim], [ffn] bf16 */
  void *part_m, *part_l, *part_o; /* scratch fp32: [n_splits][heads] x 2, [n_splits][heads][head_dim] */
  void* msg;                     /* attn entry, CP > 1: packed partial written here instead of ctx */
  const void* gathered;          /* mlp entry, CP > 1: [n_ranks][heads head_dim + 2 heads] fp32 */
  int n_ranks;
} vita_decode_layer_params;
int vita_decode_layer_attn(const vita_decode_layer_params* p, void* stream);
int vita_decode_layer_mlp(const vita_decode_layer_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Frame / image preprocessing — ImageProcessor.process_images and dynamic_preprocess,
 * H/data/processor/image_processor.py:180-223,404-448, then the bf16 cast of
 * M/tasks/inference/module.py:693.  frames [n][height][width][3] uint8 RGB (frame_stride bytes apart)
 *   -> pad_to_square != 0: expand2square with pad_rgb (:189-201)
 *   -> Pillow BICUBIC resize to out_w x out_h (:206-208 / :429, :443; 22-bit fixed-point separable passes with a
 *      uint8 intermediate, bit-exact)
 *   -> optionally pasted at (off_x, off_y) onto a canvas_w x canvas_h canvas (resize_and_pad_image :322-362; the caller
 *      pre-fills `images` with the normalised pad colour; canvas_w <= 0: no canvas)
 *   -> cut into tile x tile blocks in row-major order (the crop loops :431-441, :365-384; tile == out_w == out_h: one block)
 *   -> (x * 1.0 / 255.0 - mean) / std in float32 (:210-215) -> images [n * blocks][3][tile][tile] bf16.
 * h_bounds [out_w][2] / h_coeffs [out_w][h_ksize] and v_bounds [out_h][2] / v_coeffs [out_h][v_ksize] (int32,
 * device) are Pillow's precompute_coeffs / normalize_coeffs_8bpc tables for the (padded) source width -> out_w
 * and height -> out_h; pad_rgb, mean, std_ are HOST arrays of 3.
 * tmp: device scratch of n * source_height * out_w * 3 bytes; u8_out (optional): the uint8 resize result
 * [n][out_h][out_w][3]. */
int vita_frames_resize_norm(const void* frames, int64_t frame_stride, int n, int height, int width,
                            int pad_to_square, const int* pad_rgb, int out_w, int out_h, int tile,
                            int canvas_w, int canvas_h, int off_x, int off_y, const void* h_bounds, const void* h_coeffs, int h_ksize, const void* v_bounds,
                            const void* v_coeffs, int v_ksize, const float* mean, const float* std_,
                            void* tmp, void* images, void* u8_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITA_HIP_H */
