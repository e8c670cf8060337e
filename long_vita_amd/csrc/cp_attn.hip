// Context-parallel core attention behind the C ABI: the K/V exchange of one decoder layer + the zig-zag chunk-table attention,
// for callers that are not Python (vita_cp_attn_fwd / _bwd, vita_cp_init / vita_cp_destroy owning the RCCL
// communicator and the communication stream).  The Python operator mirror (dot_product_attention.DotProductAttention.forward_cp)
// does the same over torch.distributed; this is that path without torch.
//
// Replaces TransformerEngine's AttnFuncWithCP (the `core_attention=TEDotProductAttention` of M/core/models/gpt/gpt_layer_specs.py:40
// under --context-parallel-size > 1): a CP-1-step P2P ring.  xGMI is point to point, so a ring is single-link bound; here every
// rank pushes its shard to every peer at once:
//   forward : per kv-head split j ONE ncclAllGather of the packed shard [2 (K | V)][S_l][hg][d] -> [CP][2][S_l][hg][d], all splits
//             issued up front on the communication stream; the attention over split j (compute stream) waits only for the
//             event behind gather j, so gathers j+1.. run under it.  Global chunk 2p+h of the gathered buffer is zig-zag chunk
//             (h ? 2CP-1-p : p) (M/training/utils.py:329-341); the kernel addresses it through chunk tables.  Gather 0 runs under
//             the attention over the rank's OWN two chunks (read straight from the packed shard), the remote chunks follow it and
//             vita_attn_merge joins the two partials (needs p->scratch).
//             Measured on one MI355X with a stand-in of RCCL's real kernel footprint (280 registers, one wave per SIMD:
//             tools/probe_comm_overlap.py, profiles/r03_hwprobe_comm_overlap.txt): a communication kernel enqueued under the
//             attention gets whole CUs at the first workgroup turnover (<= 2.8 ms at the 128K / CP = 8 geometry, where the next
//             gather is needed 5.6 ms later) and costs the attention <= 1.2 %.
//   backward: vita_flash_attn_bwd_parts writes dK / dV of every visible key in the gathered layout (the dK + dV pass runs FIRST),
//             ONE ncclReduceScatter (sum, bf16) per split returns each rank its shard — issued right behind that split's dK / dV
//             pass, so it runs under the split's dQ pass (3 of the backward's 8 GEMM units) and under the later splits (r04).
// RCCL is resolved at run time (dlsym): inside a PyTorch process that is the librccl torch already loaded, a plain C host gets
// librccl.so from the ROCm installation.  No symbol of this file is needed by the single-GPU path.
#include "vita_common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

extern "C" int vita_flash_attn_fwd(const vita_attn_params* p, void* stream);
extern "C" int vita_flash_attn_bwd_parts(const vita_attn_bwd_params* p, int parts, void* stream);
extern "C" int vita_attn_merge(void* o_a, int64_t oa_row_stride, int64_t oa_head_stride, float* lse_a, const void* o_b,
                               int64_t ob_row_stride, int64_t ob_head_stride, const float* lse_b, int64_t rows, int heads,
                               int head_dim, void* stream);

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
constexpr int kNcclBfloat16 = 9, kNcclSum = 0;        // rccl.h: ncclDataType_t / ncclRedOp_t
constexpr int kMaxSplit = 8, kMaxCP = 16;

struct Rccl {
  int (*GetUniqueId)(ncclUniqueId*);
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  int (*CommDestroy)(ncclComm_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  bool ok;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    memset(&x, 0, sizeof(x));
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return x;
    }
    x.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
    x.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
    x.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    x.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    x.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclReduceScatter");
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.ReduceScatter;
    return x;
  }();
  return r;
}

}  // namespace

struct vita_cp_context {
  ncclComm_t comm;
  int cp_size, cp_rank;
  hipStream_t comm_stream;
  hipEvent_t ready;                 // compute -> communication: the packed shard is written
  hipEvent_t gathered[kMaxSplit];   // communication -> compute: gather j has landed
  hipEvent_t reduced;               // communication -> compute: the reduce-scatters have landed
};

extern "C" int vita_cp_unique_id(void* id_out128) {
  if (!id_out128) return VITA_ERR_INVALID_ARG;
  if (!rccl().ok) return VITA_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != 0) return VITA_ERR_LAUNCH;
  memcpy(id_out128, id.internal, 128);
  return VITA_OK;
}

extern "C" int vita_cp_init(vita_cp_context** out, int cp_size, int cp_rank, const void* unique_id128) {
  if (!out || cp_size < 1 || cp_size > kMaxCP || cp_rank < 0 || cp_rank >= cp_size) return VITA_ERR_INVALID_ARG;
  if (unique_id128 && !rccl().ok) return VITA_ERR_UNSUPPORTED;
  vita_cp_context* c = (vita_cp_context*)calloc(1, sizeof(vita_cp_context));
  if (!c) return VITA_ERR_LAUNCH;
  if (unique_id128) {
    ncclUniqueId id;
    memcpy(id.internal, unique_id128, 128);
    if (rccl().CommInitRank(&c->comm, cp_size, id, cp_rank) != 0) { free(c); return VITA_ERR_LAUNCH; }
  }      // else: EXTERNAL exchange — the host moves the shards itself (its own transport, or one process simulating the ranks)
  c->cp_size = cp_size; c->cp_rank = cp_rank;
  bool ok = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->reduced, hipEventDisableTiming) == hipSuccess;
  for (int j = 0; j < kMaxSplit && ok; ++j) ok = hipEventCreateWithFlags(&c->gathered[j], hipEventDisableTiming) == hipSuccess;
  if (!ok) { if (c->comm) rccl().CommDestroy(c->comm); free(c); return VITA_ERR_LAUNCH; }
  *out = c;
  return VITA_OK;
}

extern "C" int vita_cp_destroy(vita_cp_context* c) {
  if (!c) return VITA_ERR_INVALID_ARG;
  (void)hipStreamSynchronize(c->comm_stream);
  for (int j = 0; j < kMaxSplit; ++j) (void)hipEventDestroy(c->gathered[j]);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->reduced);
  (void)hipStreamDestroy(c->comm_stream);
  if (c->comm) rccl().CommDestroy(c->comm);
  free(c);
  return VITA_OK;
}

extern "C" size_t vita_cp_attn_workspace_bytes(int cp_size, int64_t s_local, int n_kv_heads, int head_dim) {
  return (size_t)cp_size * 2 * (size_t)s_local * n_kv_heads * head_dim * 2;            // the gathered K/V of one layer, bf16
}

// one split's second output partial (bf16 [s_local][n_q_heads / n_split][head_dim]) + the lse of both partials (fp32)
extern "C" size_t vita_cp_attn_scratch_bytes(int64_t s_local, int n_q_heads, int n_split, int head_dim) {
  if (s_local <= 0 || n_q_heads <= 0 || n_split <= 0 || n_q_heads % n_split) return 0;
  const size_t hj = (size_t)(n_q_heads / n_split);
  const size_t o_b = (((size_t)s_local * hj * head_dim * 2) + 255) & ~(size_t)255;
  return o_b + 2 * hj * (size_t)s_local * sizeof(float);
}

namespace {

struct Geo {
  int32_t q_gid[2], kv_gid[2 * kMaxCP];
  int64_t kv_row[2 * kMaxCP];
};

Geo make_geo(int cp, int r, int64_t s_local) {
  Geo g;
  const int64_t c = s_local / 2;
  g.q_gid[0] = r; g.q_gid[1] = 2 * cp - 1 - r;
  for (int p = 0; p < cp; ++p) {
    g.kv_gid[2 * p] = p; g.kv_gid[2 * p + 1] = 2 * cp - 1 - p;
    g.kv_row[2 * p] = (int64_t)p * 2 * s_local; g.kv_row[2 * p + 1] = (int64_t)p * 2 * s_local + c;
  }
  return g;
}

int check(const vita_cp_attn_params* p, const vita_cp_context* c) {
  if (!p || !c || !p->q || !p->kv_packed || !p->workspace) return VITA_ERR_INVALID_ARG;
  if (p->n_split < 1 || p->n_split > kMaxSplit || p->n_kv_heads % p->n_split || p->n_q_heads % p->n_kv_heads) return VITA_ERR_INVALID_ARG;
  if (p->s_local <= 0 || (p->s_local & 1) || p->head_dim != 128) return VITA_ERR_UNSUPPORTED;
  if (p->workspace_bytes < vita_cp_attn_workspace_bytes(c->cp_size, p->s_local, p->n_kv_heads, p->head_dim)) return VITA_ERR_INVALID_ARG;
  return VITA_OK;
}

}  // namespace

extern "C" int vita_cp_attn_fwd(vita_cp_context* c, const vita_cp_attn_params* p, void* stream) {
  int rc = check(p, c);
  if (rc != VITA_OK) return rc;
  if (!p->out) return VITA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int cp = c->cp_size, hg = p->n_kv_heads / p->n_split, G = p->n_q_heads / p->n_kv_heads, d = p->head_dim;
  const int64_t s_l = p->s_local;
  const size_t shard = (size_t)2 * s_l * hg * d;                                          // elements per rank and split
  const bf16_t* kv = (const bf16_t*)p->kv_packed;
  bf16_t* ws = (bf16_t*)p->workspace;
  const bool exchange = c->comm != nullptr;          // external exchange: the caller has filled `workspace`, ordered before `stream`
  if (exchange) {
    if (hipEventRecord(c->ready, st) != hipSuccess || hipStreamWaitEvent(c->comm_stream, c->ready, 0) != hipSuccess) return VITA_ERR_LAUNCH;
    for (int j = 0; j < p->n_split; ++j) {
      if (rccl().AllGather(kv + j * shard, ws + (size_t)j * cp * shard, shard, kNcclBfloat16, c->comm, c->comm_stream) != 0) return VITA_ERR_LAUNCH;
      if (hipEventRecord(c->gathered[j], c->comm_stream) != hipSuccess) return VITA_ERR_LAUNCH;
    }
  }
  const Geo g = make_geo(cp, c->cp_rank, s_l);
  const int hj = hg * G;                                                                    // query heads of one split
  const bool own_first = cp > 1 && p->scratch && p->scratch_bytes >= vita_cp_attn_scratch_bytes(s_l, p->n_q_heads, p->n_split, d);
  for (int j = 0; j < p->n_split; ++j) {
    const bf16_t* rows = ws + (size_t)j * cp * shard;                                    // [cp][2][s_l][hg][d]: K of rank r at r*2*s_l rows, V at + s_l
    vita_attn_params a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)p->q + (int64_t)j * hg * p->q_group_stride;
    a.q_row_stride = p->q_row_stride; a.q_head_stride = p->q_head_stride; a.q_group_stride = p->q_group_stride;
    a.k = rows; a.k_row_stride = (int64_t)hg * d; a.k_head_stride = d;
    a.v = rows + (size_t)s_l * hg * d; a.v_row_stride = (int64_t)hg * d; a.v_head_stride = d;
    a.o = (bf16_t*)p->out + (int64_t)j * hj * p->out_head_stride;
    a.o_row_stride = p->out_row_stride; a.o_head_stride = p->out_head_stride;
    a.lse = p->lse ? p->lse + (int64_t)j * hj * s_l : nullptr;
    a.batch = 1; a.n_q_heads = hj; a.n_kv_heads = hg; a.head_dim = d;
    a.chunk_len = s_l / 2; a.q_valid = a.kv_valid = s_l / 2;
    a.n_q_chunks = 2; a.n_kv_chunks = 2 * cp;
    a.q_chunk_gid = g.q_gid; a.kv_chunk_gid = g.kv_gid; a.kv_chunk_row = g.kv_row;
    a.causal = 1; a.softmax_scale = p->softmax_scale;
    if (j == 0 && own_first) {
      // (1) the rank's own two chunks, straight from its packed shard — no remote byte needed, gather 0 is still in flight
      char* sc = (char*)p->scratch;
      const size_t o_b_bytes = (((size_t)s_l * hj * d * 2) + 255) & ~(size_t)255;
      bf16_t* o_b = (bf16_t*)sc;
      float* lse_b = (float*)(sc + o_b_bytes);
      float* lse_a = p->lse ? p->lse : lse_b + (size_t)hj * s_l;                          // split 0's slice of the caller's lse, or scratch
      const int64_t own_row[2] = {0, s_l / 2};
      vita_attn_params own = a;
      own.k = kv; own.v = kv + (size_t)s_l * hg * d;
      own.n_kv_chunks = 2; own.kv_chunk_gid = g.q_gid; own.kv_chunk_row = own_row; own.lse = lse_a;
      rc = vita_flash_attn_fwd(&own, stream);
      if (rc != VITA_OK) return rc;
      // (2) the 2 CP - 2 remote chunks once gather 0 has landed, into the scratch partial
      if (exchange && hipStreamWaitEvent(st, c->gathered[0], 0) != hipSuccess) return VITA_ERR_LAUNCH;
      int32_t rem_gid[2 * kMaxCP];
      int64_t rem_row[2 * kMaxCP];
      int n_rem = 0;
      for (int i = 0; i < 2 * cp; ++i)
        if (i / 2 != c->cp_rank) { rem_gid[n_rem] = g.kv_gid[i]; rem_row[n_rem] = g.kv_row[i]; ++n_rem; }
      vita_attn_params rem = a;
      rem.o = o_b; rem.o_row_stride = (int64_t)hj * d; rem.o_head_stride = d; rem.lse = lse_b;
      rem.n_kv_chunks = n_rem; rem.kv_chunk_gid = rem_gid; rem.kv_chunk_row = rem_row;
      rc = vita_flash_attn_fwd(&rem, stream);
      if (rc != VITA_OK) return rc;
      // (3) O = O_a e^(lse_a - lse) + O_b e^(lse_b - lse), lse_a <- lse
      rc = vita_attn_merge(a.o, a.o_row_stride, a.o_head_stride, lse_a, o_b, (int64_t)hj * d, d, lse_b, s_l, hj, d, stream);
      if (rc != VITA_OK) return rc;
      continue;
    }
    if (exchange && hipStreamWaitEvent(st, c->gathered[j], 0) != hipSuccess) return VITA_ERR_LAUNCH;
    rc = vita_flash_attn_fwd(&a, stream);
    if (rc != VITA_OK) return rc;
  }
  return VITA_OK;
}

extern "C" int vita_cp_attn_bwd(vita_cp_context* c, const vita_cp_attn_params* p, const void* d_out, const float* lse,
                                const float* delta, void* dq, void* dkv_packed, void* stream) {
  int rc = check(p, c);
  if (rc != VITA_OK) return rc;
  if (!d_out || !lse || !delta || !dq || (c->comm && !dkv_packed) || !p->dkv_workspace) return VITA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int cp = c->cp_size, hg = p->n_kv_heads / p->n_split, G = p->n_q_heads / p->n_kv_heads, d = p->head_dim;
  const int64_t s_l = p->s_local;
  const size_t shard = (size_t)2 * s_l * hg * d;
  const bf16_t* ws = (const bf16_t*)p->workspace;                                        // gathered K/V of the forward (or its recompute)
  bf16_t* dws = (bf16_t*)p->dkv_workspace;                                               // dK/dV in the gathered layout, same size
  const Geo g = make_geo(cp, c->cp_rank, s_l);
  for (int j = 0; j < p->n_split; ++j) {
    const bf16_t* rows = ws + (size_t)j * cp * shard;
    bf16_t* drows = dws + (size_t)j * cp * shard;
    vita_attn_bwd_params b;
    memset(&b, 0, sizeof(b));
    b.q = (const bf16_t*)p->q + (int64_t)j * hg * p->q_group_stride;
    b.q_row_stride = p->q_row_stride; b.q_head_stride = p->q_head_stride; b.q_group_stride = p->q_group_stride;
    b.k = rows; b.k_row_stride = (int64_t)hg * d; b.k_head_stride = d;
    b.v = rows + (size_t)s_l * hg * d; b.v_row_stride = (int64_t)hg * d; b.v_head_stride = d;
    b.d_o = (const bf16_t*)d_out + (int64_t)j * hg * G * p->out_head_stride;
    b.do_row_stride = p->out_row_stride; b.do_head_stride = p->out_head_stride;
    b.lse = lse + (int64_t)j * hg * G * s_l; b.delta = delta + (int64_t)j * hg * G * s_l;
    b.dq = (bf16_t*)dq + (int64_t)j * hg * p->q_group_stride;
    b.dq_row_stride = p->q_row_stride; b.dq_head_stride = p->q_head_stride; b.dq_group_stride = p->q_group_stride;
    b.dk = drows; b.dk_row_stride = (int64_t)hg * d; b.dk_head_stride = d;
    b.dv = drows + (size_t)s_l * hg * d; b.dv_row_stride = (int64_t)hg * d; b.dv_head_stride = d;
    b.n_q_heads = hg * G; b.n_kv_heads = hg; b.head_dim = d;
    b.chunk_len = s_l / 2; b.n_q_chunks = 2; b.n_kv_chunks = 2 * cp;
    b.q_chunk_gid = g.q_gid; b.kv_chunk_gid = g.kv_gid; b.kv_chunk_row = g.kv_row;
    b.softmax_scale = p->softmax_scale;
    // r04: dK / dV of split j first; its reduce-scatter goes out on the communication stream at once and runs under the dQ pass of
    // this split and under everything of the later splits (through r03 every reduce-scatter waited for the last split's last kernel)
    rc = vita_flash_attn_bwd_parts(&b, VITA_ATTN_BWD_DKV, stream);
    if (rc != VITA_OK) return rc;
    if (c->comm) {
      if (hipEventRecord(c->gathered[j], st) != hipSuccess || hipStreamWaitEvent(c->comm_stream, c->gathered[j], 0) != hipSuccess)
        return VITA_ERR_LAUNCH;
      if (rccl().ReduceScatter(drows, (bf16_t*)dkv_packed + j * shard, shard, kNcclBfloat16, kNcclSum, c->comm, c->comm_stream) != 0)
        return VITA_ERR_LAUNCH;
    }
    rc = vita_flash_attn_bwd_parts(&b, VITA_ATTN_BWD_DQ, stream);
    if (rc != VITA_OK) return rc;
  }
  if (!c->comm) return VITA_OK;        // external exchange: the caller reduces p->dkv_workspace ([split][rank][dK | dV][s_l][hg][d]) itself
  if (hipEventRecord(c->reduced, c->comm_stream) != hipSuccess || hipStreamWaitEvent(st, c->reduced, 0) != hipSuccess) return VITA_ERR_LAUNCH;
  return VITA_OK;
}
