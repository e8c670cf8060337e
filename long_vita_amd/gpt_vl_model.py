"""GPTVLModel — mirror of M/core/models/multimodal/gpt_vl_model.py:233-416 (forward) for the
prefill path: ViT provider -> embedding(+scatter) -> RoPE -> 48 decoder layers -> logits-masked head.

Decoder layer = Megatron TransformerLayer built from the TE spec
(M/core/models/gpt/gpt_layer_specs.py:35-49): RMSNorm -> linear_qkv(+bias) -> RoPE -> core attention
-> linear_proj -> residual;  RMSNorm -> linear_fc1 -> SwiGLU -> linear_fc2 -> residual.
Weights are in MEGATRON layout: linear_qkv rows per kv group [q x (np/ng), k, v]
(R/tools/hf2mcore_long_vita.py:597-609), linear_fc1 = cat[gate, up] (:612).

Every arithmetic step is a libvita_hip.so kernel; activations live in a per-model workspace that
is reused across the 48 layers (sized once for the local sequence length).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops, parallel_state as mpu, tracing, training_utils
from .dot_product_attention import DotProductAttention
from .language_model_embedding import LanguageModelEmbedding
from .layers import ColumnParallelLinear
from .rotary_pos_embedding import RotaryEmbedding


@dataclass
class GPTConfig:
    """stage3 .sh:155-200 / config_14B.json:31-56."""
    num_layers: int = 48
    hidden: int = 5120
    heads: int = 40
    kv_groups: int = 8
    head_dim: int = 128
    ffn: int = 13824
    vocab: int = 152064
    eps: float = 1e-6
    rope_theta: float = 1e6
    output_multiplier_scale: float = 0.0        # args.output_multiplier_scale (gpt_vl_model.py:349-350); 0 = off
    output_logit_softcapping: float = 0.0       # args.output_logit_softcapping (:352-355); 0 = off

    @property
    def qpg(self):
        return self.heads // self.kv_groups

    @property
    def qkv_out(self):
        return (self.heads + 2 * self.kv_groups) * self.head_dim


class GPTVLModel:
    def __init__(self, cfg: GPTConfig, params: dict, external_feature_model=None):
        self.cfg, self.p = cfg, params
        self.external_feature_model = external_feature_model
        self.embedding = LanguageModelEmbedding.from_weight(params["embed"])
        self.rotary_pos_emb = RotaryEmbedding(cfg.head_dim, rotary_base=cfg.rope_theta, device=params["embed"].device)
        self.core_attention = DotProductAttention(cfg.heads, cfg.kv_groups, cfg.head_dim, causal=True)
        self.output_layer = ColumnParallelLinear.from_weight(params["lm_head"], bias=None)
        self._ws = {}
        # K/V all-gather messages per layer (split by kv head; gather j+1 overlaps attention j): None = ops.cp_kv_split's rule for the
        # local sequence length (as fine as every attention launch still fills the chip for >= 5 rounds); an int fixes it
        self.kv_split = None
        self.force_cp_path = bool(int(os.environ.get("VITA_FORCE_CP", "0")))   # diagnostics only
        self.decode_fused = bool(int(os.environ.get("VITA_DECODE_FUSED", "1")))   # one C call per half layer (decode)
        self.decode_graph = bool(int(os.environ.get("VITA_DECODE_GRAPH", "0")))   # capture the token step (CP = 1)
        self.attn_events = None      # bench.py: list collecting (start, end) HIP events per attention launch

    # ---------------------------------------------------------------------------------------------
    @classmethod
    def from_oracle_layout(cls, cfg: GPTConfig, p: dict, external_feature_model=None, device="cuda"):
        def d(t):
            return t.to(device=device, dtype=torch.bfloat16).contiguous()
        q = {"embed": d(p["embed"]), "final_ln": d(p["final_ln"]), "lm_head": d(p["lm_head"]),
             "layers": [{k: d(v) for k, v in lp.items()} for lp in p["layers"]]}
        return cls(cfg, q, external_feature_model)

    @classmethod
    def random_init(cls, cfg: GPTConfig, seed: int = 1234, device="cuda", std: float = 0.02,
                    external_feature_model=None):
        """Seeded synthetic weights generated on the device (identical on every rank)."""
        g = torch.Generator(device=device).manual_seed(seed)

        def rn(*shape):
            return (torch.randn(*shape, generator=g, device=device) * std).to(torch.bfloat16)

        def ones(n):
            return torch.ones(n, dtype=torch.bfloat16, device=device)

        p = {"embed": rn(cfg.vocab, cfg.hidden), "final_ln": ones(cfg.hidden), "lm_head": rn(cfg.vocab, cfg.hidden),
             "layers": []}
        for _ in range(cfg.num_layers):
            p["layers"].append({"ln1": ones(cfg.hidden), "qkv_w": rn(cfg.qkv_out, cfg.hidden), "qkv_b": rn(cfg.qkv_out),
                                "o_w": rn(cfg.hidden, cfg.heads * cfg.head_dim), "ln2": ones(cfg.hidden),
                                "fc1_w": rn(2 * cfg.ffn, cfg.hidden), "fc2_w": rn(cfg.hidden, cfg.ffn)})
        return cls(cfg, p, external_feature_model)

    # ---------------------------------------------------------------------------------------------
    def _workspace(self, s: int, device):
        ws = self._ws.get(s)
        if ws is None:
            c = self.cfg
            e = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=device)  # noqa: E731
            n_msg = self.kv_split if self.kv_split is not None else ops.cp_kv_split(c.kv_groups, c.heads, s)
            ws = {"x": e(s, c.hidden), "qkv": e(s, c.qkv_out), "ctx": e(1, s, c.heads, c.head_dim),
                  "act": e(s, c.ffn), "kv": e(n_msg, 2, s, c.kv_groups // n_msg, c.head_dim)}
            self._ws = {k: v for k, v in self._ws.items() if k == "decode"}    # keep one prefill size only
            self._ws[s] = ws
        return ws

    def decoder_layer(self, h: torch.Tensor, lp: dict, cos, sin, ws, kv_dst: Optional[torch.Tensor] = None) -> torch.Tensor:
        """h [s, hidden] updated in place.  kv_dst [2, cap, groups, d]: this rank's cache shard for the
        layer; the rotated K and V rows of the local sequence are stored into rows [0, s)."""
        c = self.cfg
        s = h.shape[0]
        cp = mpu.get_context_parallel_world_size()
        use_cp = cp > 1 or self.force_cp_path          # force: exercise pack + all-gather + chunk tables at CP = 1
        x = ops.rmsnorm(h, lp["ln1"], c.eps, out=ws["x"])
        qkv = ops.gemm(x, lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"], out=ws["qkv"])
        ops.rope_qkv_(qkv, c.kv_groups, c.qpg, c.head_dim, cos, sin, ws["kv"] if use_cp else None, ws["kv"].shape[0])
        m5 = qkv.view(1, s, c.kv_groups, c.qpg + 2, c.head_dim)
        q5 = m5[:, :, :, : c.qpg]                                  # grouped query view, read in place
        if kv_dst is not None:
            if use_cp:       # packed send buffer [split, 2, s, groups/split, d] -> [2, s, groups, d]
                kv_dst[:, :s].view(2, s, ws["kv"].shape[0], -1, c.head_dim).copy_(ws["kv"].permute(1, 2, 0, 3, 4))
            else:
                kv_dst[0, :s].copy_(m5[0, :, :, c.qpg])
                kv_dst[1, :s].copy_(m5[0, :, :, c.qpg + 1])
        ev = None
        if self.attn_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if use_cp:
            if training_utils.get_packed_segments() is not None:
                raise NotImplementedError("packed samples under context parallelism are not built")
            ctx = self.core_attention.forward_cp(q5, ws["kv"], out=ws["ctx"], events=ev)
        else:
            if ev:
                ev[0].record()
            seg = training_utils.get_packed_segments()       # position ids with resets -> packed samples
            ctx = ops.flash_attn(q5, m5[:, :, :, c.qpg], m5[:, :, :, c.qpg + 1], causal=True, out=ws["ctx"],
                                 seg_start=None if seg is None else seg[0])
            if ev:
                ev[1].record()
        if ev:
            self.attn_events.append(ev)
        tp = mpu.get_tensor_model_parallel_world_size()
        if tp == 1:
            ops.gemm(ctx.view(s, c.heads * c.head_dim), lp["o_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        else:       # row-parallel: partial sums over the rank's heads -> bf16 all-reduce -> residual add
            self._row_parallel(ctx.view(s, c.heads * c.head_dim), lp["o_w"], h, ws["x"])
        x = ops.rmsnorm(h, lp["ln2"], c.eps, out=ws["x"])
        act = ops.gemm(x, lp["fc1_w"], ops.EPI_SWIGLU, out=ws["act"])
        if tp == 1:
            ops.gemm(act, lp["fc2_w"], ops.EPI_RESIDUAL, residual=h, out=h)
        else:
            self._row_parallel(act, lp["fc2_w"], h, ws["x"])
        return h

    @staticmethod
    def _row_parallel(x: torch.Tensor, w: torch.Tensor, h: torch.Tensor, scratch: torch.Tensor) -> torch.Tensor:
        """h += all_reduce_TP(x @ w^T)  (RowParallelLinear.forward + bias_dropout_add, M/core/tensor_parallel/layers.py)."""
        import torch.distributed as dist
        part = ops.gemm(x, w, ops.EPI_NONE, out=scratch)
        dist.all_reduce(part, group=mpu.get_tensor_model_parallel_group())
        return ops.add_(h, part)

    # ---------------------------------------------------------------------------------------------
    def forward(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None, attention_mask=None,
                decoder_input: Optional[torch.Tensor] = None, labels=None, inference_params=None,
                packed_seq_params=None, extra_block_kwargs=None, external_inputs: Optional[dict] = None,
                tokentype_ids=None, logit_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns logits [b, n_sel (or s), vocab] (labels=None branch, gpt_vl_model.py:357-370)."""
        if labels is not None:
            raise NotImplementedError("the loss / backward path is long_vita_amd.training.TrainStep")
        assert packed_seq_params is None
        ip = inference_params
        if ip is not None:
            if getattr(ip, "external_inputs", None) is not None and not ip.key_value_memory_dict:   # :261-266
                external_inputs = ip.external_inputs
            if getattr(ip, "logit_mask", None) is not None:                                          # :281-283
                logit_mask = ip.logit_mask
            if hasattr(ip, "use_kv_cache") and not ip.use_kv_cache:                                  # :285-286
                ip = None
        if ip is not None and ip.key_value_memory_dict:
            return self._decode_forward(input_ids, position_ids, ip)
        if decoder_input is None:                                                         # :252-277
            if external_inputs:
                with tracing.range("prefill: vision tower"):
                    feats = self.external_feature_model(**external_inputs)               # :267
                efd = {"features": feats}
                for k in external_inputs:
                    if "indices" in k or k == "pre_len":
                        efd[k] = external_inputs[k]
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids,
                                               external_feature_dict=efd)
            else:
                decoder_input = self.embedding(input_ids=input_ids, position_ids=position_ids)
        s, b, hdim = decoder_input.shape
        if b != 1:
            raise ValueError("the Long-VITA prefill path runs batch 1")
        h = decoder_input.view(s, hdim)
        rotary_seq_len = RotaryEmbedding.get_rotary_seq_len(s)                           # :289-293
        cos, sin = self.rotary_pos_emb(rotary_seq_len)                                   # :295
        ws = self._workspace(s, h.device)
        if ip is not None:
            self._allocate_cache(ip, s, h.device)
        for li, lp in enumerate(self.p["layers"]):                                        # self.decoder(...) :299
            with tracing.range(f"prefill: layer {li}"):
                self.decoder_layer(h, lp, cos, sin, ws, None if ip is None else ip.key_value_memory_dict[li + 1])
        if ip is not None:
            self._compact_cache(ip, s)
        tracing.push("prefill: final norm + masked head")
        # final RMSNorm is per-row, so norm only the rows the masked head keeps
        if logit_mask is not None:
            idx = ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1))
            rows = ops.row_gather(h, idx)
            sel_mask = None
        else:
            rows, sel_mask = h, None
        rows = ops.rmsnorm(rows, self.p["final_ln"], self.cfg.eps)
        logits, _ = self.output_layer(rows.view(rows.shape[0], 1, hdim), weight=None, logit_mask=sel_mask)   # :339
        logits = self._gather_vocab_parallel(logits)
        ops.logit_postprocess_(logits, self.cfg.output_multiplier_scale, self.cfg.output_logit_softcapping)   # :349-355
        if bool(torch.isnan(logits.float().sum())):                                       # :393-396
            tracing.pop()
            raise ValueError("found NaN in local forward logits calculation")
        tracing.pop()
        return logits.transpose(0, 1).contiguous()                                        # [s b v] -> [b s v] :370

    @staticmethod
    def _gather_vocab_parallel(logits: torch.Tensor) -> torch.Tensor:
        """[..., V/TP] per rank -> [..., V] (gather_from_tensor_model_parallel_region of the vocab-parallel output layer)."""
        tp = mpu.get_tensor_model_parallel_world_size()
        if tp == 1:
            return logits
        import torch.distributed as dist
        flat = torch.empty((tp,) + tuple(logits.shape), dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(flat.view(-1), logits.contiguous().view(-1), group=mpu.get_tensor_model_parallel_group())
        return flat.movedim(0, -2).reshape(*logits.shape[:-1], tp * logits.shape[-1])

    __call__ = forward

    # ---------------------------------------------------------------------------------------------
    # decode against the sharded KV cache (inference_params.py)
    # ---------------------------------------------------------------------------------------------
    def _allocate_cache(self, ip, s_local: int, device):
        c = self.cfg
        cp = mpu.get_context_parallel_world_size()
        prompt = ip.prefill_valid_tokens if ip.prefill_valid_tokens is not None else s_local * cp
        if prompt > s_local * cp:
            raise ValueError("prefill_valid_tokens exceeds the tokens fed")
        room = max(0, ip.max_sequence_length - prompt)
        cap = s_local + -(-room // cp) + 1
        buf = torch.empty(c.num_layers, 2, cap, c.kv_groups, c.head_dim, dtype=torch.bfloat16, device=device)
        ip.key_value_memory_dict = {li + 1: buf[li] for li in range(c.num_layers)}    # Megatron keys by layer_number
        ip.prefill_valid_tokens = prompt

    def _compact_cache(self, ip, s_local: int):
        """Drop the rows of padded prompt positions: keep rows whose global position < prefill_valid_tokens,
        second zig-zag chunk moved up against the first."""
        cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
        P = ip.prefill_valid_tokens
        if cp == 1:
            ip.local_len = P
        else:
            cl = s_local // 2
            va = min(max(P - r * cl, 0), cl)
            vb = min(max(P - (2 * cp - 1 - r) * cl, 0), cl)
            if vb > 0 and va < cl:
                for kv in ip.key_value_memory_dict.values():
                    kv[:, va: va + vb].copy_(kv[:, cl: cl + vb].clone())
            ip.local_len = va + vb
        ip.decode_steps = 0
        ip.consumed_tokens = P

    def _decode_workspace(self, device):
        ws = self._ws.get("decode")
        cp = mpu.get_context_parallel_world_size()
        if ws is not None and ws["gmsg"].shape[0] != cp:          # the model object outlived a change of the context-parallel size
            ws = None                                             # (r05: found by the first two-PROCESS run of the decode loop)
        if ws is None:
            c = self.cfg
            e = lambda *shape, dt=torch.bfloat16: torch.empty(*shape, dtype=dt, device=device)  # noqa: E731
            msg = c.heads * c.head_dim + 2 * c.heads
            ws = {"x": e(1, c.hidden), "qkv": e(1, c.qkv_out), "ctx": e(c.heads, c.head_dim), "act": e(c.ffn),
                  "h": e(1, c.hidden), "msg": e(msg, dt=torch.float32), "gmsg": e(cp, msg, dt=torch.float32)}
            self._ws["decode"] = ws
        return ws

    def _decode_token(self, token: torch.Tensor, position: torch.Tensor, ip, counters=None) -> torch.Tensor:
        """One token through the 48 layers: token [1] int64, position [1] int64 -> final-normed hidden [1, hidden].
        counters = (row_dev int64 [1], len_dev int32 [1]): the shard's row count lives on the device (the captured
        graph of this function is replayed as the cache grows; CP = 1 only)."""
        import torch.distributed as dist
        c = self.cfg
        cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
        tp = mpu.get_tensor_model_parallel_world_size()     # TP > 1: cfg / params / cache hold this rank's heads, groups and ffn slice
        ws = self._decode_workspace(token.device)
        h = ops.row_gather(self.p["embed"], token.reshape(1), out=ws["h"], check_bounds=False)   # [1, hidden]
        cos, sin = ops.rope_table(position.reshape(1), self.rotary_pos_emb.inv_freq)
        owner = (ip.decode_steps % cp) == r
        row = ip.local_len
        length = row + (1 if owner else 0)
        for li, lp in enumerate(self.p["layers"]):
            kv = ip.key_value_memory_dict[li + 1]
            if length > kv.shape[1]:
                raise RuntimeError("KV cache shard is full (max_sequence_length reached)")
            x = ops.rmsnorm(h, lp["ln1"], c.eps, out=ws["x"])
            qkv = ops.gemv(x.view(-1), lp["qkv_w"], ops.EPI_BIAS, lp["qkv_b"], out=ws["qkv"].view(-1))
            ops.rope_qkv_(ws["qkv"], c.kv_groups, c.qpg, c.head_dim, cos, sin, None, 1)
            m4 = qkv.view(c.kv_groups, c.qpg + 2, c.head_dim)
            if counters is not None:
                kv[0].index_copy_(0, counters[0], m4[None, :, c.qpg])
                kv[1].index_copy_(0, counters[0], m4[None, :, c.qpg + 1])
                pm, pl, po = ops.decode_attn_partial(m4[:, : c.qpg], kv[0], kv[1], kv.shape[1], len_dev=counters[1])
            else:
                if owner:
                    kv[0, row].copy_(m4[:, c.qpg])
                    kv[1, row].copy_(m4[:, c.qpg + 1])
                pm, pl, po = ops.decode_attn_partial(m4[:, : c.qpg], kv[0], kv[1], length)
            if cp == 1:
                ctx = ops.decode_attn_merge(pm, pl, po, True, out=ws["ctx"])
            else:
                ops.decode_attn_merge(pm, pl, po, False, packed_out=ws["msg"])
                dist.all_gather_into_tensor(ws["gmsg"].view(-1), ws["msg"], group=mpu.get_context_parallel_group())
                gm, gl, go = ops.unpack_partials(ws["gmsg"], c.heads, c.head_dim)
                ctx = ops.decode_attn_merge(gm, gl, go, True, out=ws["ctx"])
            if tp == 1:
                ops.gemv(ctx.view(-1), lp["o_w"], ops.EPI_RESIDUAL, residual=h.view(-1), out=h.view(-1))
            else:   # row-parallel: this rank's heads give a partial sum -> bf16 all-reduce over TP -> residual add (as in prefill)
                part = ops.gemv(ctx.view(-1), lp["o_w"], ops.EPI_NONE, out=ws["x"].view(-1))
                dist.all_reduce(part, group=mpu.get_tensor_model_parallel_group())
                ops.add_(h, part.view(1, -1))
            x = ops.rmsnorm(h, lp["ln2"], c.eps, out=ws["x"])
            act = ops.gemv(x.view(-1), lp["fc1_w"], ops.EPI_SWIGLU, out=ws["act"])
            if tp == 1:
                ops.gemv(act, lp["fc2_w"], ops.EPI_RESIDUAL, residual=h.view(-1), out=h.view(-1))
            else:
                part = ops.gemv(act, lp["fc2_w"], ops.EPI_NONE, out=ws["x"].view(-1))
                dist.all_reduce(part, group=mpu.get_tensor_model_parallel_group())
                ops.add_(h, part.view(1, -1))
        if counters is not None:
            counters[0].add_(1)
            counters[1].add_(1)
        else:
            ip.local_len = length
            ip.decode_steps += 1
        return ops.rmsnorm(h, self.p["final_ln"], c.eps)

    def _decode_token_fused(self, token: torch.Tensor, position: torch.Tensor, ip) -> torch.Tensor:
        """Same arithmetic as _decode_token, 2 C calls per layer (vita_decode_layer_attn / _mlp: 7 kernel launches
        issued from C, RMSNorm folded into the GEMVs, RoPE + cache append in one kernel) instead of ~13 launches
        from Python.  The per-layer parameter structs are built once per request."""
        import ctypes as C

        import torch.distributed as dist

        from . import lib as _L
        c = self.cfg
        cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
        ws = self._decode_workspace(token.device)
        st = getattr(ip, "_layer_structs", None)
        if st is None:
            n, H, D = ops.DECODE_MAX_SPLITS, c.heads, c.head_dim
            f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=token.device)  # noqa: E731
            extra = {"pm": f32(n, H), "pl": f32(n, H), "po": f32(n, H, D),
                     "cos": torch.empty(1, D // 2, dtype=torch.bfloat16, device=token.device),
                     "sin": torch.empty(1, D // 2, dtype=torch.bfloat16, device=token.device)}
            structs = []
            for li, lp in enumerate(self.p["layers"]):
                kv = ip.key_value_memory_dict[li + 1]
                s = _L.DecodeLayerParams()
                for name in ("ln1", "qkv_w", "qkv_b", "o_w", "ln2", "fc1_w", "fc2_w"):
                    setattr(s, name, lp[name].data_ptr())
                s.hidden, s.heads, s.kv_groups, s.head_dim, s.ffn = c.hidden, c.heads, c.kv_groups, c.head_dim, c.ffn
                s.eps, s.softmax_scale = c.eps, 1.0 / (c.head_dim ** 0.5)
                s.h = ws["h"].data_ptr()
                s.cos, s.sin = extra["cos"].data_ptr(), extra["sin"].data_ptr()
                s.k_cache, s.v_cache = kv[0].data_ptr(), kv[1].data_ptr()
                s.kv_row_stride, s.kv_group_stride, s.capacity = kv.stride(1), kv.stride(2), kv.shape[1]
                s.qkv, s.ctx, s.act = ws["qkv"].data_ptr(), ws["ctx"].data_ptr(), ws["act"].data_ptr()
                s.part_m, s.part_l, s.part_o = extra["pm"].data_ptr(), extra["pl"].data_ptr(), extra["po"].data_ptr()
                if cp > 1:
                    s.msg, s.gathered, s.n_ranks = ws["msg"].data_ptr(), ws["gmsg"].data_ptr(), cp
                structs.append(s)
            st = ip._layer_structs = (structs, extra)
        structs, extra = st
        h = ops.row_gather(self.p["embed"], token.reshape(1), out=ws["h"], check_bounds=False)
        cos, sin = ops.rope_table(position.reshape(1), self.rotary_pos_emb.inv_freq)
        extra["cos"].copy_(cos)
        extra["sin"].copy_(sin)
        owner = (ip.decode_steps % cp) == r
        row = ip.local_len
        length = row + (1 if owner else 0)
        if length > structs[0].capacity:
            raise RuntimeError("KV cache shard is full (max_sequence_length reached)")
        n_splits = ops.decode_splits(length)
        lib, stream = _L.load(), torch.cuda.current_stream().cuda_stream
        group = mpu.get_context_parallel_group() if cp > 1 else None
        for s in structs:
            s.append_row, s.len, s.n_splits = (row if owner else -1), length, n_splits
            _L.check(lib.vita_decode_layer_attn(C.byref(s), stream), "vita_decode_layer_attn")
            if cp > 1:
                dist.all_gather_into_tensor(ws["gmsg"].view(-1), ws["msg"], group=group)
            _L.check(lib.vita_decode_layer_mlp(C.byref(s), stream), "vita_decode_layer_mlp")
        ip.local_len = length
        ip.decode_steps += 1
        return ops.rmsnorm(h, self.p["final_ln"], c.eps)

    def _decode_graphed(self, token: torch.Tensor, position: torch.Tensor, ip) -> torch.Tensor:
        """CP = 1: the ~450 launches of one token step are captured once per request into a HIP graph (the
        launch-bound inner loop of decode) and replayed; token id, position and the cache row count are device
        scalars.  Returns logits [1, vocab] (a static buffer, overwritten by the next step)."""
        g = getattr(ip, "_graph", None)
        if g is None:
            dev = token.device
            st = {"tok": torch.zeros(1, dtype=torch.int64, device=dev),
                  "pos": torch.zeros(1, dtype=torch.int64, device=dev),
                  "row": torch.full((1,), ip.local_len, dtype=torch.int64, device=dev),
                  "len": torch.full((1,), ip.local_len + 1, dtype=torch.int32, device=dev)}

            def body():
                rows = self._decode_token(st["tok"], st["pos"], ip, counters=(st["row"], st["len"]))
                return self.output_layer(rows.view(1, 1, -1), weight=None, logit_mask=None)[0].view(1, -1)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up outside capture, on throw-away counters
                st["tok"].copy_(token.reshape(1)); st["pos"].copy_(position.reshape(1))
                keep = (st["row"].clone(), st["len"].clone())
                body()
                st["row"].copy_(keep[0]); st["len"].copy_(keep[1])
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["out"] = body()
            g = ip._graph = (graph, st)
        graph, st = g
        st["tok"].copy_(token.reshape(1))
        st["pos"].copy_(position.reshape(1))
        graph.replay()
        ip.local_len += 1
        ip.decode_steps += 1
        return st["out"]

    def _decode_forward(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor], ip) -> torch.Tensor:
        """tokens[:, prev:ctx] of the cached decode loop (generation.py:127-131): every CP rank is fed the same
        (unsliced) tokens and returns the same logits [1, t, vocab]."""
        b, t = input_ids.shape
        if b != 1:
            raise ValueError("the Long-VITA decode path runs batch 1")
        if position_ids is None:
            position_ids = (torch.arange(t, device=input_ids.device) + ip.sequence_len_offset)[None]
        ip.consumed_tokens = t
        tp = mpu.get_tensor_model_parallel_world_size()
        if t == 1 and self.decode_graph and mpu.get_context_parallel_world_size() == 1 and tp == 1:
            kv = ip.key_value_memory_dict[1]
            if ip.local_len + 1 > kv.shape[1]:
                raise RuntimeError("KV cache shard is full (max_sequence_length reached)")
            return self._decode_graphed(input_ids[0], position_ids[0], ip).view(1, 1, -1).clone()
        ip._graph = None                      # eager steps move the python-side counters only
        # tensor parallelism (the released server runs TP 8 x CP 4, server_cp .sh:102-104): the kernel-by-kernel step, whose
        # row-parallel GEMVs all-reduce their partial sums; the C-side fused layer adds the residual inside the GEMV epilogue
        step = self._decode_token_fused if (self.decode_fused and tp == 1) else self._decode_token
        rows = [step(input_ids[0, j: j + 1], position_ids[0, j: j + 1], ip) for j in range(t)]
        rows = rows[0] if t == 1 else torch.cat(rows, dim=0)
        logits, _ = self.output_layer(rows.view(t, 1, -1), weight=None, logit_mask=None)
        logits = self._gather_vocab_parallel(logits)
        return logits.transpose(0, 1).contiguous()
