"""Core attention — mirror of M/core/transformer/dot_product_attention.py
(`DotProductAttention.forward(query, key, value, attention_mask, attn_mask_type, packed_seq_params)`).

ViT branch  (:312-329)  non-causal, flash_attn_func            -> vita_flash_attn_fwd, d = 64
LLM CP = 1  (:374-390)  causal, _flash_attention_forward        -> vita_flash_attn_fwd, d = 128
LLM CP > 1  (TE AttnFuncWithCP P2P ring, gpt_layer_specs.py:40) -> ONE all-gather of the packed
            K/V shard over the CP group (RCCL; every peer pushes over its own xGMI link) followed by
            one kernel launch over the zig-zag chunk tables.  The gathered buffer keeps rank order,
            so chunk 2p+h of the buffer is global chunk (h ? 2CP-1-p : p).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops, parallel_state as mpu, recompute_cache, training_utils

HEAD_SIZES = (64, 96, 128)         # head sizes the attention kernels are instantiated for (attn.hip, attn_bwd.hip)


class HipDotProductAttention(torch.nn.Module):
    """`core_attention` submodule for the layer specs (gpt_layer_specs.py): Megatron's constructor
    `(config, layer_number, attn_mask_type, attention_type, attention_dropout=None)`, forward
    `(query, key, value, attention_mask, attn_mask_type=None, packed_seq_params=None)` -> [sq, b, hp]
    (M/core/transformer/dot_product_attention.py:153,165,285-289).  No parameters; autograd through FlashAttnFn."""

    def __init__(self, config, layer_number: int, attn_mask_type, attention_type: str = "self", attention_dropout: float = None):
        super().__init__()
        self.config, self.layer_number = config, max(1, layer_number)
        self.attn_mask_type, self.attention_type = attn_mask_type, attention_type
        tp = mpu.get_tensor_model_parallel_world_size()
        kv = getattr(config, "num_query_groups", None) or config.num_attention_heads
        self.num_attention_heads_per_partition = config.num_attention_heads // tp
        self.num_query_groups_per_partition = kv // tp
        self.hidden_size_per_attention_head = config.kv_channels
        p = config.attention_dropout if attention_dropout is None else attention_dropout
        if p and p > 0.0:
            raise NotImplementedError("attention dropout > 0 is not on the Long-VITA path (--attention-dropout 0.0)")
        causal = "causal" in str(attn_mask_type).lower()
        self._impl = DotProductAttention(self.num_attention_heads_per_partition, self.num_query_groups_per_partition,
                                         self.hidden_size_per_attention_head, causal=causal)

    def forward(self, query, key, value, attention_mask=None, attn_mask_type=None, packed_seq_params=None):
        needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
        region, phase = recompute_cache.current()
        if not needs_grad:
            if region is not None and phase == "store" and self._impl.causal and query.shape[1] == 1 and packed_seq_params is None:
                return self._forward_and_keep(query, key, value)
            if region is not None and phase == "replay" and self._impl.causal and query.shape[1] == 1 and packed_seq_params is None:
                # a replayed call that needs no gradient (frozen layer, input without grad) still owns the slot its store run
                # filled: consume it — and use it, the context is the result asked for (skipping it shifted every
                # later layer's slot)
                kept = recompute_cache.take(like=query.transpose(0, 1))
                if kept is not None:
                    sq, b, np_, hn = query.shape
                    return kept[0].transpose(0, 1).reshape(sq, b, np_ * hn)
            return self._impl.forward(query, key, value, attention_mask, attn_mask_type, packed_seq_params)
        assert packed_seq_params is None, (
            "Packed sequence is not supported by DotProductAttention."
            "Please use TEDotProductAttention instead.")
        from .autograd_fns import FlashAttnCPFn, FlashAttnFn, FlashAttnNonCausalFn
        sq, b, np_, hn = query.shape
        if not self._impl.causal:                      # the ViT layers (AttnMaskType.no_mask): batch = frames
            scale = self._impl.softmax_scale if self._impl.softmax_scale is not None else 1.0 / hn ** 0.5
            q, k, v = (_pad_head_dim(t.transpose(0, 1)) for t in (query, key, value))      # SigLIP: 72 -> 96 zero columns
            out = FlashAttnNonCausalFn.apply(q, k, v, scale)[..., :hn]
            return out.transpose(0, 1).reshape(sq, b, np_ * hn)
        if b != 1:
            # vita_flash_attn_bwd is a batch-1 kernel (every Long-VITA script trains --micro-batch-size 1)
            raise ValueError("the autograd path of HipDotProductAttention runs micro-batch 1 "
                             f"(got batch {b}); split the batch or run under torch.no_grad()")
        q, k, v = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        if self._impl.causal and mpu.get_context_parallel_world_size() > 1:
            # TE's AttnFuncWithCP behind M/core/models/gpt/gpt_layer_specs.py:40: K / V all-gather forward,
            # dK / dV reduce-scatter backward
            if training_utils.get_packed_segments() is not None:
                raise NotImplementedError("packed samples under context parallelism are not built (reference stage 2 is CP = 1)")
            out = FlashAttnCPFn.apply(q, k, v, self._impl, recompute_cache.take(like=q))
        else:
            seg = training_utils.get_packed_segments() if self._impl.causal else None
            out = FlashAttnFn.apply(q, k, v, self._impl.softmax_scale, self._impl.causal,
                                    None if seg is None else seg[0], None if seg is None else seg[1], recompute_cache.take(like=q))
        return out.transpose(0, 1).reshape(sq, b, np_ * hn)

    def _forward_and_keep(self, query, key, value):
        """The first (no-grad) run of a checkpointed layer under VITA_KEEP_ATTENTION=1: the same kernels as the plain call, with the
        log-sum-exp written out, and (context, lse) left with recompute_cache for the replay in the backward."""
        from .autograd_fns import FlashAttnCPFn
        sq, b, np_, hn = query.shape
        q, k, v = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        if mpu.get_context_parallel_world_size() > 1:
            if training_utils.get_packed_segments() is not None:
                raise NotImplementedError("packed samples under context parallelism are not built (reference stage 2 is CP = 1)")
            o, lse = FlashAttnCPFn.run_forward(q, k, v, self._impl)
        else:
            seg = training_utils.get_packed_segments()
            o, lse = ops.flash_attn(q, k, v, causal=True, softmax_scale=self._impl.softmax_scale, return_lse=True,
                                    seg_start=None if seg is None else seg[0])
        recompute_cache.store((o, lse))
        return o.transpose(0, 1).reshape(sq, b, np_ * hn)


def _pad_head_dim(t: torch.Tensor) -> torch.Tensor:
    """[..., hn] -> [..., 64 | 96 | 128] with zero columns when hn is none of them (SigLIP-400M: kv_channels = 72 -> 96,
    M/pretrain_long_vita.py:276; through r04: -> 128): zero columns of Q / K add nothing to a score, zero columns of V give zero output
    columns, which the caller drops; the softmax scale stays 1 / sqrt(hn).  A no-op for the sizes the kernels tile."""
    hn = t.shape[-1]
    if hn in HEAD_SIZES:
        return t
    if hn > 128:
        raise NotImplementedError(f"head size {hn}: the attention kernels are built for 64, 96 and 128 (smaller sizes are zero-padded)")
    return torch.nn.functional.pad(t, (0, min(d for d in HEAD_SIZES if d > hn) - hn))


class DotProductAttention:
    def __init__(self, num_attention_heads: int, num_query_groups: int, kv_channels: int, causal: bool = True,
                 softmax_scale: Optional[float] = None):
        self.np, self.ng, self.hn = num_attention_heads, num_query_groups, kv_channels
        self.causal = causal
        self.softmax_scale = softmax_scale
        self._kv_gather = None
        self._o_remote = None
        # own chunks first, remote chunks after gather 0 has landed (VITA_CP_LOCAL_FIRST=0: everything waits for gather 0)
        self.local_first = bool(int(os.environ.get("VITA_CP_LOCAL_FIRST", "1")))
        self._streams = []
        self.split_streams = bool(int(os.environ.get("VITA_CP_STREAMS", "1")))
        # bench.py (N > 1): a list that receives, per K / V gather, {"bytes": message bytes this rank SENT, "wait": (event, event)} — the two
        # HIP events bracket the point where the attention's stream waits for the gather, so their distance is the time the compute
        # stream actually idled for the exchange (0 when the gather had landed under the previous launch)
        self.comm_log = None

    def _wait_gather(self, work, nbytes: int):
        if work is None:
            return
        if self.comm_log is None:
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        work.wait()
        e1.record()
        self.comm_log.append({"bytes": nbytes, "wait": (e0, e1)})

    # -- Megatron calling convention: [s, b, heads, d] ---------------------------------------------
    def forward(self, query, key, value, attention_mask=None, attn_mask_type=None, packed_seq_params=None):
        assert packed_seq_params is None, (
            "Packed sequence is not supported by DotProductAttention."
            "Please use TEDotProductAttention instead.")                               # :156-159
        sq, b, np_, hn = query.shape
        q = query.transpose(0, 1)
        k = key.transpose(0, 1)
        v = value.transpose(0, 1)
        cp = mpu.get_context_parallel_world_size()
        if self.causal and cp > 1:
            if b != 1:
                raise ValueError("context-parallel attention runs batch 1")
            kv = torch.stack([key.reshape(sq, self.ng, hn), value.reshape(sq, self.ng, hn)]).contiguous()
            out = self.forward_cp(q.reshape(1, sq, self.ng, np_ // self.ng, hn), kv)
        else:
            # position ids with resets -> packed samples (_flash_attention_forward(position_ids=...), :374-390)
            seg = training_utils.get_packed_segments() if self.causal else None
            if seg is not None and (b != 1 or cp > 1):
                raise NotImplementedError("packed samples run micro-batch 1, CP = 1 (reference stage 2)")
            if not self.causal and hn not in HEAD_SIZES:
                scale = self.softmax_scale if self.softmax_scale is not None else 1.0 / hn ** 0.5
                out = ops.flash_attn(_pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v), causal=False, softmax_scale=scale)[..., :hn]
            else:
                out = ops.flash_attn(q, k, v, causal=self.causal, softmax_scale=self.softmax_scale,
                                     seg_start=None if seg is None else seg[0])
        return out.transpose(0, 1).reshape(sq, b, np_ * hn)                           # [sq, b, hp] :285-289

    __call__ = forward

    # -- context-parallel core ----------------------------------------------------------------------
    def forward_cp(self, q5: torch.Tensor, kv_local: torch.Tensor, out: Optional[torch.Tensor] = None, events=None,
                   lse: Optional[torch.Tensor] = None):
        """q5 [1, S_l, ng, qpg, d] grouped query view; kv_local packed [kv_split, 2, S_l, ng/kv_split, d]
        (vita_rope_qkv_fwd).  One all-gather per kv-head split, all issued up front on RCCL's stream;
        the attention over split j waits only for gather j, so gather j+1 runs under it.
        lse: fp32 [1, np, S_l] (contiguous) receives the row log-sum-exp over ALL keys (what the backward needs)."""
        cp, r = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
        if kv_local.dim() == 4:
            kv_local = kv_local.unsqueeze(0)
        n_split, _, s_l, hg, d = kv_local.shape
        if s_l % 2:
            raise ValueError("local sequence must hold two zig-zag chunks")
        c = s_l // 2
        qpg = self.np // self.ng
        if q5.dim() != 5:
            q5 = q5.reshape(1, s_l, self.ng, qpg, d)
        if out is None:
            out = torch.empty((1, s_l, self.np, d), dtype=q5.dtype, device=q5.device)
        gathered = self._gather_buffer(kv_local, cp).view(n_split, cp, 2 * s_l * hg * d)
        group = mpu.get_context_parallel_group()
        works = []
        for j in range(n_split):
            if cp > 1 or dist.is_initialized():
                works.append(dist.all_gather_into_tensor(gathered[j].view(-1), kv_local[j].reshape(-1), group=group,
                                                         async_op=True))
            else:                                   # forced CP path without a process group
                gathered[j].view(-1).copy_(kv_local[j].reshape(-1))
                works.append(None)
        kv_gid, kv_row = [], []
        for p in range(cp):
            kv_gid += [p, 2 * cp - 1 - p]
            kv_row += [p * 2 * s_l, p * 2 * s_l + c]
        if events:
            events[0].record()
        # The per-split launches are a quarter of the heads each (640 workgroups at 128K / CP = 8 for 256 CUs): run them on
        # separate HIP streams so that the workgroups of split j+1 fill the CUs that split j's tail leaves idle.
        main = torch.cuda.current_stream()
        side = self._side_streams(n_split - 1, q5.device) if (self.split_streams and n_split > 1) else []
        ready = None
        if side:
            ready = torch.cuda.Event()
            ready.record(main)                                      # q5 rotated, kv packed, gathers issued
        done = []
        for j in range(n_split):
            stream = main if (j == 0 or not side) else side[j - 1]
            with torch.cuda.stream(stream):
                if stream is not main:
                    stream.wait_event(ready)
                rows = gathered[j].view(cp * 2 * s_l, hg, d)        # K rows of rank p at p*2*s_l, V at +s_l
                qj, oj = q5[:, :, j * hg:(j + 1) * hg], out[:, :, j * hg * qpg:(j + 1) * hg * qpg]
                lj = None if lse is None else lse[:, j * hg * qpg:(j + 1) * hg * qpg]     # batch 1: a head slice is contiguous
                own = mpu.zigzag_chunk_ids(cp, r)
                if j == 0 and cp > 1 and self.local_first:
                    # Gathers 1.. run under the attention of the split before them; gather 0 has nothing in front of it.  The rank's
                    # OWN two chunks need no remote K / V: attend to them (straight from the packed send buffer) while gather 0
                    # is in flight, then to the remote chunks, and merge the two partials (TE's ring does its
                    # local block first for the same reason).
                    _, lse_a = ops.flash_attn(qj, kv_local[0, 0].unsqueeze(0), kv_local[0, 1].unsqueeze(0), causal=True,
                                              softmax_scale=self.softmax_scale, chunk_len=c, q_chunk_gid=own, kv_chunk_gid=own,
                                              kv_chunk_row=[0, c], out=oj, return_lse=True, lse_out=lj)
                    self._wait_gather(works[j], kv_local[j].numel() * kv_local.element_size())
                    rem = [i for i in range(2 * cp) if i // 2 != r]
                    o_b = self._remote_buffer(oj)
                    _, lse_b = ops.flash_attn(qj, rows.unsqueeze(0), rows[s_l:].unsqueeze(0), causal=True,
                                              softmax_scale=self.softmax_scale, chunk_len=c, q_chunk_gid=own,
                                              kv_chunk_gid=[kv_gid[i] for i in rem], kv_chunk_row=[kv_row[i] for i in rem], out=o_b,
                                              return_lse=True)
                    ops.attn_merge_(oj, lse_a, o_b, lse_b)
                else:
                    self._wait_gather(works[j], kv_local[j].numel() * kv_local.element_size())     # this stream waits for gather j only
                    ops.flash_attn(qj, rows.unsqueeze(0), rows[s_l:].unsqueeze(0), causal=True, softmax_scale=self.softmax_scale,
                                   chunk_len=c, q_chunk_gid=own, kv_chunk_gid=kv_gid, kv_chunk_row=kv_row, out=oj, lse_out=lj)
                if stream is not main:
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    done.append(ev)
        for ev in done:
            main.wait_event(ev)                                     # the o-projection (main stream) needs every split
        if events:
            events[1].record()
        return out

    def _remote_buffer(self, like: torch.Tensor) -> torch.Tensor:
        shape = tuple(like.shape)
        if self._o_remote is None or tuple(self._o_remote.shape) != shape or self._o_remote.device != like.device:
            self._o_remote = torch.empty(shape, dtype=like.dtype, device=like.device)
        return self._o_remote

    def _side_streams(self, n: int, device):
        if len(self._streams) < n:
            self._streams += [torch.cuda.Stream(device=device) for _ in range(n - len(self._streams))]
        return self._streams[:n]

    def _gather_buffer(self, kv_local, cp):
        n = kv_local.numel() * cp
        if self._kv_gather is None or self._kv_gather.numel() != n or self._kv_gather.device != kv_local.device:
            self._kv_gather = torch.empty(n, dtype=kv_local.dtype, device=kv_local.device)
        return self._kv_gather
