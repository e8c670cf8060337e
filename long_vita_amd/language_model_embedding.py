"""Word embedding + visual-token scatter — mirror of
M/core/models/common/embeddings/language_model_embedding.py (constructor :27-35, forward :91-174) as a
Megatron-constructible `torch.nn.Module`: `LanguageModelEmbedding(config, vocab_size, max_sequence_length,
position_embedding_type, num_tokentypes)`, parameter `word_embeddings.weight` (vocab-parallel rows, as
tensor_parallel.VocabParallelEmbedding keeps them, M/core/tensor_parallel/layers.py:155-250), autograd through
autograd_fns.EmbeddingScatterFn.  Lookup, scatter and their backward are libvita_hip.so kernels."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch.nn import Parameter

from . import autograd_fns as F_, parallel_state as mpu


class VocabParallelEmbedding(torch.nn.Module):
    """Rows [rank * V / TP, (rank + 1) * V / TP) of the table on each tensor-parallel rank; `weight` is the parameter name
    Megatron's checkpoints use (embedding.word_embeddings.weight)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, *, init_method=None, config=None, weight: Optional[torch.Tensor] = None):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        tp, r = mpu.get_tensor_model_parallel_world_size(), mpu.get_tensor_model_parallel_rank()
        if num_embeddings % tp:
            raise ValueError(f"{num_embeddings} is not divisible by {tp}")
        self.num_embeddings_per_partition = num_embeddings // tp
        self.vocab_start_index = r * self.num_embeddings_per_partition
        self.vocab_end_index = self.vocab_start_index + self.num_embeddings_per_partition
        if weight is not None:                               # stand-alone driver: wrap an existing (replicated) table
            self.weight = Parameter(weight, requires_grad=False)
            self.num_embeddings_per_partition, self.vocab_start_index, self.vocab_end_index = weight.shape[0], 0, weight.shape[0]
            return
        dtype = getattr(config, "params_dtype", torch.bfloat16)
        if getattr(config, "use_cpu_initialization", False):
            w = torch.empty(self.num_embeddings_per_partition, embedding_dim, dtype=dtype)
        else:
            w = torch.empty(self.num_embeddings_per_partition, embedding_dim, dtype=dtype, device=torch.cuda.current_device())
        self.weight = Parameter(w)
        if getattr(config, "perform_initialization", True) and init_method is not None:
            with torch.no_grad():
                init_method(self.weight)
        setattr(self.weight, "tensor_model_parallel", True)
        setattr(self.weight, "partition_dim", 0)
        setattr(self.weight, "partition_stride", 1)

    def local_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Ids relative to this rank's rows; -1 where another rank holds the row (layers.py:216-224 masks those to zero)."""
        ids = input_ids.reshape(-1)
        if self.vocab_start_index == 0 and self.vocab_end_index >= self.num_embeddings:
            return ids.contiguous()
        inside = (ids >= self.vocab_start_index) & (ids < self.vocab_end_index)
        return torch.where(inside, ids - self.vocab_start_index, torch.full_like(ids, -1)).contiguous()


class LanguageModelEmbedding(torch.nn.Module):
    """RoPE models add no position embedding (position_embedding_type='rope',
    M/core/models/multimodal/gpt_vl_model.py:110-118); `learned_absolute` and token types are not on the Long-VITA path."""

    def __init__(self, config, vocab_size: int, max_sequence_length: int, position_embedding_type: str = "learned_absolute",
                 num_tokentypes: int = 0, parallel_word_embedding: bool = True):
        super().__init__()
        if position_embedding_type == "learned_absolute" or num_tokentypes > 0:
            raise NotImplementedError("learned position / token-type embeddings are not on the Long-VITA path "
                                      "(every reference script passes --position-embedding-type rope)")
        self.config, self.vocab_size, self.max_sequence_length = config, vocab_size, max_sequence_length
        self.add_position_embedding = False
        self.num_tokentypes = num_tokentypes
        self.tokentype_embeddings = None
        self.word_embeddings = VocabParallelEmbedding(num_embeddings=vocab_size, embedding_dim=config.hidden_size,
                                                      init_method=getattr(config, "init_method", None), config=config)
        self.embedding_dropout = torch.nn.Dropout(getattr(config, "hidden_dropout", 0.0))

    @classmethod
    def from_weight(cls, weight: torch.Tensor):
        """Stand-alone driver (gpt_vl_model.GPTVLModel): wrap an existing replicated table."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.config, self.vocab_size, self.max_sequence_length = None, weight.shape[0], None
        self.add_position_embedding, self.num_tokentypes, self.tokentype_embeddings = False, 0, None
        self.word_embeddings = VocabParallelEmbedding(weight.shape[0], weight.shape[1], weight=weight)
        self.embedding_dropout = torch.nn.Dropout(0.0)
        return self

    @property
    def weight(self) -> torch.Tensor:
        return self.word_embeddings.weight

    def forward(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None, tokentype_ids=None,
                external_feature_dict: Optional[dict] = None) -> torch.Tensor:
        if tokentype_ids is not None:
            raise AssertionError("tokentype embeddings are not built (language_model_embedding.py:150)")
        b, s = input_ids.shape
        we_mod = self.word_embeddings
        h = we_mod.weight.shape[1]
        feats = tgt = src = None
        if external_feature_dict is not None:
            d = external_feature_dict
            ok = "features" in d and (len(d) == 1 or (len(d) == 2 and ("pre_len" in d or "indices" in d))
                                      or (len(d) == 3 and "src_indices" in d and "tgt_indices" in d))
            assert ok, "The format of external_feature_dict is not right!"          # :104-108
            f3 = d["features"]
            L = f3.shape[1]
            if "indices" in d:                                                      # :119-123
                ib, is_ = d["indices"].unbind(dim=0)
                tgt = (ib.reshape(-1) * s + is_.reshape(-1)).contiguous()
            elif "pre_len" in d:                                                    # :124-126
                pre = int(d["pre_len"])
                nb = f3.shape[0]
                tgt = (torch.arange(nb, device=f3.device)[:, None] * s + pre + torch.arange(L, device=f3.device)[None, :]).reshape(-1)
            elif "src_indices" in d:                                                # :128-131
                sb, ss = d["src_indices"]
                tb, ts = d["tgt_indices"]
                tgt, src = (tb * s + ts).contiguous(), (sb * L + ss).contiguous()
            # else: "+= features.mean() * 0" (:132-134) changes nothing in the forward pass
            if tgt is not None:
                feats = f3.reshape(-1, f3.shape[-1])
        ids = we_mod.local_ids(input_ids)
        tp = mpu.get_tensor_model_parallel_world_size()
        if tp > 1 and we_mod.num_embeddings_per_partition != we_mod.num_embeddings:
            # vocab-parallel lookup: every rank contributes its rows, the sum is the embedding (layers.py:216-232); the
            # visual rows are scattered after the reduction so they are not summed TP times
            we = F_.EmbeddingScatterFn.apply(we_mod.weight, ids, None, None, None)
            we = F_.ReduceFromTP.apply(we)
            if feats is not None:
                we = ScatterRowsFn.apply(we, feats, tgt, src)
        else:
            we = F_.EmbeddingScatterFn.apply(we_mod.weight, ids, feats, tgt, src)
        out = we.view(b, s, h).transpose(0, 1).contiguous()                            # [b s h] -> [s b h] (:143)
        cfg = self.config
        if cfg is not None and getattr(cfg, "sequence_parallel", False) and tp > 1:    # :157-166: this rank's sequence shard;
            out = F_.ScatterToSP.apply(out)                                            # backward = all-gather along the sequence
        if self.training and self.embedding_dropout.p > 0:
            out = self.embedding_dropout(out)
        return out


class ScatterRowsFn(torch.autograd.Function):
    """we[tgt] = feats[src] on a copy (the reference clones before the in-place scatter, :109)."""

    @staticmethod
    def forward(ctx, we, feats, tgt, src):
        from . import ops
        out = we.contiguous().clone()
        ops.row_scatter_(out, tgt, feats.contiguous(), src)
        ctx.save_for_backward(tgt, src)
        ctx.fshape = tuple(feats.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        tgt, src = ctx.saved_tensors
        g = g.contiguous()
        d_feats = torch.zeros(ctx.fshape, dtype=g.dtype, device=g.device)
        rows = ops.row_gather(g, tgt)
        ops.row_scatter_(d_feats, src if src is not None else torch.arange(tgt.numel(), device=g.device), rows)
        d_we = g.clone()
        d_we[tgt] = 0                                   # overwritten rows carry no gradient to the table
        return d_we, d_feats, None, None
