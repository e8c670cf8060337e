"""Word embedding + visual-token scatter — mirror of
M/core/models/common/embeddings/language_model_embedding.py:91-174 (forward)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class LanguageModelEmbedding:
    """word_embeddings weight [vocab, hidden] bf16; RoPE models add no position embedding
    (position_embedding_type='rope', M/core/models/multimodal/gpt_vl_model.py:110-118)."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight

    def forward(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None, tokentype_ids=None,
                external_feature_dict: Optional[dict] = None) -> torch.Tensor:
        if tokentype_ids is not None:
            raise AssertionError("tokentype embeddings are not built (language_model_embedding.py:150)")
        b, s = input_ids.shape
        h = self.weight.shape[1]
        # VocabParallelEmbedding lookup (TP=1) — M/core/tensor_parallel/layers.py:216-232
        we = ops.row_gather(self.weight, input_ids.reshape(-1)).view(b * s, h)
        if external_feature_dict is not None:
            d = external_feature_dict
            ok = "features" in d and (len(d) == 1 or (len(d) == 2 and ("pre_len" in d or "indices" in d))
                                      or (len(d) == 3 and "src_indices" in d and "tgt_indices" in d))
            assert ok, "The format of external_feature_dict is not right!"          # :104-108
            feats = d["features"]
            f2 = feats.reshape(-1, feats.shape[-1])
            L = feats.shape[1]
            if "indices" in d:                                                      # :119-123
                ib, is_ = d["indices"].unbind(dim=0)
                ops.row_scatter_(we, (ib.reshape(-1) * s + is_.reshape(-1)), f2)
            elif "pre_len" in d:                                                    # :124-126
                pre = int(d["pre_len"])
                nb = feats.shape[0]
                tgt = (torch.arange(nb, device=we.device)[:, None] * s + pre
                       + torch.arange(L, device=we.device)[None, :]).reshape(-1)
                ops.row_scatter_(we, tgt, f2)
            elif "src_indices" in d:                                                # :128-131
                sb, ss = d["src_indices"]
                tb, ts = d["tgt_indices"]
                ops.row_scatter_(we, tb * s + ts, f2, sb * L + ss)
            # else: "+= features.mean() * 0" (:132-134) changes nothing in the forward pass
        # [b s h] -> [s b h] (:143); b == 1 on this path makes it a view
        return we.view(b, s, h).transpose(0, 1).contiguous()

    __call__ = forward
