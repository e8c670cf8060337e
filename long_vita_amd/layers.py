"""ColumnParallelLinear with `logit_mask` — mirror of M/core/tensor_parallel/layers.py:825-904
(forward) / :402-412 (masked select + GEMM) at TP=1."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class ColumnParallelLinear:
    def __init__(self, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor] = None,
                 skip_bias_add: bool = False):
        self.weight, self.bias, self.skip_bias_add = weight, bias, skip_bias_add

    def forward(self, input_: torch.Tensor, weight: Optional[torch.Tensor] = None, logit_mask=None):
        """input_ [s, b, hidden] -> (output [n_sel or s, b, out], bias_or_None)."""
        if weight is None:
            if self.weight is None:
                raise RuntimeError("weight was not supplied to ColumnParallelLinear forward pass "
                                   "and skip_weight_param_allocation is True.")
            weight = self.weight
        elif self.weight is not None and tuple(weight.shape) != tuple(self.weight.shape):
            raise RuntimeError(f"supplied weight's shape is {tuple(weight.shape)}, "
                               f"not {tuple(self.weight.shape)} as expected")
        s, b, c = input_.shape
        x = input_.reshape(s * b, c)
        if logit_mask is not None:
            # masked_select(input, logit_mask.T.unsqueeze(2)).reshape(-1, b, c)  (:402-407), b == 1
            if b != 1:
                raise AssertionError("logit_mask requires batch 1 (gpt_vl_model.py:329)")
            idx = ops.mask_to_index(logit_mask.transpose(0, 1).reshape(-1))
            x = ops.row_gather(x.contiguous(), idx)
        bias = None if self.skip_bias_add else self.bias
        m = x.shape[0]
        if m <= 16 and bias is None:
            out = ops.gemm_skinny(x, weight)
        else:
            out = ops.gemm(x, weight, ops.EPI_BIAS if bias is not None else ops.EPI_NONE, bias)
        return out.view(m // b, b, -1), (self.bias if self.skip_bias_add else None)

    __call__ = forward
