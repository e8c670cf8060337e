"""RoPE host side — mirror of M/core/models/common/embeddings/rotary_pos_embedding.py.

The reference materialises `emb = cat(freqs, freqs)` [s, b, 1, dim] in fp32 and every layer
recomputes cos/sin from it (:200-201).  Here the table is built ONCE per forward on the device as
bf16 cos/sin [s, dim/2] (vita_rope_table) for the rank's *global* positions, which folds the
position_ids gather (:114-117) and the zig-zag CP slice (:36-47, :119-121) into an index choice.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import ops, parallel_state as mpu
from .training_utils import get_position_ids, zigzag_slice


class RotaryEmbedding:
    def __init__(self, kv_channels: int, rotary_percent: float = 1.0, rotary_interleaved: bool = False,
                 seq_len_interpolation_factor: Optional[float] = None, rotary_base: float = 10000, device="cuda"):
        if rotary_interleaved or seq_len_interpolation_factor is not None or rotary_percent != 1.0:
            raise NotImplementedError("only full, non-interleaved RoPE is on the Long-VITA path")
        self.dim = kv_channels
        self.inv_freq = ops.rope_inv_freq(kv_channels, rotary_base, device)       # :74-80

    def positions(self, max_seq_len: int, offset: int = 0) -> torch.Tensor:
        """Global positions of this rank's rows, int64 [s_local]."""
        pid = get_position_ids()
        dev = self.inv_freq.device
        if pid is not None:                                                      # [s, b] (b == 1)
            pos = pid.reshape(pid.shape[0], -1)[:, 0].to(dev)
        else:
            pos = torch.arange(max_seq_len, dtype=torch.int64, device=dev) + offset
        cp = mpu.get_context_parallel_world_size()
        if cp > 1:
            pos = zigzag_slice(pos, cp, mpu.get_context_parallel_rank(), seq_dim=0)
        return pos.contiguous()

    def forward(self, max_seq_len: int, offset: int = 0):
        """-> (cos, sin) bf16 [s_local, dim/2]."""
        return ops.rope_table(self.positions(max_seq_len, offset), self.inv_freq)

    __call__ = forward

    @staticmethod
    def get_rotary_seq_len(local_rows: int) -> int:
        """:124-156 — rows on this rank x context_parallel_size."""
        return local_rows * mpu.get_context_parallel_world_size()


_COS_SIN_CACHE = {"entry": None}              # (weakref to the angle tensor, its version, (cos, sin)): ONE object, replaced whole


def _cos_sin(freqs):
    """`freqs` is either this module's (cos, sin) pair or Megatron's fp32 angle tensor [s, 1, 1, dim]
    (rotary_pos_embedding.py:106-108), which every one of the 48 layers passes again: the bf16 tables are built once per
    tensor — keyed by the tensor OBJECT (a weak reference: a freed tensor whose address is re-used by another one never hits)
    and its version counter.  The entry is read once and replaced as a whole, and a miss returns the tables it has just built,
    never what the cache holds afterwards: threads that drive different ranks in one process (the simulated-rank tests) alternate
    angle tensors, and a key / value pair written in two steps handed one rank the other rank's tables."""
    if isinstance(freqs, (tuple, list)):
        return freqs
    entry = _COS_SIN_CACHE["entry"]
    if entry is not None and entry[0]() is freqs and entry[1] == freqs._version:
        return entry[2]
    val = ops.rope_cos_sin(freqs)
    _COS_SIN_CACHE["entry"] = (weakref.ref(freqs), freqs._version, val)
    return val


def apply_rotary_pos_emb(t: torch.Tensor, freqs, config=None, cu_seqlens=None) -> torch.Tensor:
    """apply_rotary_pos_emb(t, freqs, config, cu_seqlens) (:232-259) for t [s, b, heads, d]; `freqs` as Megatron passes it
    (fp32 [s, 1, 1, d]) or a (cos, sin) pair of this module's RotaryEmbedding.  Without autograd the rotation is in place
    (Megatron only uses the returned tensor); with it, autograd_fns.RopeFn rotates a copy and its backward rotates by -theta."""
    if cu_seqlens is not None:
        raise AssertionError("thd (packed) RoPE is not on this path")
    if config is not None and getattr(config, "rotary_interleaved", False):
        raise NotImplementedError("only non-interleaved RoPE is on the Long-VITA path")
    s, b, h, d = t.shape
    cos, sin = _cos_sin(freqs)
    if cos.shape[0] != s * b and b != 1:
        raise ValueError("batch must be 1 on the Long-VITA path")
    if torch.is_grad_enabled() and t.requires_grad:
        from .autograd_fns import RopeFn
        return RopeFn.apply(t, cos, sin)
    ops.rope_apply_(t.view(s, h, d) if t.is_contiguous() else t[:, 0], cos, sin)
    return t
