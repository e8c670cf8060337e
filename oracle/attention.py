"""Unfused attention restatement.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import math

import torch

from .glue import zigzag_chunk_ids


def core_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool, scale: float = None,
                   q_pos: torch.Tensor = None, k_pos: torch.Tensor = None, cu_seqlens: torch.Tensor = None) -> torch.Tensor:
    """M/core/transformer/dot_product_attention.py:171-175 (GQA repeat_interleave) + :186-289
    (baddbmm / softmax / bmm).  q [sq, b, np, hn], k/v [sk, b, ng, hn] -> [sq, b, np*hn].
    Computed in fp32 (attention_softmax_in_fp32), result cast to q.dtype.
    q_pos / k_pos: global positions for the causal rule (default arange).
    cu_seqlens: packed samples (flash_attn_varlen_func branch, :334-367) — a query only sees keys of its own sample."""
    sq, b, np_, hn = q.shape
    sk, _, ng, _ = k.shape
    rep = np_ // ng
    if rep > 1:
        k = k.repeat_interleave(rep, dim=2)
        v = v.repeat_interleave(rep, dim=2)
    scale = 1.0 / math.sqrt(hn) if scale is None else scale
    qf = q.float().permute(1, 2, 0, 3)          # [b, np, sq, hn]
    kf = k.float().permute(1, 2, 0, 3)
    vf = v.float().permute(1, 2, 0, 3)
    scores = torch.matmul(qf, kf.transpose(-1, -2)) * scale          # [b, np, sq, sk]
    if causal:
        dev = q.device                                   # (the tensors' device: the 16K / 48-layer parity test evaluates this on the GPU)
        qp = torch.arange(sq, device=dev) if q_pos is None else q_pos.to(dev)
        kp = torch.arange(sk, device=dev) if k_pos is None else k_pos.to(dev)
        mask = kp[None, :] > qp[:, None]
        scores = scores.masked_fill(mask[None, None], float("-inf"))
    if cu_seqlens is not None:
        assert causal and sq == sk
        seg = torch.bucketize(torch.arange(sq), cu_seqlens.to(torch.int64)[1:], right=True)     # sample id of every row
        scores = scores.masked_fill((seg[:, None] != seg[None, :])[None, None], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, vf)                                   # [b, np, sq, hn]
    return ctx.permute(2, 0, 1, 3).reshape(sq, b, np_ * hn).to(q.dtype)


def zigzag_cp_attention(q_local, k_locals, v_locals, cp_size: int, cp_rank: int, seq_len: int, scale=None):
    """Context-parallel causal attention for one rank: local q [S_l, b, np, hn] against the K/V of
    every rank (lists of [S_l, b, ng, hn] in rank order), with the zig-zag global positions of
    M/training/utils.py:329-341.  Equivalent to what TE's AttnFuncWithCP ring computes."""
    C = seq_len // (2 * cp_size)

    def positions(r):
        a, b_ = zigzag_chunk_ids(cp_size, r)
        return torch.cat([torch.arange(a * C, (a + 1) * C), torch.arange(b_ * C, (b_ + 1) * C)])

    k_all = torch.cat(k_locals, dim=0)
    v_all = torch.cat(v_locals, dim=0)
    k_pos = torch.cat([positions(r) for r in range(cp_size)])
    return core_attention(q_local, k_all, v_all, True, scale, q_pos=positions(cp_rank), k_pos=k_pos)
