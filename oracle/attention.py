"""Unfused attention restatement.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import math

import torch

from .glue import zigzag_chunk_ids


class _BaddbmmChain(torch.autograd.Function):
    """torch.baddbmm(beta=0, alpha) on half-precision operands as autograd sees it (what Megatron's unfused path runs, :206-212):
    forward rounds alpha * (q k^T) ONCE; the backward formula `grad.bmm(other) * alpha` rounds the bmm result and then the scaled
    result — two roundings whenever alpha is not a power of two (hn = 128).  Operands arrive as fp32 copies of half values."""

    @staticmethod
    def forward(ctx, qf, kf, alpha, dtype):
        ctx.save_for_backward(qf, kf)
        ctx.alpha, ctx.dtype = alpha, dtype
        return (torch.matmul(qf, kf.transpose(-1, -2)) * alpha).to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        qf, kf = ctx.saved_tensors
        g = g.to(ctx.dtype).float()
        dq = torch.matmul(g, kf).to(ctx.dtype).float() * ctx.alpha
        dk = torch.matmul(g.transpose(-1, -2), qf).to(ctx.dtype).float() * ctx.alpha
        return dq, dk, None, None


def core_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool, scale: float = None,
                   q_pos: torch.Tensor = None, k_pos: torch.Tensor = None, cu_seqlens: torch.Tensor = None,
                   chain: bool = False) -> torch.Tensor:
    """M/core/transformer/dot_product_attention.py:171-175 (GQA repeat_interleave) + :186-289
    (baddbmm / softmax / bmm).  q [sq, b, np, hn], k/v [sk, b, ng, hn] -> [sq, b, np*hn].
    Computed in fp32 (attention_softmax_in_fp32), result cast to q.dtype.
    chain=True (with bf16 / fp16 inputs): the dtype chain of that UNFUSED path as Megatron runs it in half precision — the
    reference's CPU-capable attention: `torch.baddbmm(..., alpha = 1 / norm_factor)` writes the scores in q.dtype (:206-212, one
    rounding of the scaled product), scale_mask_softmax computes in fp32 and casts the probabilities back to q.dtype, `torch.bmm`
    (:268) rounds the context to q.dtype; under autograd the gradients are rounded at the same three tensors.
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
    chain = chain and q.dtype != torch.float32
    if chain:
        scores = _BaddbmmChain.apply(qf, kf, scale, q.dtype)
    else:
        scores = torch.matmul(qf, kf.transpose(-1, -2)) * scale      # [b, np, sq, sk]
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
    if chain:
        probs = probs.to(q.dtype).float()
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


def core_attention_row_blocked(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos0: int = 0, chain: bool = False,
                               score_bytes: int = 20 << 30) -> torch.Tensor:
    """`core_attention(causal=True)` for sequences whose [np, sq, sk] score tensor does not fit (128K / 1M rows): the same function
    called on causal ROW BLOCKS, one kv group at a time — rows [r0, r1) of group g against keys [0, q_pos0 + r1) only (a key beyond
    a row's position gets probability exactly 0 in the full evaluation, so leaving it out changes nothing but the reduction order).
    q [sq, b, np, hn] = rows q_pos0 .. q_pos0 + sq - 1 of the sequence; k / v [sk, b, ng, hn] = keys 0 .. sk - 1, sk >= q_pos0 + sq.
    Block height: the largest power of two whose fp32 scores (rep x rows x keys x 4 B) stay under `score_bytes`."""
    sq, b, np_, hn = q.shape
    sk, _, ng, _ = k.shape
    rep = np_ // ng
    assert sk >= q_pos0 + sq and np_ == rep * ng
    rows = 1 << max(int(score_bytes // (4 * rep * b * (q_pos0 + sq))).bit_length() - 1, 0)
    rows = max(min(rows, sq), 1)
    out = torch.empty(sq, b, np_, hn, dtype=q.dtype, device=q.device)
    for g in range(ng):
        for r0 in range(0, sq, rows):
            r1 = min(r0 + rows, sq)
            pos = torch.arange(q_pos0 + r0, q_pos0 + r1, device=q.device)
            o = core_attention(q[r0:r1, :, g * rep:(g + 1) * rep], k[:q_pos0 + r1, :, g:g + 1], v[:q_pos0 + r1, :, g:g + 1], True,
                               q_pos=pos, chain=chain)
            out[r0:r1, :, g * rep:(g + 1) * rep] = o.view(r1 - r0, b, rep, hn)
    return out.view(sq, b, np_ * hn)
