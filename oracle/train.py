"""Training-step oracle: loss and gradients by torch autograd over the forward restatements
(what the reference does: autograd over its modules; loss semantics of
M/core/models/multimodal/gpt_vl_model.py:371-416 + M/pretrain_long_vita.py:778-838).
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import glue, llm as ollm
from .attention import core_attention


def _leafify(p):
    if isinstance(p, dict):
        return {k: _leafify(v) for k, v in p.items()}
    if isinstance(p, list):
        return [_leafify(v) for v in p]
    return p.detach().clone().requires_grad_(True)


def _grads(p):
    if isinstance(p, dict):
        return {k: _grads(v) for k, v in p.items()}
    if isinstance(p, list):
        return [_grads(v) for v in p]
    return p.grad


def loss_and_grads(tokens, labels, loss_mask, params, cfg: ollm.LLMConfig, cp_size: int = 1,
                   is_instruction: bool = True, feature_fn=None, indices=None, position_ids=None,
                   output_multiplier_scale=None, output_logit_softcapping=None):
    """Mean cross-entropy over the logit-masked rows, with the reference's per-rank selection and
    instruction shift: rank r keeps the masked positions among ITS zig-zag positions (local order),
    pairs logits[k] with labels[k+1] of that selection (gpt_vl_model.py:380-391), all ranks' pairs are
    averaged together (pretrain_long_vita.py:793-803).  cp_size = 1 is the plain case.
    feature_fn(params) -> features [N, L, hidden] (differentiable) scattered at `indices` [2, N, L].
    position_ids [1, S] with resets (--reset-position-ids, M/training/utils.py:221-245): RoPE follows them
    (rotary_pos_embedding.py:114-117) and, being non-monotonic, they make transformers' _flash_attention_forward treat
    the row as packed samples — cu_seqlens = positions of the zeros (dot_product_attention.py:374-390)."""
    p = _leafify(params)
    S = tokens.shape[1]
    we = p["embed"][tokens]
    efd = None
    if feature_fn is not None:
        efd = {"features": feature_fn(p), "indices": indices}
    h = glue.embedding_scatter(we, efd)
    cu = None
    pos_sb = None
    if position_ids is not None:
        pos_sb = position_ids.transpose(0, 1)
        flat = position_ids[0]
        if not bool((torch.diff(flat) >= 0).all()):
            cu = torch.cat([(flat == 0).nonzero().flatten(), torch.tensor([S])]).to(torch.int32)
    freqs = glue.rope_emb(S, glue.rope_inv_freq(cfg.head_dim, cfg.rope_theta), pos_sb)
    for lp in p["layers"]:
        h, _ = ollm.decoder_layer(h, lp, cfg, freqs,
                                  lambda q, k, v: core_attention(q, k, v, causal=True, cu_seqlens=cu))
    h = glue.rmsnorm(h, p["final_ln"], cfg.eps)                               # [S, 1, hidden]
    losses = []
    for r in range(cp_size):
        pos = glue.calibration_index(S, cp_size, r) if cp_size > 1 else torch.arange(S)
        sel = pos[loss_mask[0, pos].bool()]
        if sel.numel() == 0:
            continue
        logits = ollm.linear(h[sel, 0], p["lm_head"])                         # bf16 logits like the GPU GEMM
        logits = glue.logit_postprocess(logits, output_multiplier_scale, output_logit_softcapping)     # gpt_vl_model.py:349-355
        lab = labels[0, sel]
        if is_instruction:
            logits, lab = logits[:-1], lab[1:]
        if logits.shape[0]:
            losses.append(F.cross_entropy(logits.float(), lab, reduction="none"))
    allv = torch.cat(losses)
    loss = allv.sum() / allv.numel()
    loss.backward()
    return loss.detach(), _grads(p)
