"""Prefill step of the CP-aware decode loop — mirror of
M/inference/text_generation/generation.py (one iteration of the `for context_length` loop, :123-205,
which IS the measured unit of the prefill metric).

  get_batch_on_this_cp_rank(tokens, position_ids, external_inputs)   :517-539
  logit-mask position rule                                            :141-165
  sync_output (all-gather of the masked logits + un-zig-zag)          :542-566
  last-token pick                                                     :179-205
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import parallel_state as mpu
from . import training_utils


def get_batch_on_this_cp_rank(tokens, position_ids, external_inputs):
    """:517-539 — identical contract (incl. KeyError when the rank owns no visual token, quirk 1)."""
    cp_size = mpu.get_context_parallel_world_size()
    if cp_size == 1:
        return tokens, position_ids, external_inputs
    batch = {"tokens": tokens, "position_ids": position_ids,
             "external_images": external_inputs["images"], "external_indices": external_inputs["indices"]}
    batch = training_utils.get_batch_on_this_cp_rank(batch, seq_length=tokens.shape[1])
    ext = {"images": batch["external_images"], "src_indices": batch["external_src_indices"],
           "tgt_indices": batch["external_tgt_indices"]}
    return batch["tokens"], batch["position_ids"], ext


def build_logit_mask(tokens2use: torch.Tensor, context_length: int, reference_compat: bool = True):
    """:141-165.  Returns (logit_mask [b, s_local] bool, cp_output_block or None).

    reference_compat=True reproduces the reference bit for bit, including its behaviour when
    context_length is a multiple of S/(2*CP) (index -1 wraps, block off by one — SURVEY.md §9
    quirk 2).  reference_compat=False marks the position of token context_length-1 instead."""
    logit_mask = torch.zeros_like(tokens2use).bool()
    cp_size = mpu.get_context_parallel_world_size()
    if cp_size == 1:
        logit_mask[:, context_length - 1] = 1
        return logit_mask, None
    half = int(tokens2use.size(1) / 2)
    if reference_compat:
        ccl = context_length % half
        block = context_length // half
        logit_mask[:, ccl - 1] = 1
        logit_mask[:, half + ccl - 1] = 1
    else:
        ccl = (context_length - 1) % half
        block = (context_length - 1) // half
        logit_mask[:, ccl] = 1
        logit_mask[:, half + ccl] = 1
    return logit_mask, block


def sync_output(output: torch.Tensor) -> torch.Tensor:
    """:542-566 — all-gather [b, 2, V] logits over CP and order the 2*CP halves by chunk id."""
    cp_size, cp_rank = mpu.get_context_parallel_world_size(), mpu.get_context_parallel_rank()
    if cp_size == 1:
        return output
    group = mpu.get_context_parallel_group()
    flat = torch.empty((cp_size * output.shape[0],) + tuple(output.shape[1:]), dtype=output.dtype,
                       device=output.device)                       # concat-on-dim-0 form (gloo and RCCL)
    dist.all_gather_into_tensor(flat, output.contiguous(), group=group)
    gathered = flat.view((cp_size,) + tuple(output.shape))
    halves = [h for r in range(cp_size) for h in gathered[r].chunk(2, dim=1)]
    # chunk ids are a pure function of (cp_size, rank): no second all-gather needed (:552-557)
    ids = [i for r in range(cp_size) for i in mpu.zigzag_chunk_ids(cp_size, r)]
    order = sorted(range(2 * cp_size), key=lambda j: ids[j])
    return torch.cat([halves[j] for j in order], dim=1)


@torch.no_grad()
def prefill_step(model, tokens: torch.Tensor, context_length: int, external_inputs: Optional[dict] = None,
                 reference_compat: bool = True) -> torch.Tensor:
    """One full-sequence forward producing the logits of token `context_length - 1`
    ([b, vocab]); tokens [1, S] on every rank, external_inputs = {"images", "indices"} (global)."""
    _, seq_length = tokens.size()
    position_ids = torch.arange(seq_length, dtype=torch.long, device=tokens.device).unsqueeze(0).expand_as(tokens)
    if external_inputs:
        tokens2use, positions2use, ext2use = get_batch_on_this_cp_rank(tokens, position_ids, external_inputs)
    else:
        cp = mpu.get_context_parallel_world_size()
        if cp > 1:
            tokens2use = training_utils.zigzag_slice(tokens, cp, mpu.get_context_parallel_rank())
            positions2use = training_utils.zigzag_slice(position_ids, cp, mpu.get_context_parallel_rank())
        else:
            tokens2use, positions2use = tokens, position_ids
        ext2use = None
    logit_mask, block = build_logit_mask(tokens2use, context_length, reference_compat)
    logits = model(tokens2use, positions2use, None, external_inputs=ext2use, logit_mask=logit_mask)
    logits = sync_output(logits)
    if block is None:
        return logits[:, -1, :]                                                      # :183-187
    return logits[:, block, :]                                                       # :190
