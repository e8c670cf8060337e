"""Prefill step of the CP-aware decode loop — mirror of
M/inference/text_generation/generation.py (one iteration of the `for context_length` loop, :123-205,
which IS the measured unit of the prefill metric).

  get_batch_on_this_cp_rank(tokens, position_ids, external_inputs)   :517-539
  logit-mask position rule                                            :141-165
  sync_output (all-gather of the masked logits + un-zig-zag)          :542-566
  last-token pick                                                     :179-205
  generate_tokens_probs_and_return_on_first_stage (the loop itself)   :33-280
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
    if external_inputs["images"].shape[0] == 0:      # inference_module's per-rank loader: no visual token on this rank
        r = mpu.get_context_parallel_rank()
        return training_utils.zigzag_slice(tokens, cp_size, r), training_utils.zigzag_slice(position_ids, cp_size, r), None
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


# ------------------------------------------------------------------------------------------------
# the decode loop
# ------------------------------------------------------------------------------------------------
def _sample_strategy(logits: torch.Tensor, do_sample=False, top_k=0, top_p=0.0, temperature=1.0):
    """:473-512 (_sample_strategy + top_k_logits) — greedy, or temperature / top-k / top-p sampling.  Sampling policy, not
    path arithmetic: plain torch on the [b, vocab] row.  Same filtering rules as the reference (top-k for top_k > 0: ties
    with the k-th value survive; top-p for top_p > 0: the first token above the threshold is kept) and the same return value
    (the filtered softmax when sampling, the untouched logits otherwise); the caller's tensor is never modified (the
    reference divides an fp32 input in place)."""
    if not do_sample:
        return logits, torch.argmax(logits, dim=-1).view(-1)
    logits = logits.float().clone()
    logits /= temperature
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = float("-inf")
    if top_p > 0.0:
        srt, idx = torch.sort(logits, descending=True, dim=-1)
        drop = torch.cumsum(torch.softmax(srt, dim=-1), dim=-1) > top_p
        drop[..., 1:] = drop[..., :-1].clone()
        drop[..., 0] = False
        logits[drop.gather(-1, idx.argsort(dim=-1))] = float("-inf")      # un-sort the mask (the reference loops over rows)
    probs = torch.softmax(logits, dim=-1)
    return probs, torch.multinomial(probs, num_samples=1).view(-1)


def _cp_prefill_length(prompt_len: int, cp_size: int) -> int:
    """Tokens a CP prefill is fed: the prompt padded up to a multiple of 2*CP*256 (the attention kernel's
    q-tile granularity per zig-zag chunk; the reference requires 2*CP, M/training/arguments.py:211-213)."""
    unit = 2 * cp_size * 256
    return -(-prompt_len // unit) * unit


@torch.no_grad()
def generate_tokens_probs_and_return_on_first_stage(model, tokens, lengths, return_output_log_probs=False,
                                                    do_sample=False, top_k=0, top_p=0.0, temperature=1.0,
                                                    use_eod_token_for_early_termination=True, external_inputs=None,
                                                    *, use_kv_cache=True, logit_mask=True, termination_id=None,
                                                    reference_compat=False):
    """Generator with the reference's contract (:33-280): `tokens` [1, max_sequence_length] holds the prompt
    (length lengths[0]) followed by padding and is filled in place; yields (tokens[:, :ctx+1], lengths,
    output_log_probs) per generated token.  args.use_kv_cache / args.logit_mask / the tokenizer's eod are keyword
    arguments here (the reference reads them from get_args(), :71-76).

    use_kv_cache=False is the reference's CP behaviour: the whole buffer is re-prefilled for every token
    (:127-135).  use_kv_cache=True prefills once — under CP on the prompt padded to _cp_prefill_length — and then
    feeds one token per step to the sharded cache; every CP rank gets identical logits, so there is no
    sync_output on those steps."""
    from .inference_params import ForwardStep
    batch_size, max_sequence_length = tokens.shape
    if batch_size != 1:
        raise ValueError("the Long-VITA decode loop runs batch 1")
    min_prompt_length = int(lengths.min().item())
    if min_prompt_length >= max_sequence_length:
        raise ValueError("context length + tokens_to_generate too large")                  # :85-86
    cp_size = mpu.get_context_parallel_world_size()
    forward_step = ForwardStep(model, batch_size, max_sequence_length, external_inputs=None)
    ip = forward_step.inference_params
    ip.use_kv_cache = use_kv_cache
    output_log_probs = None
    if return_output_log_probs:
        output_log_probs = torch.empty((batch_size, max_sequence_length - 1, model.cfg.vocab), dtype=torch.float32,
                                       device=tokens.device)
    position_ids = torch.arange(max_sequence_length, dtype=torch.long, device=tokens.device)[None]
    prev_context_length = 0
    context_length = min_prompt_length
    for context_length in range(min_prompt_length, max_sequence_length):
        first = prev_context_length == 0
        if use_kv_cache and not first:
            # cached step: one token, replicated on every rank
            ip.logit_mask = None
            logits = forward_step(tokens[:, prev_context_length:context_length],
                                  position_ids[:, prev_context_length:context_length], None)
            last_token_logits = logits[:, -1, :]
        else:
            if use_kv_cache:
                fed = context_length if cp_size == 1 else _cp_prefill_length(context_length, cp_size)
                if fed > max_sequence_length:
                    tokens_in = torch.nn.functional.pad(tokens[:, :context_length], (0, fed - context_length))
                else:
                    tokens_in = tokens[:, :fed]
                pos_in = torch.arange(fed, dtype=torch.long, device=tokens.device)[None]
                ip.prefill_valid_tokens = context_length
            else:
                tokens_in, pos_in = tokens, position_ids
            if external_inputs:
                tokens2use, positions2use, ext2use = get_batch_on_this_cp_rank(tokens_in, pos_in, external_inputs)
            elif cp_size > 1:
                r = mpu.get_context_parallel_rank()
                tokens2use = training_utils.zigzag_slice(tokens_in, cp_size, r)
                positions2use = training_utils.zigzag_slice(pos_in, cp_size, r)
                ext2use = None
            else:
                tokens2use, positions2use, ext2use = tokens_in, pos_in, None
            ip.external_inputs = ext2use
            block = None
            if logit_mask:
                ip.logit_mask, block = build_logit_mask(tokens2use, context_length, reference_compat)
            logits = sync_output(forward_step(tokens2use, positions2use, None))
            if logit_mask:
                last_token_logits = logits[:, -1, :] if block is None else logits[:, block, :]
            else:
                last_token_logits = logits[:, context_length - 1, :]
        _, new_sample = _sample_strategy(last_token_logits, do_sample=do_sample, top_k=top_k, top_p=top_p,
                                         temperature=temperature)
        started = lengths <= context_length
        tokens[started, context_length] = new_sample[started]
        if return_output_log_probs:
            output_log_probs[:, context_length - 1, :] = torch.log_softmax(last_token_logits.float(), dim=1)
        prev_context_length = context_length
        yield tokens[:, : context_length + 1], lengths, output_log_probs
        if (use_eod_token_for_early_termination and termination_id is not None
                and bool((new_sample == termination_id).all())):
            break
