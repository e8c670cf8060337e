"""InferenceParams + ForwardStep — the decode-loop state the reference threads through the model.

InferenceParams mirrors Megatron's `megatron.core.InferenceParams` (max_batch_size,
max_sequence_length, sequence_len_offset, batch_size_offset, key_value_memory_dict,
swap_key_value_dict) plus the three attributes Long-VITA hangs on it: `external_inputs`
(M/inference/text_generation/forward_step.py:27-39), `logit_mask` and `use_kv_cache`
(M/inference/text_generation/generation.py:139-168, read back in
M/core/models/multimodal/gpt_vl_model.py:261-286).

What differs: the reference disables the cache under context parallelism (server_cp .sh:184) and
re-prefills the whole sequence per generated token.  Here `key_value_memory_dict[layer_number]` is this
rank's *shard* of the cache — one tensor [2, capacity, kv_groups, head_dim] holding the K/V rows
of the rank's two zig-zag chunks (valid prompt rows compacted to the front) followed by the rows of
the generated tokens this rank owns (round-robin over the CP ranks).  A new token attends to every
shard with vita_decode_attn_partial and the per-rank partials are merged after one small all-gather.
"""
from __future__ import annotations

import torch


class InferenceParams:
    def __init__(self, max_batch_size: int, max_sequence_length: int):
        self.max_sequence_length = max_sequence_length
        self.max_batch_size = max_batch_size
        self.sequence_len_offset = 0
        self.batch_size_offset = 0
        self.key_value_memory_dict = {}
        # attributes the Long-VITA decode loop attaches
        self.external_inputs = None
        self.logit_mask = None
        self.use_kv_cache = True
        # shard bookkeeping
        self.prefill_valid_tokens = None     # global count of real (unpadded) prompt tokens of a CP prefill
        self.local_len = 0                   # valid rows in this rank's shard
        self.decode_steps = 0                # generated tokens appended so far (owner = decode_steps % CP)
        self.consumed_tokens = None          # set by the model: global tokens consumed by the last forward

    def swap_key_value_dict(self, batch_idx):
        """Beam search reorders the batch dimension of the cache; the Long-VITA path runs batch 1."""
        if len(batch_idx) != 1 or int(batch_idx[0]) != 0:
            raise NotImplementedError("the sharded cache holds batch 1")

    def reset(self):
        self.sequence_len_offset = 0
        self.batch_size_offset = 0
        self.key_value_memory_dict = {}
        self.prefill_valid_tokens = None
        self.local_len = 0
        self.decode_steps = 0


class ForwardStep:
    """M/inference/text_generation/forward_step.py (ForwardStep.__init__ wrapper :27-39 and
    _no_pipelining_forward_step :42-57): owns the InferenceParams, calls the model, advances
    sequence_len_offset by the tokens fed."""

    def __init__(self, model, max_batch_size: int, max_sequence_length: int, external_inputs=None):
        self.model = model
        self.inference_params = InferenceParams(max_batch_size, max_sequence_length)
        self.inference_params.external_inputs = external_inputs

    @torch.no_grad()
    def __call__(self, tokens, position_ids, attention_mask=None):
        ip = self.inference_params
        logits = self.model(tokens, position_ids, attention_mask, inference_params=ip)
        if ip is not None and getattr(ip, "use_kv_cache", True):
            # the model reports the GLOBAL tokens it consumed (a CP prefill is fed a local slice of a padded prompt)
            ip.sequence_len_offset += ip.consumed_tokens if ip.consumed_tokens is not None else tokens.size(1)
            ip.consumed_tokens = None
        return logits
