"""Zig-zag context-parallel batch slice — device-side mirror of M/training/utils.py.

`get_batch_on_this_cp_rank` keeps the reference's name, argument meaning and key behaviour
(M/training/utils.py:252-343; inference twin M/inference/text_generation/generation.py:517-539):
  * every [b, s, ...] tensor is viewed as 2*CP chunks and the rank keeps chunks {r, 2CP-1-r};
  * `external_images` keeps only frames with at least one token on this rank;
  * `external_indices` [2, N, L] is replaced by `external_src_indices` (row in the *selected*
    frames, token column) and `external_tgt_indices` (batch, *local* position).
The index arithmetic runs in HIP kernels (csrc/rows.hip); the reference's
`torch.isin` + `index_of_a_in_b` (:347-350) is replaced by the closed form
chunk = pos // C, local = (chunk == r ? 0 : C) + pos % C, which is bit-identical for the unique
positions the reference requires.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops, parallel_state as mpu

_ACTUAL_SEQ_LEN = None
_POSITION_IDS = None


def get_actual_seq_len():
    return _ACTUAL_SEQ_LEN


def set_actual_seq_len(actual_seq_len):
    global _ACTUAL_SEQ_LEN
    _ACTUAL_SEQ_LEN = actual_seq_len


def get_position_ids():
    return _POSITION_IDS


def set_position_ids(position_ids):
    global _POSITION_IDS
    _POSITION_IDS = position_ids


def zigzag_slice(val: torch.Tensor, cp_size: int, cp_rank: int, seq_dim: int = 1) -> torch.Tensor:
    """val[..., s, ...] -> the rank's two chunks, concatenated (utils.py:329-341)."""
    s = val.shape[seq_dim]
    if s % (2 * cp_size):
        raise ValueError(f"sequence length {s} not divisible by 2*CP = {2 * cp_size}")
    c = s // (2 * cp_size)
    a, b = mpu.zigzag_chunk_ids(cp_size, cp_rank)
    return torch.cat([val.narrow(seq_dim, a * c, c), val.narrow(seq_dim, b * c, c)], dim=seq_dim)


def get_batch_on_this_cp_rank(batch: dict, seq_length: Optional[int] = None, cp_size: Optional[int] = None,
                              cp_rank: Optional[int] = None, reset_position_ids: bool = False) -> dict:
    cp_size = mpu.get_context_parallel_world_size() if cp_size is None else cp_size
    cp_rank = mpu.get_context_parallel_rank() if cp_rank is None else cp_rank
    if reset_position_ids:
        set_position_ids(batch["position_ids"].transpose(0, 1).contiguous())      # utils.py:267-270
    if cp_size <= 1:
        return batch
    batch = dict(batch)
    if seq_length is None:
        seq_length = next(v.shape[1] for k, v in batch.items()
                          if v is not None and not k.startswith("external_") and k != "attention_mask")

    hit = local = selected = None
    if "external_indices" in batch:
        ind = batch["external_indices"]
        indices_s = ind[1].contiguous()
        hit, local = ops.cp_index_remap(indices_s, seq_length, cp_size, cp_rank)    # [N, L] uint8 / int64
        any_hit = bool(hit.any().item())
        if any_hit:
            selected = ops.mask_to_index(ops.rows_any(hit))                          # frames on this rank

    for key, val in list(batch.items()):
        if key == "external_images":
            if selected is not None:
                batch[key] = ops.row_gather(val.contiguous(), selected)
            continue
        if key == "external_indices":
            if selected is not None:
                n_img, tok = hit.shape
                hit_idx = ops.mask_to_index(hit)
                img_rank = ops.index_inverse(selected, n_img)
                sb, ss, tb, ts = ops.cp_src_tgt(hit_idx, tok, img_rank, val[0].reshape(-1), local.reshape(-1))
                batch["external_src_indices"] = torch.stack([sb, ss])
                batch["external_tgt_indices"] = torch.stack([tb, ts])
            batch.pop(key)
            continue
        if key == "attention_mask" or val is None:
            continue
        batch[key] = zigzag_slice(val, cp_size, cp_rank, seq_dim=1)
    return batch


def get_packed_segments():
    """Packed samples as the reference's GPU path detects them: transformers' _flash_attention_forward
    (called at M/core/transformer/dot_product_attention.py:374-390 with position_ids=get_position_ids()) switches to
    flash_attn_varlen when the position ids are not monotonic, with cu_seqlens at the zeros of position_ids.
    Returns (seg_start, seg_end) int32 [s] on the device, or None.  CP = 1 only."""
    pid = _POSITION_IDS
    if pid is None:
        return None
    cached = getattr(get_packed_segments, "_cache", None)
    if cached is not None and cached[0] is pid:
        return cached[1]
    pos = pid.reshape(pid.shape[0], -1)[:, 0]                              # stored [s, b] (:268-270), b == 1
    seg = None
    if pos.numel() > 1 and not bool((torch.diff(pos) >= 0).all()):
        from . import ops
        cu = (pos == 0).nonzero().flatten()
        if cu.numel() == 0 or int(cu[0]) != 0:
            cu = torch.cat([cu.new_zeros(1), cu])
        seg = ops.segments_from_cu_seqlens(cu, pos.numel())
    get_packed_segments._cache = (pid, seg)
    return seg
